cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/hang
run () {  # tag, env..., events
  tag=$1; shift
  ( time ( env "$@" timeout 75 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/hang/$tag -o p -- python3 scripts/dev/setup_probe.py $EV > gpurun_out/hang/$tag.out 2> gpurun_out/hang/$tag.log; echo "$tag rc $?" ) ) 2>&1 | grep "rc\|real"
  rm -rf gpurun_out/hang/$tag
  tail -1 gpurun_out/hang/$tag.out | cut -c1-200
}
EV=4e7
run native_4e7 PISA_HIP_UPLOAD_THREADS=0
run native_4e7_threads PISA_HIP_UPLOAD_THREADS=3
