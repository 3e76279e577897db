# first construction of an engine in a fresh process, without and with pisa_amd.warm_up() (round 6)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/first_setup
for rep in 1 2; do
  echo "--- plain"; python3 scripts/dev/setup_probe.py 1e7 2>&1 | grep '"rep"' | cut -c1-260
  echo "--- warm_up"; SETUP_PROBE_WARM_UP=1 python3 scripts/dev/setup_probe.py 1e7 2>&1 | grep 'rep\|warm_up_ms' | cut -c1-260
done | tee gpurun_out/first_setup/first_setup.txt
