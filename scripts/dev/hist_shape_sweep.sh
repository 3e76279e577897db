# headline step against the accumulate kernels' launch shape; lines "workgroups threads replicas events" on stdin
# (workgroups 0 = the default: one per CU, dealt by load)
cd $GRAFT_REPO_ROOT
export PISA_HIP_LIB=${PISA_HIP_LIB:-${GRAFT_REPO_ROOT:-$PWD}/pisa_amd/libpisa_hip_dev.so}   # development build: make -C pisa_amd/csrc dev
while read -r b t c ev; do
  [ -z "$b" ] && continue
  if [ "$b" = 0 ]; then unset PISA_HIP_HIST_BLOCKS; else export PISA_HIP_HIST_BLOCKS=$b; fi
  PISA_HIP_HIST_THREADS=$t PISA_HIP_HIST_COPIES=$c python bench.py --events $ev --legs none --no-cpu-baseline --no-drop-probe 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('blocks $b threads $t copies $c events $ev: %.2f us/step, fused %.2f us, frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['avg_launch_ms']*1e3, d['roofline']['frac']))"
done
