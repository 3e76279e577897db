# headline step against the fused kernel's launch shape: "workgroups threads replicas deep events"
cd $GRAFT_REPO_ROOT
while read -r b t c d ev; do
  [ -z "$b" ] && continue
  PISA_HIP_HIST_BLOCKS=$b PISA_HIP_HIST_THREADS=$t PISA_HIP_HIST_COPIES=$c PISA_HIP_HIST_DEEP=$d python bench.py --events $ev --legs none --no-cpu-baseline --no-drop-probe 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('blocks $b threads $t copies $c deep $d events $ev: %.2f us/step, fused %.2f us, frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['avg_launch_ms']*1e3, d['roofline']['frac']))"
done
