# coordinate-form (SURVEY 8d unit of work) kernel: launch time against debug switches / workgroup count
for d in 0 2; do for b in 512 2400; do
PISA_HIP_HIST_BLOCKS=$b PISA_HIP_HIST_DBG=$d python bench.py --legs coordinate_form --steps 60 --no-batch-probe --no-drop-probe 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('dbg',$d,'blocks',$b, d['legs']['coordinate_form']['roofline']['avg_launch_ms'])"
done; done
