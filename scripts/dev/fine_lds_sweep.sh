cd $GRAFT_REPO_ROOT
for kb in 64 96 128 150 64; do
PISA_HIP_HIST_LDS_KB=$kb python bench.py --binning fine3d --legs none --no-cpu-baseline --no-drop-probe 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('lds $kb KB: %.2f us/step, fused %.2f us, frac %.3f, llh %r' % (d['ms_per_step']*1e3, d['roofline']['avg_launch_ms']*1e3, d['roofline']['frac'], d['last_llh']))"
done
