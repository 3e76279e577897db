cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for sp in 1 0; do
PISA_HIP_SPLIT_TAIL=$sp python bench.py --legs multi_point --no-cpu-baseline --no-drop-probe 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); m=d['legs']['multi_point']; print('split $sp: step %.2f us  K3 %.2f K5 %.2f K9 %.2f us/point' % (d['ms_per_step']*1e3, m['K3']['us_per_point'], m['K5']['us_per_point'], m['K9']['us_per_point']))"
done; done
