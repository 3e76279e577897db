"""host profile of one evaluation of the unmodified osc_example.cfg (BASELINE C1)"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pisa_amd.core.pipeline import Pipeline
from pisa_amd.core.units import ureg

pipe = Pipeline("settings/pipeline/osc_example.cfg")
pipe.get_outputs()
def run(n):
    tot = 0.0
    for i in range(n):
        pipe.params.theta23.value = (40.0 + 0.05 * i) * ureg.degree
        maps = pipe.get_outputs()
        tot += float(maps[1].hist[0, 0])
    return tot
run(20)
torch.cuda.synchronize(); t0 = time.perf_counter(); run(200); torch.cuda.synchronize()
print("%.1f us per evaluation" % ((time.perf_counter() - t0) / 200 * 1e6))
pr = cProfile.Profile(); pr.enable(); run(100); pr.disable()
pstats.Stats(pr).sort_stats(os.environ.get("SORT", "cumulative")).print_stats(30)
