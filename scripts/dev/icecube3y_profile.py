"""host profile (cProfile, tottime) of one step of the published 3-year analysis through DistributionMaker
(bench leg icecube3y_boundary, all 16 parameters moving)"""
import cProfile
import os
import pstats
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
tmp = tempfile.mkdtemp(prefix="pisa_hip_3y_")
subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "make_synthetic_3y_mc.py"), tmp, "200000", "3"],
                      stdout=subprocess.DEVNULL)
os.environ["PISA_RESOURCES"] = tmp
from pisa_amd.core.distribution_maker import DistributionMaker  # noqa: E402
from pisa_amd.core.pipeline import Pipeline  # noqa: E402

template = DistributionMaker(["settings/pipeline/IceCube_3y_neutrinos.cfg", "settings/pipeline/IceCube_3y_muons.cfg"])
data = Pipeline("settings/pipeline/IceCube_3y_data.cfg").get_outputs()[0]
free = [p.name for p in template.params.free]
rs = np.random.RandomState(1)


def step(which=free):
    for name in which:
        p = template.params[name]
        lo, hi = ((p.range[0].magnitude, p.range[1].magnitude) if p.range is not None
                  else (p.value.magnitude * 0.9, p.value.magnitude * 1.1))
        p.value = (p.nominal_value.magnitude + 0.05 * (hi - lo) * (rs.rand() - 0.5)) * p.value.units
    total = template.get_outputs(return_sum=True)[0]
    return data.metric_total(expected_values=total, metric="mod_chi2")


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    step()
torch.cuda.synchronize()
print("==== %.1f us per step (16 free parameters)" % ((time.perf_counter() - t0) / 200 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)
st.sort_stats("cumulative").print_stats(45)
