"""Boundary-level cost of one step of the published 3-year analysis (neutrinos + muons
DistributionMaker, all systematics moving, metric against the released data histogram)."""
import os, subprocess, sys, time, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
n = sys.argv[1] if len(sys.argv) > 1 else "1e6"
tmp = tempfile.mkdtemp()
subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "make_synthetic_3y_mc.py"), tmp, n, "3"])
os.environ["PISA_RESOURCES"] = tmp
from pisa_amd.core.distribution_maker import DistributionMaker
from pisa_amd.core.pipeline import Pipeline
from pisa_amd.core.units import ureg

template = DistributionMaker(["settings/pipeline/IceCube_3y_neutrinos.cfg", "settings/pipeline/IceCube_3y_muons.cfg"])
data = Pipeline("settings/pipeline/IceCube_3y_data.cfg").get_outputs()[0]
free = [p.name for p in template.params.free]
print("free:", free)
rs = np.random.RandomState(1)


def step(which):
    for name in which:
        p = template.params[name]
        lo, hi = (p.range[0].magnitude, p.range[1].magnitude) if p.range is not None else (p.value.magnitude * 0.9, p.value.magnitude * 1.1)
        nominal = p.nominal_value.magnitude
        p.value = (nominal + 0.05 * (hi - lo) * (rs.rand() - 0.5)) * p.value.units
    total = template.get_outputs(return_sum=True)[0]
    return data.metric_total(expected_values=total, metric="mod_chi2")


for label, which in (("all free params", free), ("osc only", [f for f in free if f in ("theta23", "deltam31")]),
                     ("flux only", [f for f in free if f in ("nue_numu_ratio", "delta_index", "Barr_uphor_ratio", "Barr_nu_nubar_ratio")])):
    for _ in range(3):
        step(which)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = 30
    for _ in range(k):
        v = step(which)
    torch.cuda.synchronize()
    print("%-16s %.3f ms per step (metric %.6g)" % (label, (time.perf_counter() - t0) / k * 1e3, v))

if len(sys.argv) > 2 and sys.argv[2] == "profile":
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    which = free if len(sys.argv) < 4 else [f for f in free if f in ("theta23", "deltam31")]
    for _ in range(30):
        step(which)
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)
    st.sort_stats("tottime").print_stats(25)
