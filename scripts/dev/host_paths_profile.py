"""host profiles (cProfile, tottime) of the three reference-API paths: FastPlan replay, Stage protocol in full
(example_hip.cfg), the unmodified osc_example.cfg (C1).  usage: host_paths_profile.py [events] [which,...]"""
import cProfile
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from pisa_amd.core.pipeline import Pipeline  # noqa: E402
from pisa_amd.core.units import ureg  # noqa: E402

n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e6
which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["fast", "stage", "c1"]
N_PROF = 200


def profile(tag, fn, n_time=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_time):
        fn()
    torch.cuda.synchronize()
    print("==== %s: %.1f us per evaluation" % (tag, (time.perf_counter() - t0) / n_time * 1e6))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(N_PROF):
        fn()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(32)
    st.sort_stats("cumulative").print_stats(28)


if "fast" in which or "stage" in which:
    pipe = Pipeline(bench._pipeline_cfg(n))
    data = sum(pipe.get_outputs()).fluctuate("poisson", random_state=0)
    rs = np.random.RandomState(1)

    def one():
        pipe.params.theta23.value = (31.0 + 28.0 * rs.rand()) * ureg.degree
        pipe.params.deltam31.value = (1e-3 + 6e-3 * rs.rand()) * ureg.eV ** 2
        return data.metric_total(expected_values=sum(pipe.get_outputs()), metric="llh")

    if "fast" in which:
        profile("FastPlan replay (example_hip.cfg, %g events)" % n, one)
    if "stage" in which:
        pipe.fast_path = False
        pipe._plan = None
        profile("Stage protocol in full (example_hip.cfg, %g events)" % n, one, n_time=100)
    del pipe
if "c1" in which:
    pipe = Pipeline("settings/pipeline/osc_example.cfg")
    pipe.get_outputs()
    k = [0]

    def one_c1():
        k[0] += 1
        pipe.params.theta23.value = (40.0 + 0.01 * k[0]) * ureg.degree
        maps = pipe.get_outputs()
        return float(maps[1].hist[0, 0])

    profile("osc_example.cfg unmodified (C1)", one_c1, n_time=200)
