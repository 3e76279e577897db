"""where one evaluation through the Pipeline boundary (FastPlan replay) spends its wall time: medians of the
host-side intervals of `one()` at the headline size"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from pisa_amd.core.pipeline import Pipeline  # noqa: E402
from pisa_amd.core.units import ureg  # noqa: E402

n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e7
pipe = Pipeline(bench._pipeline_cfg(n))
data = sum(pipe.get_outputs()).fluctuate("poisson", random_state=0)
rs = np.random.RandomState(1)
T = []
pc = time.perf_counter


def one():
    t0 = pc()
    a = (31.0 + 28.0 * rs.rand()) * ureg.degree
    b = (1e-3 + 6e-3 * rs.rand()) * ureg.eV ** 2
    t1 = pc()
    pipe.params.theta23.value = a
    pipe.params.deltam31.value = b
    t2 = pc()
    maps = pipe.get_outputs()
    t3 = pc()
    tot = sum(maps)
    t4 = pc()
    v = data.metric_total(expected_values=tot, metric="llh")
    t5 = pc()
    T.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0))
    return v


for _ in range(50):
    one()
T.clear()
for _ in range(500):
    one()
T = np.array(T) * 1e6
for name, col in zip(("quantities", "param setters", "get_outputs (plan: matrices + prob3 + accumulate launches)", "sum(maps)",
                      "metric_total (tail launch + wait)", "total"), T.T):
    print("%-60s median %7.2f us" % (name, np.median(col)))
