"""inside FastPlan.run at the headline size: medians of the pieces on the host (wrapped with timers)"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from pisa_amd.core import fastplan  # noqa: E402
from pisa_amd.core.pipeline import Pipeline  # noqa: E402
from pisa_amd.core.units import ureg  # noqa: E402

pipe = Pipeline(bench._pipeline_cfg(1e7))
data = sum(pipe.get_outputs()).fluctuate("poisson", random_state=0)
pipe.get_outputs()
plan = pipe._plan
acc = {}
pc = time.perf_counter


def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name

    def w(*a, **k):
        t0 = pc()
        r = fn(*a, **k)
        acc.setdefault(label, []).append(pc() - t0)
        return r

    setattr(obj, name, w)


wrap(plan, "_writes")
wrap(plan, "_changed")
wrap(plan.osc, "_matrices")
wrap(plan.engine, "front")
lib = plan._lib
orig = lib.pisa_hip_prob3_grid_planned


def planned(*a):
    t0 = pc()
    r = orig(*a)
    acc.setdefault("ctypes prob3_grid_planned", []).append(pc() - t0)
    return r


class LibProxy:
    def __getattr__(self, k):
        return planned if k == "pisa_hip_prob3_grid_planned" else getattr(lib, k)


plan._lib = LibProxy()
wrap(plan, "run")
rs = np.random.RandomState(1)
for i in range(550):
    pipe.params.theta23.value = (31.0 + 28.0 * rs.rand()) * ureg.degree
    pipe.params.deltam31.value = (1e-3 + 6e-3 * rs.rand()) * ureg.eV ** 2
    v = data.metric_total(expected_values=sum(pipe.get_outputs()), metric="llh")
    if i == 49:
        acc.clear()
for k, v in acc.items():
    print("%-32s median %6.2f us" % (k, np.median(v) * 1e6))
