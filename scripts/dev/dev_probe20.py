"""cProfile of Pipeline.get_outputs() (osc_example.cfg) by own time."""
import cProfile
import pstats
import time

import numpy as np

from pisa_amd.core.pipeline import Pipeline
from pisa_amd.core.units import ureg

pipe = Pipeline("settings/pipeline/osc_example.cfg")
pipe.get_outputs()
rs = np.random.RandomState(0)
ts = []
for _ in range(200):
    pipe.params.theta23.value = (40 + 10 * rs.rand()) * ureg.degree
    t0 = time.perf_counter()
    pipe.get_outputs()
    ts.append(time.perf_counter() - t0)
print("per eval mean %.3f ms, min %.3f ms" % (1e3 * np.mean(ts[20:]), 1e3 * np.min(ts)))
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    pipe.params.theta23.value = (40 + 10 * rs.rand()) * ureg.degree
    pipe.get_outputs()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(30)
