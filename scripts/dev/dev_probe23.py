"""cProfile (own time) of Pipeline.get_outputs() for IceCube_3y_neutrinos.cfg on a synthetic MC file."""
import cProfile
import os
import pstats
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tmp = tempfile.mkdtemp()
subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "make_synthetic_3y_mc.py"), tmp, "1e6", "0"],
                      stdout=subprocess.DEVNULL)
os.environ["PISA_RESOURCES"] = tmp
from pisa_amd.core.pipeline import Pipeline  # noqa: E402
from pisa_amd.core.units import ureg  # noqa: E402

pipe = Pipeline("settings/pipeline/IceCube_3y_neutrinos.cfg")
pipe.get_outputs()
rs = np.random.RandomState(0)
ts = []
for _ in range(100):
    pipe.params.theta23.value = (40 + 10 * rs.rand()) * ureg.degree
    t0 = time.perf_counter()
    pipe.get_outputs()
    ts.append(time.perf_counter() - t0)
print("per eval mean %.3f ms, min %.3f ms" % (1e3 * np.mean(ts[10:]), 1e3 * np.min(ts)))
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    pipe.params.theta23.value = (40 + 10 * rs.rand()) * ureg.degree
    pipe.get_outputs()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
