"""Development probe: per-evaluation wall time of the reference-style Pipeline objects
(osc_example.cfg; IceCube_3y_neutrinos.cfg on a synthetic MC file) with one osc parameter changed."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tmp = tempfile.mkdtemp()
n = sys.argv[1] if len(sys.argv) > 1 else "1e6"
subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "make_synthetic_3y_mc.py"), tmp, n, "0"])
os.environ["PISA_RESOURCES"] = tmp
from pisa_amd.core.pipeline import Pipeline  # noqa: E402
from pisa_amd.core.units import ureg  # noqa: E402

for cfg in ("settings/pipeline/osc_example.cfg", "settings/pipeline/IceCube_3y_neutrinos.cfg"):
    t0 = time.perf_counter()
    pipe = Pipeline(cfg)
    pipe.get_outputs()
    t_setup = time.perf_counter() - t0
    rs = np.random.RandomState(0)
    ts = []
    for _ in range(30):
        pipe.params.theta23.value = (40 + 10 * rs.rand()) * ureg.degree
        t0 = time.perf_counter()
        pipe.get_outputs()
        ts.append(time.perf_counter() - t0)
    print("%s: setup+first eval %.2f s; per eval mean %.3f ms, min %.3f ms" % (cfg, t_setup, 1e3 * np.mean(ts[5:]), 1e3 * np.min(ts)))
    if "3y" in cfg:
        import cProfile
        import pstats

        pr = cProfile.Profile()
        pr.enable()
        for _ in range(20):
            pipe.params.theta23.value = (40 + 10 * rs.rand()) * ureg.degree
            pipe.get_outputs()
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
