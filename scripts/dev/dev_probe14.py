"""Development probe: per-evaluation time of the example pipeline with utils.kde instead of utils.hist."""
import sys
import time
from collections import OrderedDict

import numpy as np

from pisa_amd.core.config_parser import parse_pipeline_config
from pisa_amd.core.pipeline import Pipeline
from pisa_amd.core.units import ureg

for n in [float(x) for x in (sys.argv[1:] or ["1.2e5", "1.2e6"])]:
    cfg2 = parse_pipeline_config("settings/pipeline/example_hip.cfg")
    cfg = OrderedDict()
    for k, v in cfg2.items():
        if k == ("utils", "hist"):
            cfg[("utils", "kde")] = OrderedDict(calc_mode="events", apply_mode=v["apply_mode"], bw_method="silverman",
                                                 alpha=0.3, oversample=10, coszen_reflection=0.5, adaptive=True)
        else:
            cfg[k] = v
    cfg["pipeline"]["output_key"] = "weights"
    cfg[("data", "synthetic_events")]["params"].params.n_events.value = n
    t0 = time.perf_counter()
    pipe = Pipeline(cfg)
    pipe.get_outputs()
    t_first = time.perf_counter() - t0
    rs = np.random.RandomState(0)
    ts = []
    for _ in range(3):
        pipe.params.theta23.value = (40 + 10 * rs.rand()) * ureg.degree
        t0 = time.perf_counter()
        pipe.get_outputs()
        ts.append(time.perf_counter() - t0)
    print("utils.kde, %d events: first %.2f s, per evaluation %.1f ms" % (n, t_first, 1e3 * np.mean(ts)), flush=True)
