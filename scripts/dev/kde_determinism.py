"""The concurrent KDE stage (4 host threads / streams) gives bit-identical maps run after run and equal
to the single-thread order."""
import os, sys
from collections import OrderedDict
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pisa_amd.core.config_parser import parse_pipeline_config
from pisa_amd.core.pipeline import Pipeline
from pisa_amd.core.units import ureg

cfg = parse_pipeline_config("settings/pipeline/example_hip.cfg")
out = OrderedDict()
for k, v in cfg.items():
    out[("utils", "kde") if k == ("utils", "hist") else k] = (OrderedDict(calc_mode="events", apply_mode=v["apply_mode"])
                                                               if k == ("utils", "hist") else v)
out["pipeline"]["output_key"] = "weights"
out[("data", "synthetic_events")]["params"].params.n_events.value = float(sys.argv[1]) if len(sys.argv) > 1 else 2e6
pipe = Pipeline(out)
ref = None
for it in range(8):
    pipe.params.theta23.value = (44.0 + 1e-9 * (it % 2)) * ureg.degree     # forces a re-evaluation
    pipe.params.theta23.value = 44.0 * ureg.degree
    pipe["kde"].kde_workers = 1 if it == 7 else 4
    for s in pipe.stages:
        s.param_hash = None
    maps = [m.hist.copy() for m in pipe.get_outputs()]
    if ref is None:
        ref = maps
    same = all(np.array_equal(a, b) for a, b in zip(ref, maps))
    print("iteration %d workers %d identical %s total %.10g" % (it, pipe["kde"].kde_workers, same, sum(m.sum() for m in maps)))
    assert same
print("ok")
