import cProfile, pstats, sys, io
from collections import OrderedDict
sys.path.insert(0, "/root/repo")
import torch
from pisa_amd.core.config_parser import parse_pipeline_config
from pisa_amd.core.pipeline import Pipeline
from pisa_amd.core.units import ureg
cfg = parse_pipeline_config("settings/pipeline/example_hip.cfg")
out = OrderedDict()
for k, v in cfg.items():
    if k == ("utils", "hist"):
        out[("utils", "kde")] = OrderedDict(calc_mode="events", apply_mode=v["apply_mode"])
    else:
        out[k] = v
out["pipeline"]["output_key"] = "weights"
out[("data", "synthetic_events")]["params"].params.n_events.value = 1e7
pipe = Pipeline(out)
pipe.get_outputs()
pipe.params.theta23.value = 43.0 * ureg.degree
pipe.get_outputs()
pipe.params.theta23.value = 44.0 * ureg.degree
pr = cProfile.Profile()
pr.enable()
pipe.get_outputs()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue())
