"""when, inside one C3 evaluation, the KDE stage hands its jobs to the library: time of every `submit`, of `wait` (start, end)
and of the end of `get_outputs`, medians over the evaluations.  usage: kde_submit_times.py [events] [evaluations]"""
import sys, time, json
from collections import OrderedDict
import numpy as np, torch
sys.path.insert(0, ".")
from pisa_amd.core.config_parser import parse_pipeline_config
from pisa_amd.core.pipeline import Pipeline
from pisa_amd.core.units import ureg
from pisa_amd import kernels as K

n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e7
cfg = parse_pipeline_config("settings/pipeline/example_hip.cfg")
out = OrderedDict()
for k, v in cfg.items():
    if k == ("utils", "hist"):
        out[("utils", "kde")] = OrderedDict(calc_mode="events", apply_mode=v["apply_mode"])
    else:
        out[k] = v
out["pipeline"]["output_key"] = "weights"
out[("data", "synthetic_events")]["params"].params.n_events.value = n
pipe = Pipeline(out)
T = []
pc = time.perf_counter
sub0, wait0 = K.KdeLatticeBatch.submit, K.KdeLatticeBatch.wait


def submit(self, jobs):
    a = pc(); r = sub0(self, jobs); T.append(("submit", a, pc())); return r


def wait(self):
    a = pc(); r = wait0(self); T.append(("wait", a, pc())); return r


K.KdeLatticeBatch.submit, K.KdeLatticeBatch.wait = submit, wait
rows = []
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    pipe.params.theta23.value = (42.0 + it) * ureg.degree
    torch.cuda.synchronize(); T.clear(); t0 = pc()
    pipe.get_outputs()
    torch.cuda.synchronize(); t1 = pc()
    subs = [x for x in T if x[0] == "submit"]
    w = [x for x in T if x[0] == "wait"][0]
    rows.append([1e3 * (s[1] - t0) for s in subs] + [1e3 * (subs[-1][2] - t0), 1e3 * (w[1] - t0), 1e3 * (w[2] - t0), 1e3 * (t1 - t0)])
rows = np.median(np.array(rows[2:]), axis=0)
print("submits start at (ms):", " ".join("%.2f" % v for v in rows[:-4]))
print("last submit returns %.2f, wait starts %.2f, wait ends %.2f, evaluation ends %.2f ms" % tuple(rows[-4:]))
