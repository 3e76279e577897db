"""C3-shaped pipeline at full size: 1e7 synthetic events, KDE stage on; timing per evaluation."""
import sys, time, json
from collections import OrderedDict
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from pisa_amd.core.config_parser import parse_pipeline_config
from pisa_amd.core.pipeline import Pipeline
from pisa_amd.core.units import ureg

n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e7
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-12     # (the bench leg's explicit cut-off; "0" / "1e-14": others)
cfg = parse_pipeline_config("settings/pipeline/example_hip.cfg")
out = OrderedDict()
for k, v in cfg.items():
    if k == ("utils", "hist"):
        out[("utils", "kde")] = OrderedDict(calc_mode="events", apply_mode=v["apply_mode"], tol=tol)
    else:
        out[k] = v
out["pipeline"]["output_key"] = "weights"
out[("data", "synthetic_events")]["params"].params.n_events.value = n
t0 = time.perf_counter()
pipe = Pipeline(out, profile=True)
print("setup %.2f s" % (time.perf_counter() - t0), flush=True)
times = []
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
    pipe.params.theta23.value = (42.0 + it) * ureg.degree
    torch.cuda.synchronize(); t0 = time.perf_counter()
    maps = pipe.get_outputs()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = pipe["kde"].stats
    times.append(dt)
    print(json.dumps(dict(it=it, s=dt, total=float(sum(m.hist.sum() for m in maps)), **st)), flush=True)
import hashlib
digest = hashlib.sha256(b"".join(np.ascontiguousarray(np.asarray(m.hist, dtype=np.float64)).tobytes() for m in maps)).hexdigest()[:16]
print(json.dumps(dict(median_ms=round(1e3 * float(np.median(times[1:])), 2), min_ms=round(1e3 * min(times[1:]), 2), n=len(times) - 1, tol=tol,
                      maps_sha_last=digest)))
pipe.report_profile()
