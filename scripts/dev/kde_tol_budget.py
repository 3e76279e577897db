"""The cut-off of the KDE against the parity budget (round-5 verdict item 1b): C3-shaped pipeline, maps for several `tol`
against the all-pairs evaluation (tol = 0), per BIN (the oversampled lattice folded, x volume, block-summed): the largest
relative difference over all bins of all 12 maps, and where the sparsest bin sits.
    python scripts/dev/kde_tol_budget.py [n_events]"""
import sys, time, json, os
from collections import OrderedDict
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pisa_amd.core.config_parser import parse_pipeline_config
from pisa_amd.core.pipeline import Pipeline

n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e7

def maps_for(tol):
    cfg = parse_pipeline_config("settings/pipeline/example_hip.cfg")
    out = OrderedDict()
    for k, v in cfg.items():
        if k == ("utils", "hist"):
            out[("utils", "kde")] = OrderedDict(calc_mode="events", apply_mode=v["apply_mode"], tol=tol)
        else:
            out[k] = v
    out["pipeline"]["output_key"] = "weights"
    out[("data", "synthetic_events")]["params"].params.n_events.value = n
    pipe = Pipeline(out)
    pipe.get_outputs()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m = pipe.get_outputs()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return np.stack([np.asarray(x.hist, dtype=np.float64) for x in m]), dt

ref, t_ref = maps_for(0.0)
print(json.dumps(dict(tol=0.0, ms=round(t_ref * 1e3, 2), bins=int(ref.size), min_bin=float(ref.min()), max_bin=float(ref.max()),
                      ratio_min_max=float(ref.min() / ref.max()))), flush=True)
for tol in (1e-16, 1e-14, 1e-13, 1e-12, 1e-11, 1e-10, 1e-9):
    got, dt = maps_for(tol)
    rel = np.abs(got - ref) / np.abs(ref)
    i = np.unravel_index(np.argmax(rel), rel.shape)
    print(json.dumps(dict(tol=tol, ms=round(dt * 1e3, 2), max_rel_per_bin=float(rel.max()), at_bin_value_over_max=float(ref[i] / ref.max()),
                          median_rel=float(np.median(rel)), total_rel=float(abs(got.sum() - ref.sum()) / ref.sum()))), flush=True)
