"""Dump the sorted source arrays (whitened coordinates, coefficients, s2) and lattice geometry of a few C3 estimators
(development: input of scripts/dev/kde_pass_model.py, the CPU count of lattice-kernel passes per patch shape).
    python scripts/dev/kde_dump_estimator.py [n_events] [n_estimators] -> gpurun_out/kde_dump/est<i>.npz"""
import sys, os, json
from collections import OrderedDict
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pisa_amd import kernels as K
from pisa_amd.core.config_parser import parse_pipeline_config
from pisa_amd.core.pipeline import Pipeline
from pisa_amd.utils import kde_hist

n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e7
ne = int(sys.argv[2]) if len(sys.argv) > 2 else 3
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
stage = pipe["kde"]
captured = []
_orig = kde_hist.kde_histogramdd_batch
def _capture(samples, *a, **kw):
    for smp in samples:
        w = smp["weights"]() if callable(smp["weights"]) else smp["weights"]
        captured.append((w.clone(), dict(channels=smp["channels"])))
    return _orig(samples, *a, **kw)
kde_hist.kde_histogramdd_batch = _capture
pipe.get_outputs()
kde_hist.kde_histogramdd_batch = _orig
os.makedirs("gpurun_out/kde_dump", exist_ok=True)
i = 0
for w, kw in captured:
    pid_bin, d2d, chans = kw["channels"]
    g = kde_hist._evaluation_grid(d2d, stage.oversample, stage.coszen_name, stage.coszen_reflection)
    for idx, data in chans:
        if i >= ne:
            break
        x = data.T.clone()
        if g["cz_bin"] != 0:
            x[[0, g["cz_bin"]]] = x[[g["cz_bin"], 0]]
        est = K.KdeEstimator(x.contiguous(), torch.nan_to_num(w[idx]).contiguous(), bw_method=stage.bw_method,
                             adaptive=stage.adaptive, alpha=stage.alpha)
        axes = g["bin_points"]
        origin = [a[0] for a in axes]; step = [(a[-1] - a[0]) / (len(a) - 1) for a in axes]; count = [len(a) for a in axes]
        d = est.evaluate_lattice(origin, step, count)
        ys, coef, s2 = est.arrays()
        U = np.linalg.cholesky(est.inv_cov).T
        info = est.info() if hasattr(est, "info") else None
        np.savez_compressed("gpurun_out/kde_dump/est%d.npz" % i, ys=ys.cpu().numpy().astype(np.float32), s2=s2.cpu().numpy().astype(np.float32),
                            U=U, mean=np.asarray(est.mean if hasattr(est, "mean") else [0, 0]), origin=origin, step=step, count=count,
                            r_cut=est.r_cut if hasattr(est, "r_cut") else np.nan, pairs_eval=est.pairs_eval)
        print(json.dumps(dict(i=i, n=est.n, count=count, pairs_eval=est.pairs_eval)), flush=True)
        i += 1
