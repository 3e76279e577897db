"""Facts about the estimators of one C3 evaluation (1e7 events): grid, dense cells, pairs, per-call time on ONE stream.
    python scripts/dev/kde_facts.py [n_events] [n_containers]"""
import sys, time, json
from collections import OrderedDict
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from pisa_amd import kernels as K
from pisa_amd.core.config_parser import parse_pipeline_config
from pisa_amd.core.pipeline import Pipeline
from pisa_amd.utils import kde_hist

n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e7
nc = int(sys.argv[2]) if len(sys.argv) > 2 else 12
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
def _capture(samples, *a, **kw):   # the event weights as the stage hands them over
    for smp in samples:
        w = smp["weights"]() if callable(smp["weights"]) else smp["weights"]
        captured.append((w.clone(), dict(channels=smp["channels"])))
    return _orig(samples, *a, **kw)
kde_hist.kde_histogramdd_batch = _capture
pipe.get_outputs()
kde_hist.kde_histogramdd_batch = _orig
jobs = []
for w, kw in captured[:nc]:
    pid_bin, d2d, chans = kw["channels"]
    g = kde_hist._evaluation_grid(d2d, stage.oversample, stage.coszen_name, stage.coszen_reflection)
    for idx, data in chans:
        x = data.T.clone()
        if g["cz_bin"] != 0:
            x[[0, g["cz_bin"]]] = x[[g["cz_bin"], 0]]
        jobs.append(("c%d" % len(jobs), x.contiguous(), torch.nan_to_num(w[idx]).contiguous(), g))
torch.cuda.synchronize()
tot = [0.0, 0.0]
for rep in range(3):
    for name, x, w, g in jobs:
        axes = g["bin_points"]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        est = K.KdeEstimator(x, w, bw_method=stage.bw_method, adaptive=stage.adaptive, alpha=stage.alpha, tol=stage.tol)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        d = est.evaluate_lattice([a[0] for a in axes], [(a[-1] - a[0]) / (len(a) - 1) for a in axes], [len(a) for a in axes])
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if rep == 2:
            tot[0] += t1 - t0; tot[1] += t2 - t1
            ys, coef, s2 = est.arrays()
            U = np.linalg.cholesky(est.inv_cov).T
            step = [(a[-1] - a[0]) / (len(a) - 1) for a in axes]
            print(json.dumps(dict(c=name, n=est.n, create_ms=round((t1 - t0) * 1e3, 3), lattice_ms=round((t2 - t1) * 1e3, 3),
                                  n_cells=est.n_cells, n_dense=est.n_dense, cell=round(est.cell, 3),
                                  pairs_pilot=est.pairs_pilot, pilot_per_src=round(est.pairs_pilot / est.n, 1),
                                  pairs_eval=est.pairs_eval, eval_per_src=round(est.pairs_eval / est.n, 1),
                                  count=[len(a) for a in axes], da=round(U[0, 0] * step[0], 4), sa=round(U[0, 1] * step[1], 4), db=round(U[1, 1] * step[1], 4),
                                  s2=[round(float(s2.min()), 3), round(float(s2.max()), 3)], checksum=float(d.sum()))), flush=True)
            yy = ys.cpu().numpy()
            cid = (np.floor((yy[0] - yy[0].min()) / est.cell).astype(np.int64) * 100000
                   + np.floor((yy[1] - yy[1].min()) / est.cell).astype(np.int64))
            pop = np.sort(np.unique(cid, return_counts=True)[1])[::-1]
            print("   sources per non-empty cell: max %d, top five %s, median %d; cells with > 512: %d, holding %.0f %% of the sources"
                  % (pop[0], pop[:5].tolist(), int(np.median(pop)), int((pop > 512).sum()), 100.0 * pop[pop > 512].sum() / pop.sum()), flush=True)
print(json.dumps(dict(estimators=len(jobs), create_ms=round(tot[0] * 1e3, 2), lattice_ms=round(tot[1] * 1e3, 2))))
