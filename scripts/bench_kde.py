#!/usr/bin/env python
"""Timing of the KDE estimator (`pisa_hip_kde_create/evaluate`) on one GPU: pilot + adaptive
evaluation on a C3-shaped channel (2-D reco_coszen x ln reco_energy, 80 x 120 evaluation points).

    python scripts/bench_kde.py [N ...]      # prints one JSON line per N
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pisa_amd import kernels as K  # noqa: E402


def main():
    sizes = [int(float(a)) for a in sys.argv[1:]] or [100000, 400000]
    tol = float(os.environ.get("KDE_TOL", K.KDE_DEFAULT_TOL))
    for n in sizes:
        rs = np.random.RandomState(0)
        cz = np.clip(rs.rand(n) * 2 - 1 + rs.randn(n) * 0.15, -1, 1)
        le = np.log(10 ** (rs.rand(n) * 3) * np.exp(rs.randn(n) * 0.2))
        x = K.to_device(np.stack([cz, le]))
        w = K.to_device(rs.rand(n) + 0.1)
        gq = np.array([g.ravel() for g in np.meshgrid(np.linspace(-1.5, 1.5, 120), np.linspace(np.log(5.0), np.log(100.0), 80), indexing="ij")])
        q = K.to_device(gq)
        res = dict(n=n, m=q.shape[1], tol=tol)
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            est = K.KdeEstimator(x, w, adaptive=True, alpha=0.1, tol=tol)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            out = est(q)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        res.update(create_ms=(t1 - t0) * 1e3, eval_ms=(t2 - t1) * 1e3, pairs_pilot=est.pairs_pilot,
                   pairs_eval=est.pairs_eval, frac_pilot=est.pairs_pilot / float(n) ** 2,
                   frac_eval=est.pairs_eval / float(n * q.shape[1]),
                   pilot_pairs_per_s=est.pairs_pilot / (t1 - t0), eval_pairs_per_s=est.pairs_eval / (t2 - t1),
                   n_cells=est.n_cells, factor=est.factor, checksum=float(out.sum()))
        if n <= 200000:
            t0 = time.perf_counter()
            est0 = K.KdeEstimator(x, w, adaptive=True, alpha=0.1, tol=0.0)
            out0 = est0(q)
            torch.cuda.synchronize()
            res.update(allpairs_ms=(time.perf_counter() - t0) * 1e3,
                       max_rel_diff=float(((out - out0).abs() / out0.abs().max()).max()))
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
