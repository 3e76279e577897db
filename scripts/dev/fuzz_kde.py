"""Randomised differential run of the device KDE against its CPU oracle (development tool, GPU box; the estimator's
parity with the external `kde` package is unpinned, this checks the DEVICE against THIS build's own restatement).
Every trial draws a dimension (1-3), a sample of 40 ... 30 000 sources from clouds of different shape (uniform + smeared,
skewed, correlated, clustered, with duplicates), weights (none, positive, some exactly zero), the bandwidth rule
(scott / silverman), fixed or adaptive bandwidths with alpha in [0.05, 0.6], the cut-off tolerance (1e-14 / 1e-12 / none);
compares the densities at random points (also outside the cloud) and -- in two dimensions -- on a lattice whose step is
anywhere between far finer and far coarser than the kernels with `kde_oracle.gaussian_kde_eval`, and the lattice
evaluation twice for the same bits.    usage: fuzz_kde.py [trials] [seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import kde_oracle  # noqa: E402
from pisa_amd import kernels as K  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
only = int(sys.argv[3]) if len(sys.argv) > 3 else None      # replay ONE trial of the sequence, with diagnostics
bad = 0
t0 = time.time()
for trial in range(trials):
    dim = int(rs.randint(1, 4))
    n = int(10 ** rs.uniform(1.6, 4.5))
    shape = ["box", "skewed", "correlated", "clusters"][rs.randint(4)]
    x = np.empty((dim, n))
    if shape == "box":
        x[:] = rs.rand(dim, n) * 2 - 1 + rs.randn(dim, n) * 0.05
    elif shape == "skewed":
        x[:] = rs.gamma(rs.uniform(1.5, 5.0), rs.uniform(0.2, 1.0), (dim, n))
    elif shape == "correlated":
        base = rs.randn(n)
        for d in range(dim):
            x[d] = base * rs.uniform(-0.9, 0.9) + rs.randn(n) * rs.uniform(0.2, 1.0) + d
    else:
        centres = rs.uniform(-3, 3, (dim, 4))
        which = rs.randint(4, size=n)
        x[:] = centres[:, which] + rs.randn(dim, n) * rs.uniform(0.05, 0.6, (dim, 1))
    if rs.rand() < 0.2:                     # duplicates
        k = max(1, n // 10)
        x[:, :k] = x[:, n - k:]
    wmode = ["none", "positive", "zeros"][rs.randint(3)]
    w = None
    if wmode != "none":
        w = rs.rand(n) * 2 + 0.05
        if wmode == "zeros":
            w[rs.rand(n) < 0.3] = 0.0
            if w.sum() == 0:
                w[0] = 1.0
    bw = ["scott", "silverman"][rs.randint(2)]
    adaptive = rs.rand() < 0.75
    alpha = float(rs.uniform(0.05, 0.6))
    tol = [1e-14, 1e-12, 0.0][rs.randint(3)] if n <= 8000 else 1e-14
    m = int(rs.randint(50, 1500))
    lo, hi = x.min(axis=1, keepdims=True), x.max(axis=1, keepdims=True)
    q = lo + (hi - lo) * (rs.rand(dim, m) * 1.3 - 0.15)
    tag = "trial %d: dim %d, n %d, %s, weights %s, %s, adaptive %s alpha %.3f, tol %g" % (trial, dim, n, shape, wmode, bw, adaptive, alpha, tol)
    if dim == 2:
        counts = (int(rs.randint(2, 160)), int(rs.randint(2, 160)))
        span = 10 ** rs.uniform(-0.5, 0.5)
    if only is not None and trial != only:
        continue
    try:
        est = K.KdeEstimator(K.to_device(x), None if w is None else K.to_device(w), bw_method=bw, adaptive=adaptive,
                             alpha=alpha, tol=tol)
        got = est(K.to_device(q)).cpu().numpy()
        want = kde_oracle.gaussian_kde_eval(x, w, q, bw, adaptive, alpha)
        floor = max(tol, 1e-16) * 100 * max(want.max(), 1e-300)
        ok = np.allclose(got, want, rtol=1e-9, atol=floor) and np.all(np.isfinite(got))
        lat_ok = same = True
        if dim == 2:
            mid, half = 0.5 * (lo + hi).ravel(), 0.5 * (hi - lo).ravel() * span
            origin = (mid - half).tolist()
            step = [2 * half[0] / max(counts[0] - 1, 1), 2 * half[1] / max(counts[1] - 1, 1)]
            lat = est.evaluate_lattice(origin, step, counts).cpu().numpy()
            pts = np.array([g.ravel() for g in np.meshgrid(origin[0] + step[0] * np.arange(counts[0]),
                                                           origin[1] + step[1] * np.arange(counts[1]), indexing="ij")])
            ref = kde_oracle.gaussian_kde_eval(x, w, pts, bw, adaptive, alpha)
            # the cut-off drops what is below `tol` of a kernel's peak: the floor is relative to the density INSIDE the
            # cloud (`want`: random points in and around it), not to the largest value of a lattice that may lie outside
            lat_ok = np.allclose(lat.ravel(), ref, rtol=1e-9, atol=max(tol, 1e-16) * 100 * max(ref.max(), want.max(), 1e-300)) \
                and np.all(np.isfinite(lat))
            same = np.array_equal(lat, est.evaluate_lattice(origin, step, counts).cpu().numpy())
            if only is not None:
                err = np.abs(lat.ravel() - ref)
                k = int(np.argmax(err / np.maximum(np.abs(ref), max(tol, 1e-16) * 100 * ref.max())))
                direct = est(K.to_device(pts)).cpu().numpy()
                print("lattice", counts, "step", step, "span %.3f" % span, "ref max %.3e" % ref.max(), "worst at", k, "lattice %.6e ref %.6e direct %.6e"
                      % (lat.ravel()[k], ref[k], direct[k]), "max abs err %.3e" % err.max(), "direct-vs-ref max abs %.3e" % np.abs(direct - ref).max(),
                      "pairs", est.pairs_eval)
        if not (ok and lat_ok and same):
            bad += 1
            worst = float(np.max(np.abs(got - want) / np.maximum(np.abs(want), floor)))
            print("MISMATCH", tag, "points", ok, "(worst %.2e)" % worst, "lattice", lat_ok, "same bits", same, flush=True)
        del est
    except Exception as e:  # pylint: disable=broad-except
        bad += 1
        print("ERROR", tag, type(e).__name__, str(e)[:300], flush=True)
    if trial % 20 == 19:
        print("... %d trials, %d bad, %.0f s" % (trial + 1, bad, time.time() - t0), flush=True)
print("fuzz_kde: %d trials, %d bad" % (trials, bad))
sys.exit(1 if bad else 0)
