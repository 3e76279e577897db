"""`pisa_amd.core.translation` (free functions on the GPU) against plain numpy restatements of the reference's two regimes
(development tool, GPU box).  Every trial: a binning of 1-3 dimensions, each linear-regular, log-regular or irregular;
a sample with values on edges, outside, NaN, +-inf; weights scalar or [N, d]; host arrays or device tensors.
  * all dimensions linear and regular -> fast_histogram's rule (half-open range; bin = int((x - min) * n / (max - min)));
  * otherwise -> numpy's `histogramdd` (edges compared, last edge included) for ALL dimensions;
`lookup` with the corresponding rule, `histogram(..., averaged=True)`, counts, `resample` between a binning and its
down-sampled version.   usage: fuzz_translation.py [trials] [seed]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from pisa_amd.core.binning import MultiDimBinning, OneDimBinning  # noqa: E402
from pisa_amd.core.translation import find_index, histogram, lookup  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for trial in range(trials):
    ndim = int(rs.randint(1, 4))
    dims, kinds = [], []
    for d in range(ndim):
        kind = ["lin", "log", "irr"][rs.choice(3, p=[0.6, 0.2, 0.2])]
        n = int(rs.randint(1, 12 if ndim == 3 else 40))
        lo = rs.uniform(0.1, 10)
        if kind == "lin":
            dims.append(OneDimBinning("d%d" % d, num_bins=n, domain=[lo - 5, lo + rs.uniform(0.5, 20)], is_lin=True))
        elif kind == "log":
            dims.append(OneDimBinning("d%d" % d, num_bins=n, domain=[lo, lo * rs.uniform(1.5, 100)], is_log=True))
        else:
            dims.append(OneDimBinning("d%d" % d, bin_edges=lo + np.concatenate([[0], np.cumsum(rs.uniform(0.1, 3, n))])))
        kinds.append(kind)
    b = MultiDimBinning(dims)
    n_ev = int(10 ** rs.uniform(0, 4.3))
    sample = []
    for d in dims:
        e = d.edge_magnitudes
        x = rs.uniform(e[0] - 0.2 * (e[-1] - e[0]), e[-1] + 0.2 * (e[-1] - e[0]), n_ev)
        k = rs.rand(n_ev)
        x = np.where(k < 0.06, e[rs.randint(0, len(e), n_ev)], x)
        x = np.where((k > 0.06) & (k < 0.07), np.nan, x)
        x = np.where((k > 0.07) & (k < 0.075), np.inf, x)
        sample.append(x)
    width = int(rs.randint(1, 4))
    w = rs.rand(n_ev) if width == 1 else rs.rand(n_ev, width)
    fast = b.is_lin and not b.is_irregular      # the reference's switch (translation.py:110-116); a ONE-bin dimension is regular
    # ---- numpy restatement
    if fast:
        idx, inside = [], np.ones(n_ev, dtype=bool)
        for d, x in zip(dims, sample):
            lo, hi = d.edge_magnitudes[0], d.edge_magnitudes[-1]
            with np.errstate(invalid="ignore"):
                ok = (x >= lo) & (x < hi)
                i = np.where(ok, ((np.where(ok, x, lo) - lo) * (d.num_bins / (hi - lo))).astype(np.int64), 0)
            idx.append(np.minimum(i, d.num_bins - 1))
            inside &= ok
    else:
        idx, inside = [], np.ones(n_ev, dtype=bool)
        for d, x in zip(dims, sample):
            i = find_index(x, d.edge_magnitudes)
            ok = (i >= 0) & (i < d.num_bins)
            idx.append(np.clip(i, 0, d.num_bins - 1))
            inside &= ok
    flat = np.ravel_multi_index([i[inside] for i in idx], b.shape) if n_ev else np.zeros(0, dtype=np.int64)

    def hist_of(weights):
        return np.bincount(flat, weights=None if weights is None else weights[inside], minlength=b.size).astype(float)

    want = hist_of(w) if width == 1 else np.stack([hist_of(w[:, j]) for j in range(width)], axis=1)
    counts = hist_of(None)
    problems = []
    try:
        dev = rs.rand() < 0.4
        s_in = [torch.as_tensor(x, device="cuda") for x in sample] if dev else (np.stack(sample, axis=1) if rs.rand() < 0.3 and ndim > 1 else sample)
        w_in = torch.as_tensor(w, device="cuda") if dev else w
        got = histogram(s_in, w_in, b, averaged=False)
        got = got.cpu().numpy() if dev else got
        if got.shape != want.shape or not np.allclose(got, want, rtol=1e-12, atol=1e-13 * max(np.abs(want).max(), 1e-300)):
            problems.append("histogram (%s regime)" % ("fast_histogram" if fast else "numpy"))
        if not np.array_equal(histogram(sample, None, b, averaged=False), counts):
            problems.append("counts")
        with np.errstate(divide="ignore", invalid="ignore"):
            avg = np.nan_to_num(want / (counts if width == 1 else counts[:, None]))
        got_avg = histogram(sample, w, b, averaged=True)
        if not np.allclose(got_avg, avg, rtol=1e-12, atol=1e-13 * max(np.abs(avg).max(), 1e-300)):
            problems.append("averaged")
        values = rs.randn(b.size) if width == 1 else rs.randn(b.size, width)
        lk_want = np.where(inside if width == 1 else inside[:, None], values[np.ravel_multi_index(idx, b.shape)], 0.0)
        lk = lookup(s_in if not isinstance(s_in, np.ndarray) else sample, torch.as_tensor(values, device="cuda") if dev else values, b)
        lk = lk.cpu().numpy() if dev else lk
        if not np.array_equal(lk, lk_want):
            problems.append("lookup: %d differ" % np.count_nonzero(lk != lk_want))
    except Exception as e:  # pylint: disable=broad-except
        problems.append("%s %s" % (type(e).__name__, str(e)[:200]))
    if problems:
        bad += 1
        print("MISMATCH trial %d: dims %s %s, n %d, width %d | %s" % (trial, kinds, list(b.shape), n_ev, width, "; ".join(problems)), flush=True)
print("fuzz_translation: %d trials, %d bad" % (trials, bad))
sys.exit(1 if bad else 0)
