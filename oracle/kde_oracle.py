"""CPU oracle of the KDE map chain (TEST INFRASTRUCTURE ONLY).

Restates pisa/utils/kde_hist.py:35-387 (wrapper: oversampling, coszen
reflection, bin volumes, pid stacking) on plain numpy edge arrays, with the
density estimator of the un-vendored `kde` package restated from its call
contract (PARITY UNPINNED: see oracle/pisa_oracle.c, KDE section -- except the fixed-bandwidth
unweighted estimator, which tests/test_oracle.py checks against scipy.stats.gaussian_kde).  Binnings are
passed as lists of (name, edges, is_log) so that nothing of the product's
binning classes is needed.
"""
import numpy as np

from . import oracle as orc


def _oversample(edges, factor, is_log):
    parts = []
    for lo, hi in zip(edges[:-1], edges[1:]):
        sub = np.logspace(np.log10(lo), np.log10(hi), factor + 1) if is_log else np.linspace(lo, hi, factor + 1)
        sub[0], sub[-1] = lo, hi
        parts.append(sub[:-1])
    parts.append(np.array([edges[-1]]))
    return np.concatenate(parts)


def _centers(edges, is_log):
    return np.sqrt(edges[:-1] * edges[1:]) if is_log else 0.5 * (edges[:-1] + edges[1:])


def gaussian_kde_eval(x, weights, points, bw_method, adaptive, alpha):
    d, n = x.shape
    w = np.full(n, 1.0 / n) if weights is None else np.asarray(weights, dtype=float) / np.sum(weights)
    factor = (n * (d + 2) / 4.0) ** (-1.0 / (d + 4)) if bw_method == "silverman" else n ** (-1.0 / (d + 4))
    mean = (x * w).sum(axis=1, keepdims=True)
    xc = x - mean
    cov = (xc * w) @ xc.T / (1.0 - np.sum(w * w))
    covh = cov * factor ** 2
    inv_cov = np.linalg.inv(covh)
    norm = np.sqrt(np.linalg.det(2 * np.pi * covh))
    ones = np.ones(n)
    if adaptive:
        pilot = orc.kde_eval(x, w / norm, ones, x, inv_cov)
        # a source of weight zero contributes nothing: it is left out of the geometric mean and keeps the global
        # bandwidth (the choice of pisa_amd/csrc/kde.hip, kde_logsum_kernel; its pilot density would otherwise make
        # the estimate depend on where a cut-off sets it to exactly 0)
        pos = (w > 0) & (pilot > 0)
        glob = np.exp(np.mean(np.log(pilot[pos])))
        s = np.where(pos, (np.where(pos, pilot, 1.0) / glob) ** alpha, 1.0)
    else:
        s = ones
    return orc.kde_eval(x, w * s ** d / norm, s * s, points, inv_cov)


def get_hist(sample, dims, weights, bw_method, adaptive, alpha, coszen_reflection, coszen_name, oversample):
    norm = sample.shape[0] if weights is None else np.sum(np.nan_to_num(weights))
    dims = [(n, _oversample(np.asarray(e, dtype=float), oversample, lg), lg) for n, e, lg in dims]
    x = np.array(sample.T)
    names = [n for n, _, _ in dims]
    cz = names.index(coszen_name)
    if cz != 0:
        dims = [dims[cz]] + [d for d in dims if d[0] != coszen_name]
        x[[0, cz]] = x[[cz, 0]]
    edges = dims[0][1]
    lower, upper = edges[0] == -1, edges[-1] == 1
    pts, l = [], 0
    for name, e, lg in dims:
        c = _centers(e, lg)
        if name == coszen_name:
            l = int(len(c) * float(coszen_reflection))
            c0 = 2 * c[0] - c[1: l + 1][::-1] if lower else []
            c1 = 2 * c[-1] - c[-l - 1: -1][::-1] if upper else []
            c = np.concatenate([c0, c, c1])
        pts.append(c)
    shape = tuple(len(e) - 1 for _, e, _ in dims)
    mega = (shape[0] + (int(upper) + int(lower)) * l, shape[1])
    mini = (shape[0] - l, shape[1])
    grid = np.meshgrid(*pts, indexing="ij")
    points = np.array([g.ravel() for g in grid])
    w = None if weights is None else np.nan_to_num(weights)
    hist = gaussian_kde_eval(x, w, points, bw_method, adaptive, alpha).reshape(mega)
    h0 = h1 = 0
    if lower:
        h0 = np.flipud(np.concatenate([np.zeros(mini), hist[0:l, :]]))
        hist = hist[l:, :]
    if upper:
        h1 = np.flipud(np.concatenate([hist[-l:, :], np.zeros(mini)]))
        hist = hist[:-l, :]
    hist = hist + h1 + h0
    vol = np.multiply.outer(np.abs(np.diff(dims[0][1])), np.abs(np.diff(dims[1][1])))
    hist = hist * vol
    if oversample != 1:
        for i, (_, e, _) in enumerate(dims):
            hist = np.add.reduceat(hist, np.arange(0, len(e) - 1, oversample), axis=i)
    if cz != 0:
        hist = np.swapaxes(hist, 0, cz)
    return hist * norm


def kde_histogramdd(sample, dims, weights, bw_method="silverman", adaptive=True, alpha=0.1,
                    coszen_reflection=0.25, coszen_name="reco_coszen", oversample=10):
    """stack_pid=True form (kde_hist.py:303-372); dims = [(name, edges, is_log)] x 3 incl. 'pid'"""
    names = [n for n, _, _ in dims]
    pid_bin = names.index("pid")
    other = [0, 1, 2]
    other.pop(pid_bin)
    pid_edges = np.asarray(dims[pid_bin][1], dtype=float)
    d2d = [d for d in dims if d[0] != "pid"]
    stack = []
    for p in range(len(pid_edges) - 1):
        mask = (sample.T[pid_bin] >= pid_edges[p]) & (sample.T[pid_bin] < pid_edges[p + 1])
        data = np.array([sample.T[other[0]][mask], sample.T[other[1]][mask]])
        stack.append(get_hist(data.T, d2d, None if weights is None else weights[mask], bw_method,
                              adaptive, alpha, coszen_reflection, coszen_name, oversample))
    hist = np.dstack(stack)
    if pid_bin != 2:
        hist = np.swapaxes(hist, pid_bin, 2)
    return hist
