"""KDE-smoothed histograms (counterpart of pisa/utils/kde_hist.py:35-387).

`get_hist` / `kde_histogramdd` restate the reference's wrapper: oversampled bin
centres, coszen reflection at -1 / +1 (:122-190), multiplication by the bin
volumes, block-summing of the oversampling (:202-206), pid stacking (:303-372).

The density estimator itself (`gaussian_kde`) replaces the EXTERNAL `kde`
package (setup.py:88; call contract visible at kde_hist.py:110-120:
`k = gaussian_kde(x[D,N], weights=, bw_method=, adaptive=, alpha=)`,
`k(points[D,M]) -> density[M]` normalised to 1).  Its source is not part of the
reference tree, so the numerical details below are this build's choices
(PARITY UNPINNED): weighted full-covariance Gaussian
kernel, Silverman / Scott bandwidth factor on the sample count, Abramson
adaptive bandwidths lambda_i = (f(x_i)/g)^-alpha with g the geometric mean of
the fixed-bandwidth pilot densities.  The whole estimator -- moments, bandwidth
matrix, pilot, local bandwidths, evaluation -- is native code behind
`pisa_hip_kde_create/evaluate` (`csrc/kde.hip`): a cell list with a Gaussian
cut-off at kernel value `tol` (default 1e-14) makes the pilot and the
evaluation O(N k) instead of O(N^2) / O(N M).
"""
import copy

import numpy as np
import torch

from pisa_amd import kernels as K
from pisa_amd.core.binning import MultiDimBinning

__all__ = ["gaussian_kde", "bootstrap_kde", "get_hist", "kde_histogramdd", "kde_histogramdd_batch", "pid_channels"]


class gaussian_kde:  # pylint: disable=invalid-name
    """Call contract of `kde.gaussian_kde` (kde_hist.py:110-120).  `tol`: kernel values below it
    are dropped (cell-list cut-off, `csrc/kde.hip`); 0 evaluates all pairs."""

    def __init__(self, dataset, weights=None, bw_method="silverman", adaptive=True, alpha=0.3,
                 use_cuda=False, tol=None):  # pylint: disable=unused-argument
        x = dataset if torch.is_tensor(dataset) else K.to_device(np.atleast_2d(dataset))
        if weights is None or len(weights) == 0:
            w = None
        else:
            w = weights if torch.is_tensor(weights) else K.to_device(np.asarray(weights))
        self.d, self.n = x.shape
        self._est = K.KdeEstimator(x, w, bw_method=bw_method, adaptive=adaptive, alpha=alpha,
                                   tol=K.KDE_DEFAULT_TOL if tol is None else tol)
        self.factor = self._est.factor
        self.sum_w = self._est.sum_w
        self.covariance = self._est.covariance
        self.inv_cov = self._est.inv_cov
        self._norm = self._est.norm

    def __call__(self, points):
        q = points if torch.is_tensor(points) else K.to_device(np.atleast_2d(points))
        return self._est(q)

    evaluate = __call__

    def evaluate_grid(self, axes):
        """`self(points)` for points = meshgrid(*axes, indexing="ij") flattened: axes with uniform
        steps go through the lattice entry point (the steps of a regular binning's oversampled
        centres, the coszen reflection included, are uniform), anything else is written out"""
        axes = [np.asarray(a, dtype=np.float64) for a in axes]
        lat = _lattice_of(axes)
        if lat is not None:
            return self._est.evaluate_lattice(*lat)
        grid = np.meshgrid(*axes, indexing="ij")
        return self(np.array([g.ravel() for g in grid]))

    pairs = property(lambda self: (self._est.pairs_pilot, self._est.pairs_eval))


class bootstrap_kde:  # pylint: disable=invalid-name
    """Call contract of `kde.bootstrap_kde` as `get_hist(bootstrap=True)` uses it (kde_hist.py:108-109,
    155-158): `k = bootstrap_kde(x[D,N], niter=, weights=, bw_method=, adaptive=, alpha=)`;
    `mean, errors = k(points)`.  The package is not part of the reference tree (PARITY UNPINNED); this
    build's form: `niter` estimators, each built on the sample resampled with replacement -- expressed as
    multiplicities on the weights (`numpy.random.default_rng(seed).integers` / `bincount`, the draws of the
    KDE stage's own bootstrap, stages/utils/kde.py:197-238) -- evaluated at the points; mean and standard
    deviation (ddof = 0) over the iterations.  The reference's KDE stage does not call this path (its call
    is commented out, utils/kde.py:195-196); the stage-level bootstrap is the one pipelines use."""

    def __init__(self, dataset, niter=10, weights=None, bw_method="silverman", adaptive=True, alpha=0.3,
                 use_cuda=False, tol=None, seed=0):  # pylint: disable=unused-argument
        x = dataset if torch.is_tensor(dataset) else K.to_device(np.atleast_2d(dataset))
        n = x.shape[1]
        w = None
        if weights is not None and len(weights) != 0:
            w = weights if torch.is_tensor(weights) else K.to_device(np.asarray(weights))
        rng = np.random.default_rng(seed)
        self.kernels = []
        for _ in range(int(niter)):
            draws = K.to_device(np.bincount(rng.integers(n, size=n), minlength=n).astype(np.float64))
            self.kernels.append(gaussian_kde(x, weights=draws if w is None else w * draws, bw_method=bw_method,
                                             adaptive=adaptive, alpha=alpha, tol=tol))

    def _stats(self, values):
        stack = torch.stack(values)
        return stack.mean(dim=0), stack.std(dim=0, unbiased=False)

    def __call__(self, points):
        return self._stats([k(points) for k in self.kernels])

    def evaluate_grid(self, axes):
        return self._stats([k.evaluate_grid(axes) for k in self.kernels])


_GRIDS = {}


def _evaluation_grid(binning, oversample, coszen_name, coszen_reflection):
    """everything of `get_hist` that depends on the binning alone (kde_hist.py:82-102, 122-150):
    the oversampled binning with coszen first, the evaluation axes with the reflected coszen
    points, shapes and bin volumes.  Kept per binning: a fit evaluates the same maps every step."""
    key = (hash(binning), int(oversample), coszen_name, float(coszen_reflection))
    hit = _GRIDS.get(key)
    if hit is not None:
        return hit
    over = binning.oversample(oversample)
    cz_bin = over.index(coszen_name)
    if cz_bin != 0:
        over = MultiDimBinning([over[coszen_name]] + [b for b in over if b.name != coszen_name])
    edges = over[coszen_name].edge_magnitudes
    reflect_lower, reflect_upper = edges[0] == -1, edges[-1] == 1
    bin_points, l = [], 0
    for b in over:
        c = np.asarray(b.weighted_centers.magnitude)
        if b.name == coszen_name:
            l = int(len(c) * float(coszen_reflection))
            c0 = 2 * c[0] - c[1: l + 1][::-1] if reflect_lower else []
            c1 = 2 * c[-1] - c[-l - 1: -1][::-1] if reflect_upper else []
            c = np.concatenate([c0, c, c1])
        bin_points.append(c)
    hit = dict(binning=over, cz_bin=cz_bin, reflect_lower=reflect_lower, reflect_upper=reflect_upper, l=l,
               bin_points=bin_points,
               megashape=(over.shape[0] + (int(reflect_upper) + int(reflect_lower)) * l, over.shape[1]),
               minishape=(over.shape[0] - l, over.shape[1]),
               n_points=int(np.prod([len(c) for c in bin_points])),
               volumes=over.bin_volumes(attach_units=False),
               reduce_at=[np.arange(0, b.num_bins, oversample) for b in over])
    _GRIDS[key] = hit
    return hit


def get_hist(sample, binning, weights=None, bw_method="scott", adaptive=True, alpha=0.3,
             use_cuda=False, coszen_reflection=0.25, coszen_name="coszen", oversample=1,
             bootstrap=False, bootstrap_niter=10, tol=None, stats=None):
    """kde_hist.py:35-217.  `sample` [N, D] host array or device tensor; `weights` [N] likewise."""
    if bootstrap and oversample > 1:
        # errors within a bin are highly correlated (kde_hist.py:67-70)
        raise ValueError("Bootstrapping cannot be combined with oversampling.")
    on_dev = torch.is_tensor(sample)
    if weights is None or len(weights) == 0:
        weights_d, norm = None, sample.shape[0]
    else:
        weights_d = torch.nan_to_num(weights if torch.is_tensor(weights) else K.to_device(np.asarray(weights)))
        norm = float(weights_d.sum()) if bootstrap else None   # (plain: the estimator's own sum, below)
    g = _evaluation_grid(binning, oversample, coszen_name, coszen_reflection)
    x = (sample.T if on_dev else K.to_device(np.ascontiguousarray(np.asarray(sample).T))).clone()
    assert x.shape[0] == len(g["binning"])
    cz_bin, l = g["cz_bin"], g["l"]
    if cz_bin != 0:
        x[[0, cz_bin]] = x[[cz_bin, 0]]
    variances = None
    if bootstrap:
        kernel = bootstrap_kde(x.contiguous(), niter=bootstrap_niter, weights=weights_d, bw_method=bw_method,
                               adaptive=adaptive, alpha=alpha, tol=tol)
        mean_d, err_d = kernel.evaluate_grid(g["bin_points"])
        hist = mean_d.cpu().numpy().reshape(g["megashape"])
        variances = (err_d.cpu().numpy() ** 2).reshape(g["megashape"])   # variances add under the reflection
        kernel = kernel.kernels[0]
    else:
        kernel = gaussian_kde(x.contiguous(), weights=weights_d, bw_method=bw_method, adaptive=adaptive,
                              alpha=alpha, tol=tol)
        hist = kernel.evaluate_grid(g["bin_points"]).cpu().numpy().reshape(g["megashape"])
        norm = kernel.sum_w    # sum of the weights as the estimator formed it (what `kde_histogramdd_batch` uses)
    if stats is not None:
        stats["pairs_pilot"] = stats.get("pairs_pilot", 0) + kernel.pairs[0]
        stats["pairs_eval"] = stats.get("pairs_eval", 0) + kernel.pairs[1]
        stats["all_pairs"] = stats.get("all_pairs", 0) + kernel.n * (kernel.n * bool(adaptive) + g["n_points"])
    hist, errors = _finish_hist(g, hist, variances, oversample)
    if errors is not None:
        return hist * norm, errors * norm
    return hist * norm


def _finish_hist(g, hist, variances, oversample):
    """from the densities at the evaluation points to the map (kde_hist.py:168-217): reflection at
    coszen = -1 / +1, bin volumes, block sums of the oversampling, coszen back to its axis"""
    l, cz_bin = g["l"], g["cz_bin"]

    def reflect(h):   # kde_hist.py:168-190
        if g["reflect_lower"]:
            h0 = np.flipud(np.concatenate([np.zeros(g["minishape"]), h[0:l, :]]))
            h = h[l:, :]
        else:
            h0 = 0
        if g["reflect_upper"]:
            h1 = np.flipud(np.concatenate([h[-l:, :], np.zeros(g["minishape"])]))
            h = h[:-l, :]
        else:
            h1 = 0
        return h + h1 + h0

    hist = reflect(hist) * g["volumes"]
    errors = None
    if variances is not None:
        errors = np.sqrt(reflect(variances)) * g["volumes"]
    if oversample != 1:
        for i, at in enumerate(g["reduce_at"]):
            hist = np.add.reduceat(hist, at, axis=i)
    if cz_bin != 0:
        hist = np.swapaxes(hist, 0, cz_bin)
        if errors is not None:
            errors = np.swapaxes(errors, 0, cz_bin)
    return hist, errors


def pid_channels(sample, binning):
    """stack_pid: per pid bin the device indices of its events and their [2, n] sample in the
    other two dimensions (kde_hist.py:303-372).  Depends on the (static) sample only, so callers
    that re-weight the same events every evaluation keep the result."""
    s = sample if torch.is_tensor(sample) else K.to_device(np.asarray(sample))
    names = copy.copy(binning.names)
    pid_bin = names.index("pid")
    other = [0, 1, 2]
    other.pop(pid_bin)
    names.pop(pid_bin)
    assert len(names) == 2
    pid_edges = binning["pid"].edge_magnitudes
    d2d = MultiDimBinning([b for b in binning if b.name != "pid"])
    chans = []
    for pid in range(len(pid_edges) - 1):
        mask = (s[:, pid_bin] >= pid_edges[pid]) & (s[:, pid_bin] < pid_edges[pid + 1])
        idx = torch.nonzero(mask).reshape(-1)
        chans.append((idx, s[idx][:, other].contiguous()))
    return pid_bin, d2d, chans


def kde_histogramdd(sample, binning, weights=None, bw_method="scott", adaptive=True, alpha=0.3,
                    use_cuda=False, coszen_reflection=0.25, coszen_name="coszen", oversample=1,
                    stack_pid=True, bootstrap=False, bootstrap_niter=10, tol=None, stats=None,
                    channels=None):
    """kde_hist.py:220-387.  `sample` [N, D], `weights` [N]: host arrays or device tensors.
    `channels`: result of `pid_channels(sample, binning)` if the caller kept it."""
    if weights is not None and len(weights) != sample.shape[0]:
        raise ValueError("Length of sample (%s) and weights (%s) incompatible"
                         % (sample.shape[0], len(weights)))
    kw = dict(bw_method=bw_method, adaptive=adaptive, alpha=alpha, coszen_reflection=coszen_reflection,
              coszen_name=coszen_name, oversample=oversample, bootstrap=bootstrap,
              bootstrap_niter=bootstrap_niter, tol=tol, stats=stats)
    if not stack_pid:
        return get_hist(sample=sample, binning=binning, weights=weights, **kw)
    pid_bin, d2d, chans = channels if channels is not None else pid_channels(sample, binning)
    w_d = None
    if weights is not None:
        w_d = weights if torch.is_tensor(weights) else K.to_device(np.asarray(weights))
    stack, err_stack = [], []
    for idx, data in chans:
        w = None if w_d is None else w_d[idx]
        res = get_hist(sample=data, weights=w, binning=d2d, **kw)
        if bootstrap:
            stack.append(res[0])
            err_stack.append(res[1])
        else:
            stack.append(res)
    hist = np.dstack(stack)
    errors = np.dstack(err_stack) if bootstrap else None
    if pid_bin != 2:
        hist = np.swapaxes(hist, pid_bin, 2)
        if errors is not None:
            errors = np.swapaxes(errors, pid_bin, 2)
    return (hist, errors) if bootstrap else hist


def _finish_hist_many(g, dens, oversample):
    """`_finish_hist` for a stack of density arrays [k, megashape...]: the same element operations, once"""
    l, cz_bin = g["l"], g["cz_bin"]
    lo = l if g["reflect_lower"] else 0
    hi = dens.shape[1] - (l if g["reflect_upper"] else 0)
    hist = dens[:, lo:hi, :].copy()
    # h + h1 + h0 of `_finish_hist`: the upper mirror image first, then the lower one (adding its zeros changes nothing)
    if g["reflect_upper"]:
        hist[:, hist.shape[1] - l:, :] += dens[:, :hi - 1:-1, :]
    if g["reflect_lower"]:
        hist[:, :l, :] += dens[:, lo - 1::-1, :] if lo else 0
    hist *= g["volumes"]
    if oversample != 1:
        for i, at in enumerate(g["reduce_at"]):
            hist = np.add.reduceat(hist, at, axis=i + 1)
    if cz_bin != 0:
        hist = np.swapaxes(hist, 1, cz_bin + 1)
    return hist


def _lattice_of(axes):
    """(origin, step, count) if every axis has >= 2 ascending points at uniform steps, else None"""
    axes = [np.asarray(a, dtype=np.float64) for a in axes]
    if all(len(a) >= 2 and a[-1] > a[0] and np.allclose(np.diff(a), (a[-1] - a[0]) / (len(a) - 1), rtol=1e-9, atol=0.0)
           for a in axes):
        return [a[0] for a in axes], [(a[-1] - a[0]) / (len(a) - 1) for a in axes], [len(a) for a in axes]
    return None


def _job_sample(data, cz_bin):
    """[n, D] device sample -> [D, n] contiguous with coszen first; kept on the tensor (static per channel)"""
    cache = getattr(data, "_kde_x", None)
    if cache is None:
        cache = data._kde_x = {}
    x = cache.get(cz_bin)
    if x is None:
        x = data.T.clone()
        if cz_bin != 0:
            x[[0, cz_bin]] = x[[cz_bin, 0]]
        x = cache[cz_bin] = x.contiguous()
    return x


def kde_histogramdd_batch(samples, binning, bw_method="scott", adaptive=True, alpha=0.3, coszen_reflection=0.25,
                          coszen_name="coszen", oversample=1, stack_pid=True, tol=None, stats=None, n_threads=0):
    """`[kde_histogramdd(sample=s["sample"], weights=s["weights"], channels=s.get("channels"), ...) for s in
    samples]` with all the estimators (one per sample and pid channel) built and evaluated on the library's own
    threads and streams (`K.KdeLatticeBatch`) and one copy of the densities to the host.  `weights` may be a
    callable returning the tensor: it is called when the sample's turn comes, while the estimators of the earlier
    samples already run.  Samples: device tensors (host arrays are uploaded); without bootstrap.  The maps are those of the one-by-one path bit
    for bit."""
    samples = [dict(smp) for smp in samples]
    plans, g, n_jobs = [], None, 0
    for smp in samples:
        if not torch.is_tensor(smp["sample"]):      # a host array, as `kde_histogramdd` takes it too
            smp["sample"] = K.to_device(np.ascontiguousarray(np.asarray(smp["sample"], dtype=np.float64)))
        sample = smp["sample"]
        if stack_pid:
            pid_bin, d2d, chans = smp.get("channels") or pid_channels(sample, binning)
            g = _evaluation_grid(d2d, oversample, coszen_name, coszen_reflection)
            plans.append((pid_bin, [(_job_sample(data, g["cz_bin"]), idx) for idx, data in chans]))
        else:
            g = _evaluation_grid(binning, oversample, coszen_name, coszen_reflection)
            plans.append((None, [(_job_sample(sample, g["cz_bin"]), None)]))
        n_jobs += len(plans[-1][1])
    if not n_jobs:
        return []

    def weights_of(smp):
        w = smp.get("weights")
        w = w() if callable(w) else w
        if w is not None and not torch.is_tensor(w):
            w = K.to_device(np.asarray(w))
        if w is not None and len(w) != smp["sample"].shape[0]:
            raise ValueError("Length of sample (%s) and weights (%s) incompatible" % (smp["sample"].shape[0], len(w)))
        return w

    lat = _lattice_of(g["bin_points"])
    if lat is None:   # evaluation points not on a lattice (an irregular binning): one by one, points written out
        return [kde_histogramdd(sample=smp["sample"], binning=binning, weights=weights_of(smp), bw_method=bw_method,
                                adaptive=adaptive, alpha=alpha, coszen_reflection=coszen_reflection,
                                coszen_name=coszen_name, oversample=oversample, stack_pid=stack_pid, tol=tol,
                                stats=stats, channels=smp.get("channels")) for smp in samples]
    batch = K.KdeLatticeBatch(n_jobs, lat[0], lat[1], lat[2], samples[0]["sample"].device, bw_method=bw_method,
                              adaptive=adaptive, alpha=alpha, tol=K.KDE_DEFAULT_TOL if tol is None else tol, n_threads=n_threads)
    sizes, first_of = [], []
    try:
        for si, (smp, (pid_bin, chans)) in enumerate(zip(samples, plans)):
            w = weights_of(smp)
            first_of.append(batch.n)
            batch.submit([(x, w, idx if w is not None else None) for x, idx in chans])
            sizes += [int(x.shape[1]) for x, _ in chans]
        # sample by sample, as its estimators finish: densities to the host (one copy per sample), folded and summed into
        # bins while the later samples' estimators still run -- the element operations of `_finish_hist`, job by job
        per_sample = [[] for _ in samples]
        for si, (pid_bin, chans) in enumerate(plans):
            dens, sums = batch.wait_jobs(first_of[si], len(chans))
            dens = dens.cpu().numpy()
            hists = _finish_hist_many(g, dens.reshape((len(chans),) + tuple(g["megashape"])), oversample)
            hists = hists * np.asarray(sums, dtype=np.float64).reshape((-1,) + (1,) * (hists.ndim - 1))
            per_sample[si] = list(hists)
    finally:
        pairs = batch.wait()[2] if batch.n else (0, 0)
    if stats is not None:
        stats["pairs_pilot"] = stats.get("pairs_pilot", 0) + pairs[0]
        stats["pairs_eval"] = stats.get("pairs_eval", 0) + pairs[1]
        stats["all_pairs"] = stats.get("all_pairs", 0) + sum(n * (n * bool(adaptive) + g["n_points"]) for n in sizes)
    pid_of = [pid_bin for pid_bin, _ in plans]
    out = []
    for si, stack in enumerate(per_sample):
        if not stack_pid:
            out.append(stack[0])
            continue
        hist = np.dstack(stack)
        if pid_of[si] != 2:
            hist = np.swapaxes(hist, pid_of[si], 2)
        out.append(hist)
    return out
