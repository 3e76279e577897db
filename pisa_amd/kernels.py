"""Device-tensor front-end of the C ABI.

Arguments are ``torch`` CUDA(=HIP) tensors used purely as device buffers
(``data_ptr()`` + current stream are handed to ``libpisa_hip.so``); every
function launches hand-written gfx950 kernels and returns device tensors.
Function names follow the reference functions they replace.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

F8 = torch.float64


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """torch's current HIP stream of the current device as a raw handle.  The raw
    accessor costs ~0.3 us; `torch.cuda.current_stream()` builds a Python Stream
    object (~7 us), which matters in front of the first launch of an evaluation."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device tensors must be contiguous CUDA tensors"
    return C.c_void_p(t.data_ptr())


def device(index=None):
    if not torch.cuda.is_available():
        raise RuntimeError(
            "pisa_amd needs a HIP device (MI355X); none is visible and there is no CPU fallback"
        )
    return torch.device("cuda", torch.cuda.current_device() if index is None else index)


def to_device(a, dtype=np.float64):
    """Host numpy array -> contiguous device tensor.  Small arrays (per-evaluation scale factors, maps of a few
    hundred bins, pseudo-data) go through a page-locked block of torch's caching host allocator and an asynchronous
    copy on the current stream: a pageable source makes the runtime stage and synchronise (~20 us per call);
    the allocator keeps the block alive until the copy has run."""
    arr = np.ascontiguousarray(a, dtype=dtype)
    if 0 < arr.nbytes <= (1 << 18):
        host = torch.empty(arr.shape, dtype=torch.from_numpy(arr[:0].reshape(-1)).dtype, pin_memory=True)
        host.numpy()[...] = arr
        return host.to(device(), non_blocking=True)
    return torch.from_numpy(arr).to(device(), non_blocking=False)


def _status_buffer():
    return torch.zeros(1, dtype=torch.int32, device=device())


# ---------------------------------------------------------------- prob3
def propagate_array(params, nubar, energy, densities, distances, out=None):
    """`propagate_array` gufunc (numba_osc_hostfuncs.py:56-70)."""
    lib = _lib.lib()
    n = energy.numel()
    per_elem = 1 if densities.dim() == 2 else 0
    n_layers = densities.shape[-1]
    if out is None:
        out = torch.empty((n, 3, 3), dtype=F8, device=energy.device)
    _lib.check(lib.pisa_hip_propagate_array(
        C.byref(params), int(nubar), _ptr(energy), _ptr(densities), _ptr(distances), n,
        n_layers, per_elem, _ptr(out), _stream()))
    return out


def prob3_grid(params, energy, densities, distances, e_major=True, out_nu=None, out_nubar=None,
               out_pepmu=None, want_pepmu=False):
    """Both 'nu' and 'nubar' linked containers of a 2-D calc grid in one launch
    (prob3.py:452-459, 581-588).  Returns (P_nu, P_nubar[, pepmu])."""
    lib = _lib.lib()
    n_e, n_cz, n_layers = energy.numel(), densities.shape[0], densities.shape[1]
    if out_nu is None:
        out_nu = torch.empty((n_e * n_cz, 3, 3), dtype=F8, device=energy.device)
    if out_nubar is None:
        out_nubar = torch.empty((n_e * n_cz, 3, 3), dtype=F8, device=energy.device)
    if out_pepmu is None and want_pepmu:
        out_pepmu = torch.empty((2, 3, n_e * n_cz, 2), dtype=F8, device=energy.device)
    _lib.check(lib.pisa_hip_prob3_grid(
        C.byref(params), _ptr(energy), n_e, _ptr(densities), _ptr(distances), n_cz, n_layers,
        1 if e_major else 0, _ptr(out_nu), _ptr(out_nubar), _ptr(out_pepmu), _stream()))
    if out_pepmu is not None:
        return out_nu, out_nubar, out_pepmu
    return out_nu, out_nubar


class GridPlan:
    """Per-layer-table plan of the two-stage grid kernels (`pisa_hip_grid_plan`):
    cache matches resolved per coszen row + list of distinct shell densities."""

    def __init__(self, densities, distances):
        lib = _lib.lib()
        self.n_cz, self.n_layers = densities.shape
        h = C.c_void_p()
        torch.cuda.synchronize()
        _lib.check(lib.pisa_hip_grid_plan_create(_ptr(densities), _ptr(distances), self.n_cz,
                                                 self.n_layers, C.byref(h)))
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.lib().pisa_hip_grid_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def prob3_grid_planned(params, plan, energy, e_major=True, out_nu=None, out_nubar=None,
                       out_pepmu=None):
    """Planned grid evaluation (equal to `prob3_grid` to rounding, see csrc/prob3.hip)."""
    lib = _lib.lib()
    n_e = energy.numel()
    n = n_e * plan.n_cz
    if out_nu is None:
        out_nu = torch.empty((n, 3, 3), dtype=F8, device=energy.device)
    if out_nubar is None:
        out_nubar = torch.empty((n, 3, 3), dtype=F8, device=energy.device)
    if out_pepmu is None:
        out_pepmu = torch.empty((2, 3, n, 2), dtype=F8, device=energy.device)
    _lib.check(lib.pisa_hip_prob3_grid_planned(
        C.byref(params), plan.handle, _ptr(energy), n_e, 1 if e_major else 0, _ptr(out_nu),
        _ptr(out_nubar), _ptr(out_pepmu), _stream()))
    return out_nu, out_nubar, out_pepmu


def calc_layers(earth, coszen, max_layers):
    """`extCalcLayers` (layers.py:38-169) -> (n_layers, densities, distances)."""
    lib = _lib.lib()
    n = coszen.numel()
    dev = coszen.device
    n_layers = torch.empty(n, dtype=F8, device=dev)
    dens = torch.empty((n, max_layers), dtype=F8, device=dev)
    dist = torch.empty((n, max_layers), dtype=F8, device=dev)
    st = _status_buffer()
    _lib.check(lib.pisa_hip_calc_layers(
        C.byref(earth), _ptr(coszen), n, max_layers, _ptr(n_layers), _ptr(dens), _ptr(dist),
        _ptr(st), _stream()))
    if int(st.item()) != 0:
        # the reference raises a broadcast ValueError here (layers.py:158)
        raise ValueError("path geometry not representable (detector below the first Earth "
                         "shell boundary, or coszen exactly tangent to a shell)")
    return n_layers, dens, dist


def prob3_events(params, earth, nubar, energy, coszen, out=None):
    """Event-by-event prob3 with in-kernel layer reconstruction."""
    lib = _lib.lib()
    n = energy.numel()
    if out is None:
        out = torch.empty((n, 3, 3), dtype=F8, device=energy.device)
    st = _status_buffer()
    _lib.check(lib.pisa_hip_prob3_events(
        C.byref(params), C.byref(earth), int(nubar), _ptr(energy), _ptr(coszen), n, _ptr(out),
        _ptr(st), _stream()))
    if int(st.item()) != 0:
        raise ValueError("path geometry not representable (see calc_layers)")
    return out


def prob3_events_multi(params, earth, event_sets, status):
    """Event-by-event prob3 for several containers in one launch; each set
    receives its (prob_e, prob_mu) pairs and/or full matrices."""
    lib = _lib.lib()
    arr = event_sets if isinstance(event_sets, C.Array) else (_lib.EventSet * len(event_sets))(*event_sets)
    _lib.check(lib.pisa_hip_prob3_events_multi(C.byref(params), C.byref(earth), arr, len(arr),
                                               _ptr(status), _stream()))


def fill_probs(probability, init_flav, flav, out=None):
    """`fill_probs` (numba_osc_hostfuncs.py:206-221)."""
    lib = _lib.lib()
    n = probability.shape[0]
    if out is None:
        out = torch.empty(n, dtype=F8, device=probability.device)
    _lib.check(lib.pisa_hip_fill_probs(_ptr(probability), int(init_flav), int(flav), n, _ptr(out),
                                       _stream()))
    return out


# ---------------------------------------------------------- translation
def _sample_array(sample):
    arr = (C.c_void_p * len(sample))(*[s.data_ptr() for s in sample])
    for s in sample:
        assert s.is_cuda and s.is_contiguous()
    return arr


def lookup_regular(sample, flat_hist, binning):
    """`lookup` on regular binnings (translation.py:417-501)."""
    lib = _lib.lib()
    n = sample[0].numel()
    width = 1 if flat_hist.dim() == 1 else flat_hist.shape[1]
    out = torch.empty((n,) if flat_hist.dim() == 1 else (n, width), dtype=F8,
                      device=flat_hist.device)
    _lib.check(lib.pisa_hip_lookup_regular(
        C.byref(binning), _sample_array(sample), n, _ptr(flat_hist), width, _ptr(out), _stream()))
    return out


def histogram_regular(sample, weights, binning, averaged=False):
    """`histogram` on regular binnings (translation.py:90-129)."""
    lib = _lib.lib()
    n = sample[0].numel()
    n_bins = int(np.prod([binning.nbins[k] for k in range(binning.ndim)]))
    out = torch.empty(n_bins, dtype=F8, device=sample[0].device)
    _lib.check(lib.pisa_hip_histogram_regular(
        C.byref(binning), _sample_array(sample), n, _ptr(weights), 1 if averaged else 0,
        _ptr(out), _stream()))
    return out


def transform_apply(weights, unc_weights, ptr, col, val, n_out, errors=True):
    """(hist, sumw2, bin_unc2) = weights-in-calc-bins applied to a `hist_transform` given as CSR over
    the output bins (see pisa_hip_transform_apply)"""
    lib = _lib.lib()
    dev = weights.device
    hist = torch.empty(n_out, dtype=F8, device=dev)
    sumw2 = torch.empty(n_out, dtype=F8, device=dev) if errors else None
    unc2 = torch.empty(n_out, dtype=F8, device=dev) if errors else None
    _lib.check(lib.pisa_hip_transform_apply(_ptr(weights), _ptr(unc_weights), _ptr(ptr), _ptr(col), _ptr(val),
                                            int(n_out), _ptr(hist), _ptr(sumw2), _ptr(unc2), _stream()))
    return hist, sumw2, unc2


def event_indices(sample, binning):
    """flat bin index (int32, -1 outside) of every event in a regular binning"""
    lib = _lib.lib()
    n = sample[0].numel()
    out = torch.empty(n, dtype=torch.int32, device=sample[0].device)
    _lib.check(lib.pisa_hip_event_indices(C.byref(binning), _sample_array(sample), n, _ptr(out),
                                          _stream()))
    return out


class HistWorkspace:
    """Limb / map buffers for `reweight_hist` (allocated once, reused)."""

    def __init__(self, n_containers, n_bins, dev=None):
        dev = dev or device()
        self.n_containers, self.n_bins = n_containers, n_bins
        self.limbs = torch.zeros((n_containers, n_bins, 2, _lib.ACC_LIMBS), dtype=torch.int64,
                                 device=dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self.hist = torch.empty((n_containers, n_bins), dtype=F8, device=dev)
        self.sumw2 = torch.empty((n_containers, n_bins), dtype=F8, device=dev)


def reweight_hist(containers, calc_grid, prob_nu, prob_nubar, pepmu, out_binning, ws,
                  clear=True):
    """Fused prob3.apply + aeff.apply + hist.apply(sumw2); fills ws.limbs
    (`clear=False`: adds to them, e.g. after `finalize_metric(..., clear_limbs=True)`)."""
    lib = _lib.lib()
    arr = containers if isinstance(containers, C.Array) else (_lib.Container * len(containers))(*containers)
    fn = lib.pisa_hip_reweight_hist if clear else lib.pisa_hip_reweight_hist_acc
    _lib.check(fn(
        arr, len(arr), C.byref(calc_grid), _ptr(prob_nu), _ptr(prob_nubar), _ptr(pepmu),
        C.byref(out_binning), _ptr(ws.limbs), _ptr(ws.status), _stream()))
    return ws.limbs


def hist_finalize(ws):
    lib = _lib.lib()
    _lib.check(lib.pisa_hip_hist_finalize(_ptr(ws.limbs), ws.n_containers, ws.n_bins,
                                          _ptr(ws.hist), _ptr(ws.sumw2), _ptr(ws.status),
                                          _stream()))
    return ws.hist, ws.sumw2


FINALIZE_METRIC_MAX = 4096


def finalize_metric(ws, kind, actual, total_out, metric_status, clear_limbs=True):
    """`hist_finalize` + `metric` (maps summed over containers) in one launch;
    `total_out` may be a device tensor or a pinned host tensor.  Same bits as the
    two separate calls."""
    lib = _lib.lib()
    _lib.check(lib.pisa_hip_finalize_metric(
        _ptr(ws.limbs), ws.n_containers, ws.n_bins, _ptr(ws.hist), _ptr(ws.sumw2),
        METRIC_KIND[kind], _ptr(actual), total_out.data_ptr(), _ptr(ws.status),
        _ptr(metric_status), 1 if clear_limbs else 0, _stream()))
    return total_out


def apply_osc_weights(nu_flux, prob_e, prob_mu, weights):
    """prob3.apply_function (prob3.py:621-622), in place on `weights`.  `prob_e` / `prob_mu` may be 1-D views with
    the same element stride (columns of one table): read in place."""
    lib = _lib.lib()
    if (prob_e.dim() == 1 and prob_mu.dim() == 1 and prob_e.stride(0) == prob_mu.stride(0) > 1
            and prob_e.is_cuda and prob_mu.is_cuda and prob_e.dtype == prob_mu.dtype == F8
            and prob_e.numel() == prob_mu.numel() == weights.numel()):
        _lib.check(lib.pisa_hip_apply_osc_weights_strided(_ptr(nu_flux), prob_e.data_ptr(), prob_mu.data_ptr(),
                                                          prob_e.stride(0), weights.numel(), _ptr(weights), _stream()))
        return weights
    # (any other view -- different strides, more dimensions -- is compacted first)
    prob_e = prob_e if prob_e.is_contiguous() else prob_e.contiguous()
    prob_mu = prob_mu if prob_mu.is_contiguous() else prob_mu.contiguous()
    _lib.check(lib.pisa_hip_apply_osc_weights(_ptr(nu_flux), _ptr(prob_e), _ptr(prob_mu),
                                              weights.numel(), _ptr(weights), _stream()))
    return weights


def weight_chain_multi(items):
    """The reset -> [osc] -> [aeff] chain of several containers in one launch (`pisa_hip_weight_chain_multi`).
    items: (initial_weights, nu_flux or None, prob_e, prob_mu, weighted_aeff or None, scale) per container, device
    tensors; prob_e / prob_mu may be equal-stride 1-D views.  Returns (flat block of all weights, list of its
    per-container views)."""
    lib = _lib.lib()
    sizes = [int(it[0].numel()) for it in items]
    block = torch.empty(sum(sizes), dtype=F8, device=items[0][0].device)
    sets = (_lib.ChainSet * len(items))()
    keep, views, off = [], [], 0
    for s, (w0, flux, pe, pmu, aeff, scale), n in zip(sets, items, sizes):
        out = block[off:off + n]
        off += n
        views.append(out)
        s.n, s.d_initial_weights, s.d_weights = n, _ptr(w0), out.data_ptr()
        if flux is not None:
            if not (pe.dim() == 1 and pmu.dim() == 1 and pe.stride(0) == pmu.stride(0) >= 1 and pe.numel() == pmu.numel() == n):
                pe, pmu = pe.contiguous().reshape(-1), pmu.contiguous().reshape(-1)
                keep += [pe, pmu]
            s.d_nu_flux, s.d_prob_e, s.d_prob_mu, s.prob_stride = _ptr(flux), pe.data_ptr(), pmu.data_ptr(), max(1, pe.stride(0))
        if aeff is not None:
            s.d_weighted_aeff, s.aeff_scale = _ptr(aeff), float(scale)
    _lib.check(lib.pisa_hip_weight_chain_multi(sets, len(items), _stream()))
    return block, views


def apply_aeff(weighted_aeff, scale, weights):
    """aeff.apply_function (aeff.py:87), in place on `weights`."""
    lib = _lib.lib()
    _lib.check(lib.pisa_hip_apply_aeff(_ptr(weighted_aeff), float(scale), weights.numel(),
                                       _ptr(weights), _stream()))
    return weights


# ------------------------------------------------------------------ KDE
def kde_eval(src, coef, s2, qry, inv_cov):
    """Gaussian kernel sums (see pisa_hip_kde_eval); src [dim, n], qry [dim, m]."""
    lib = _lib.lib()
    dim, n = src.shape
    m = qry.shape[1]
    out = torch.empty(m, dtype=F8, device=qry.device)
    ic = np.ascontiguousarray(inv_cov, dtype=np.float64)
    _lib.check(lib.pisa_hip_kde_eval(dim, _ptr(src), _ptr(coef), _ptr(s2), n, _ptr(qry), m,
                                     ic.ctypes.data, _ptr(out), _stream()))
    return out


KDE_BW = {"silverman": 0, "scott": 1}
KDE_DEFAULT_TOL = 1e-14


class KdeEstimator:
    """`gaussian_kde(x, weights, bw_method, adaptive, alpha)` on the device
    (`pisa_hip_kde_create` / `_evaluate`; cell-list cut-off at kernel value `tol`, 0 = all pairs).
    x [dim, n] device tensor (dimension-major), weights [n] or None."""

    def __init__(self, x, weights=None, bw_method="silverman", adaptive=True, alpha=0.3,
                 tol=KDE_DEFAULT_TOL):
        import ctypes as C

        lib = _lib.lib()
        if bw_method not in KDE_BW:
            raise ValueError("`bw_method` should be 'scott' or 'silverman'")
        x = x.contiguous()
        self.dim, self.n = int(x.shape[0]), int(x.shape[1])
        need = int(lib.pisa_hip_kde_workspace_bytes(self.dim, self.n))
        if need < 0:
            raise ValueError("KDE of %d points in %d dimensions not supported" % (self.n, self.dim))
        work = torch.empty(need, dtype=torch.uint8, device=x.device)
        w = None if weights is None else weights.contiguous()
        h = C.c_void_p()
        _lib.check(lib.pisa_hip_kde_create(self.dim, _ptr(x), None if w is None else _ptr(w), self.n,
                                           KDE_BW[bw_method], 1 if adaptive else 0, float(alpha),
                                           float(tol), _ptr(work), need, C.byref(h), _stream()))
        self._h, self._lib = h, lib
        self._work = work   # holds the estimator's device arrays (torch's caching allocator reuses it)
        info = _lib.KdeInfo()
        _lib.check(lib.pisa_hip_kde_info(h, C.byref(info)))
        d = self.dim
        self.factor, self.norm, self.sum_w = info.factor, info.norm, info.sum_w
        self.covariance = np.array(info.covariance).reshape(3, 3)[:d, :d].copy()
        self.inv_cov = np.array(info.inv_cov).reshape(3, 3)[:d, :d].copy()
        self.mean = np.array(info.mean)[:d].copy()
        self.r_cut, self.cell, self.n_cells = info.r_cut, info.cell, info.n_cells
        self.pairs_pilot, self.n_dense = info.pairs_pilot, info.n_dense
        self.pairs_eval = 0

    def __call__(self, points):
        import ctypes as C

        q = points.contiguous()
        m = int(q.shape[1])
        out = torch.empty(m, dtype=F8, device=q.device)
        if m == 0:
            return out
        need = int(self._lib.pisa_hip_kde_eval_workspace_bytes(self._h, m))
        work = torch.empty(need, dtype=torch.uint8, device=q.device)
        _lib.check(self._lib.pisa_hip_kde_evaluate(self._h, _ptr(q), m, _ptr(work), need, _ptr(out),
                                                   _stream()))
        info = _lib.KdeInfo()
        _lib.check(self._lib.pisa_hip_kde_info(self._h, C.byref(info)))
        self.pairs_eval = info.pairs_eval
        return out

    def evaluate_lattice(self, origin, step, count):
        """densities at the points origin[d] + i_d step[d], 0 <= i_d < count[d], as a flat device
        tensor in numpy.meshgrid(indexing="ij") order (`pisa_hip_kde_evaluate_lattice`)"""
        import ctypes as C

        d = self.dim
        assert len(origin) == len(step) == len(count) == d
        o = (C.c_double * d)(*[float(v) for v in origin])
        st = (C.c_double * d)(*[float(v) for v in step])
        cnt = (C.c_int64 * d)(*[int(v) for v in count])
        if min(int(v) for v in count) < 1:
            raise ValueError("empty lattice %r" % (list(count),))
        m = int(np.prod([int(v) for v in count]))
        out = torch.empty(m, dtype=F8, device=self._work.device)
        need = int(self._lib.pisa_hip_kde_lattice_workspace_bytes(self._h, st, cnt))
        if need < 0:
            raise ValueError("invalid lattice %r x %r" % (list(step), list(count)))
        work = torch.empty(need, dtype=torch.uint8, device=out.device)
        _lib.check(self._lib.pisa_hip_kde_evaluate_lattice(self._h, o, st, cnt, _ptr(work), need, _ptr(out),
                                                           _stream()))
        info = _lib.KdeInfo()
        _lib.check(self._lib.pisa_hip_kde_info(self._h, C.byref(info)))
        self.pairs_eval = info.pairs_eval
        return out

    def arrays(self):
        """(ys [dim, n], coef [n], s2 [n]) in the estimator's cell-sorted source order (copies)"""
        import ctypes as C

        p = [C.c_void_p() for _ in range(3)]
        _lib.check(self._lib.pisa_hip_kde_arrays(self._h, *[C.byref(v) for v in p]))
        base = self._work.data_ptr()
        out = []
        for v, cnt in zip(p, (self.dim * self.n, self.n, self.n)):
            off = v.value - base
            out.append(self._work[off:off + 8 * cnt].view(F8).clone())
        return out[0].view(self.dim, self.n), out[1], out[2]

    def close(self):
        if self._h is not None:
            self._lib.pisa_hip_kde_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# --------------------------------------------------------------- metric
METRIC_KIND = {"llh": 0, "poisson_llh": 1, "chi2": 2, "mod_chi2": 3}          # the fused tails of an evaluation
# `metric()` on maps takes these too (PISA_HIP_METRIC_* of include/pisa_hip.h)
MAP_METRIC_KIND = dict(METRIC_KIND, correct_chi2=4, signed_sqrt_mod_chi2=5, mcllh_mean=6, mcllh_eff=7, conv_llh=8)
VARIANCE_METRICS = ("mod_chi2", "correct_chi2", "signed_sqrt_mod_chi2", "mcllh_mean", "mcllh_eff", "conv_llh")


class KdeLatticeBatch:
    """The estimators of one KDE-stage evaluation on the library's own threads and streams
    (`pisa_hip_kde_lattice_submit` / `_wait`): `submit` queues jobs and returns -- the caller may go on
    producing the inputs of the next ones --, `wait` returns (densities [n_jobs, m] device tensor, [sum of the
    weights used per job], (pairs_pilot, pairs_eval) summed over the jobs).
    A job: (x [dim, n] device tensor, weights device tensor or None, index int64 device tensor or None);
    weights are those of the parent sample when an index is given."""

    def __init__(self, n_jobs, origin, step, count, dev, bw_method="silverman", adaptive=True, alpha=0.3,
                 tol=KDE_DEFAULT_TOL, n_threads=0):
        import ctypes as C

        if bw_method not in KDE_BW:
            raise ValueError("`bw_method` should be 'scott' or 'silverman'")
        d = len(origin)
        assert len(step) == len(count) == d
        if min(int(v) for v in count) < 1:
            raise ValueError("empty lattice %r" % (list(count),))
        self._lib = _lib.lib()
        self.dim, self.capacity, self.n = d, int(n_jobs), 0
        self.out = torch.empty((self.capacity, int(np.prod([int(v) for v in count]))), dtype=F8, device=dev)
        self._arr = (_lib.KdeJob * max(self.capacity, 1))()
        self._keep = []
        self._o = (C.c_double * d)(*[float(v) for v in origin])
        self._st = (C.c_double * d)(*[float(v) for v in step])
        self._cnt = (C.c_int64 * d)(*[int(v) for v in count])
        self._args = (KDE_BW[bw_method], 1 if adaptive else 0, float(alpha), float(tol))
        self._threads = int(n_threads)
        self._waited = False

    def submit(self, jobs):
        import ctypes as C

        first = self.n
        assert first + len(jobs) <= self.capacity and not self._waited
        for x, w, idx in jobs:
            x = x.contiguous()
            assert int(x.shape[0]) == self.dim
            self._keep.append(x)
            j = self._arr[self.n]
            j.d_x, j.n = _ptr(x), int(x.shape[1])
            if w is not None:
                w = w.contiguous()
                self._keep.append(w)
                j.d_weights = _ptr(w)
                if idx is not None:
                    idx = idx.contiguous()
                    assert idx.dtype == torch.int64 and int(idx.numel()) == j.n
                    self._keep.append(idx)
                    j.d_index = _ptr(idx)
                else:
                    assert int(w.numel()) == j.n
            j.d_out = _ptr(self.out[self.n])
            self.n += 1
        if self.n > first:
            sub = C.cast(C.addressof(self._arr) + first * C.sizeof(_lib.KdeJob), C.POINTER(_lib.KdeJob))
            _lib.check(self._lib.pisa_hip_kde_lattice_submit(sub, self.n - first, self.dim, self._args[0], self._args[1],
                                                             self._args[2], self._args[3], self._o, self._st, self._cnt,
                                                             self._threads, _stream()))

    def wait_jobs(self, first, count):
        """returns when jobs first .. first + count - 1 are done (`pisa_hip_kde_lattice_wait_jobs`), whatever else is
        still running: (their densities [count, m] -- a view of the batch's output --, [sum of the weights used per job])"""
        import ctypes as C

        assert 0 <= first and first + count <= self.n
        _lib.check(self._lib.pisa_hip_kde_lattice_wait_jobs(
            C.c_void_p(C.addressof(self._arr) + first * C.sizeof(_lib.KdeJob)), count))
        a = self._arr
        for i in range(first, first + count):
            _lib.check(a[i].status)
        return self.out[first:first + count], [a[i].sum_w for i in range(first, first + count)]

    def wait(self):
        _lib.check(self._lib.pisa_hip_kde_lattice_wait())
        self._waited = True
        for i in range(self.n):
            _lib.check(self._arr[i].status)
        a = self._arr
        return (self.out[:self.n], [a[i].sum_w for i in range(self.n)],
                (sum(a[i].pairs_pilot for i in range(self.n)), sum(a[i].pairs_eval for i in range(self.n))))

    def __del__(self):   # the library holds pointers into this object's job array until the jobs are done
        if self.n and not self._waited:
            try:
                self._lib.pisa_hip_kde_lattice_wait()
            except Exception:   # pylint: disable=broad-except
                pass


def kde_lattice_batch(jobs, origin, step, count, bw_method="silverman", adaptive=True, alpha=0.3,
                      tol=KDE_DEFAULT_TOL, n_threads=0):
    """all jobs of a `KdeLatticeBatch` submitted at once and waited for (`pisa_hip_kde_lattice_batch`)"""
    b = KdeLatticeBatch(len(jobs), origin, step, count, jobs[0][0].device if jobs else device(), bw_method=bw_method,
                        adaptive=adaptive, alpha=alpha, tol=tol, n_threads=n_threads)
    b.submit(jobs)
    return b.wait()


def metric(kind, actual, expected, sigma2=None, per_bin=False, total_out=None, status=None):
    """Map.metric + nansum (map.py:1572-1604) on device.  `expected` (and
    `sigma2`) may be [n_maps, n_bins]; maps are summed in index order first.
    Returns a 1-element device tensor (and per-bin values if requested);
    `total_out` may also be a pinned (device-mapped) host tensor."""
    lib = _lib.lib()
    n_bins = actual.numel()
    n_maps = 1 if expected.dim() == 1 else expected.shape[0]
    dev = actual.device
    if total_out is None:
        total_out = torch.empty(1, dtype=F8, device=dev)
    pb = torch.empty(n_bins, dtype=F8, device=dev) if per_bin else None
    own_status = status is None
    if own_status:
        status = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(lib.pisa_hip_metric(
        MAP_METRIC_KIND[kind], _ptr(actual), _ptr(expected), _ptr(sigma2), n_maps, n_bins, _ptr(pb),
        total_out.data_ptr() if not total_out.is_cuda and total_out.is_pinned() else _ptr(total_out),
        _ptr(status), _stream()))
    if own_status:
        st = int(status.item())
        if st != 0:
            _lib.check(st)
    return (total_out, pb) if per_bin else total_out


def bin_scale(x, scale=None, scalar=1.0, floor=None, out=None):
    """out = x * scale * scalar [floored]; see `pisa_hip_bin_scale`."""
    lib = _lib.lib()
    if out is None:
        out = torch.empty_like(x)
    _lib.check(lib.pisa_hip_bin_scale(_ptr(x), _ptr(scale), float(scalar), 0 if floor is None else 1,
                                      0.0 if floor is None else float(floor), x.numel(), _ptr(out),
                                      _stream()))
    return out


def lookup_indices(sample, edges):
    """flat bin number (int64) of every event by the bin edges: -1 below, n_bins above (`pisa_hip_lookup_indices`)"""
    lib = _lib.lib()
    n = sample[0].numel()
    ndim = len(sample)
    edges = [e.contiguous() for e in edges]
    n_edges = (C.c_int32 * ndim)(*[e.numel() for e in edges])
    out = torch.empty(n, dtype=torch.int64, device=sample[0].device)
    _lib.check(lib.pisa_hip_lookup_indices(_sample_array(sample), _sample_array(edges), n_edges, ndim, n, _ptr(out),
                                           _stream()))
    return out


def two_nu_osc(nu_flux, theta, deltam31, energy, coszen, flav, weights):
    """weights *= flux x two-flavour probability, in place (`pisa_hip_two_nu_osc`)"""
    lib = _lib.lib()
    assert nu_flux.is_contiguous() and nu_flux.shape == (energy.numel(), 2) and weights.is_contiguous()
    _lib.check(lib.pisa_hip_two_nu_osc(_ptr(nu_flux), float(theta), float(deltam31), _ptr(energy), _ptr(coszen), int(flav),
                                       energy.numel(), _ptr(weights), _stream()))
    return weights


def power_law(energy, pivot, index, norm=1.0, nominal=None, out=None):
    """norm * nominal * (E / pivot)^index (`pisa_hip_power_law`)"""
    lib = _lib.lib()
    if out is None:
        out = torch.empty_like(energy)
    _lib.check(lib.pisa_hip_power_law(_ptr(energy), float(pivot), float(index), float(norm), _ptr(nominal), energy.numel(),
                                      _ptr(out), _stream()))
    return out


def shift_toward(x, target, fraction, clip=None, out=None):
    """x + (target - x) * fraction [clipped]; `target` a device column or a number (`pisa_hip_shift_toward`)"""
    lib = _lib.lib()
    if out is None:
        out = torch.empty_like(x)
    col = target if isinstance(target, torch.Tensor) else None
    _lib.check(lib.pisa_hip_shift_toward(_ptr(x), _ptr(col), 0.0 if col is not None else float(target), float(fraction),
                                         0 if clip is None else 1, 0.0 if clip is None else float(clip[0]),
                                         0.0 if clip is None else float(clip[1]), x.numel(), _ptr(out), _stream()))
    return out


def poly_scale(linear, quad, params, weights, scale=1.0):
    """weights *= max(0, scale * prod_k (1 + (lin_k + quad_k p_k) p_k)), in place (`pisa_hip_poly_scale`)"""
    lib = _lib.lib()
    k = len(linear)
    assert k == len(params) and (quad is None or len(quad) == k) and weights.is_contiguous()
    lin = (C.c_void_p * max(k, 1))(*[t.data_ptr() for t in linear])
    qd = None if quad is None else (C.c_void_p * max(k, 1))(*[None if t is None else t.data_ptr() for t in quad])
    p = (C.c_double * max(k, 1))(*[float(v) for v in params])
    for t in list(linear) + [t for t in (quad or []) if t is not None]:
        assert t.is_cuda and t.is_contiguous() and t.numel() == weights.numel()
    _lib.check(lib.pisa_hip_poly_scale(lin, qd, p, k, float(scale), weights.numel(), _ptr(weights), _stream()))
    return weights


def column_combination(columns, coef, n, mode="exp", out=None, dev=None):
    """f(sum_g coef_g * column_g) per event, f = exp / 1 + / identity (`pisa_hip_column_combination`)"""
    lib = _lib.lib()
    k = len(columns)
    assert k == len(coef)
    for t in columns:
        assert t.is_cuda and t.is_contiguous() and t.numel() == n and t.dtype == F8
    if out is None:
        out = torch.empty(n, dtype=F8, device=columns[0].device if k else (dev or device()))
    cols = (C.c_void_p * max(k, 1))(*[t.data_ptr() for t in columns])
    cf = (C.c_double * max(k, 1))(*[float(v) for v in coef])
    _lib.check(lib.pisa_hip_column_combination(cols, cf, k, {"exp": 0, "one_plus": 1, "sum": 2}[mode], n, _ptr(out), _stream()))
    return out


VECTOR_OPS = {"scale": 0, "mul": 1, "imul": 2, "imul_and_scale": 3, "itruediv": 4, "assign": 5, "pow": 6, "sqrt": 7,
              "replace_where_counts_gt": 8}


def vector_op(op, a, out, b=None, scalar=0.0):
    """the element-wise helpers of pisa/utils/vectorizer.py on device tensors, `out` written in place
    (`pisa_hip_vector_op`)"""
    lib = _lib.lib()
    assert out.is_contiguous() and a.numel() == out.numel() and (b is None or b.numel() == out.numel())
    _lib.check(lib.pisa_hip_vector_op(VECTOR_OPS[op], _ptr(a), _ptr(b), float(scalar), out.numel(), _ptr(out), _stream()))
    return out


def interp_linear(x_knots, y_knots, x, out=None):
    """numpy.interp(x, x_knots, y_knots) on the device; a value outside the knots raises, as scipy's interp1d does
    (`pisa_hip_interp_linear`)"""
    lib = _lib.lib()
    if out is None:
        out = torch.empty_like(x)
    status = torch.zeros(1, dtype=torch.int32, device=x.device)
    _lib.check(lib.pisa_hip_interp_linear(_ptr(x_knots), _ptr(y_knots), x_knots.numel(), _ptr(x), x.numel(), _ptr(out),
                                          _ptr(status), _stream()))
    if int(status.item()):
        raise ValueError("A value in x_new is outside the interpolation range.")
    return out


def decoherence_probs(coef, gamma, delta, two_flavor, energy, baseline, out=None):
    """P[n, 3, 3] of the vacuum decoherence model (`pisa_hip_decoherence_probs`)"""
    lib = _lib.lib()
    n = energy.numel()
    assert baseline.numel() == n
    if out is None:
        out = torch.empty((n, 3, 3), dtype=F8, device=energy.device)
    arr = [(C.c_double * 3)(*[float(v) for v in a]) for a in (coef, gamma, delta)]
    _lib.check(lib.pisa_hip_decoherence_probs(arr[0], arr[1], arr[2], 1 if two_flavor else 0, _ptr(energy), _ptr(baseline), n,
                                              _ptr(out), _stream()))
    return out


def bin_sqrt(x, out=None):
    lib = _lib.lib()
    if out is None:
        out = torch.empty_like(x)
    _lib.check(lib.pisa_hip_bin_sqrt(_ptr(x), x.numel(), _ptr(out), _stream()))
    return out


def flux_2d(table, true_energy, true_coszen, out_nu=None, out_nubar=None):
    """`calculate_2d_flux_weights` (flux_weights.py:267-349) for (nue, numu) and
    (nuebar, numubar) at once; `table` is a `pisa_amd.utils.flux_weights.FluxTable2D`."""
    lib = _lib.lib()
    n = true_energy.numel()
    dev = true_energy.device
    if out_nu is None:
        out_nu = torch.empty((n, 2), dtype=F8, device=dev)
    if out_nubar is None:
        out_nubar = torch.empty((n, 2), dtype=F8, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(lib.pisa_hip_flux_2d(C.byref(table.struct), _ptr(true_energy), _ptr(true_coszen), n,
                                    _ptr(out_nu), _ptr(out_nubar), _ptr(status), _stream()))
    if int(status.item()) != 0:
        raise ValueError("Not all coszens found between -1 and 1")  # flux_weights.py:318-319
    return out_nu, out_nubar


def barr_sets(columns):
    """argument block of `barr_simple_multi`: one (true_energy, true_coszen, nu_flux_nominal,
    nubar_flux_nominal, nubar, out) tuple of device tensors per container.  The block holds raw
    pointers: the caller keeps the tensors alive."""
    arr = (_lib.BarrSet * len(columns))()
    for d, (e, cz, nu, nub, nubar, out) in zip(arr, columns):
        d.n = e.numel()
        assert cz.numel() == d.n and nu.numel() == 2 * d.n and nub.numel() == 2 * d.n and out.numel() == 2 * d.n
        d.d_true_energy, d.d_true_coszen = _ptr(e), _ptr(cz)
        d.d_nu_flux_nominal, d.d_nubar_flux_nominal, d.d_out = _ptr(nu), _ptr(nub), _ptr(out)
        d.nubar = int(nubar)
    return arr


def barr_simple_multi(sets, nue_numu_ratio, nu_nubar_ratio, delta_index, Barr_uphor_ratio,
                      Barr_nu_nubar_ratio):
    """`apply_sys_vectorized` for every container of a pipeline in one launch (flux/barr_simple.py:83-104)."""
    _lib.check(_lib.lib().pisa_hip_barr_simple_multi(
        sets, len(sets), float(nue_numu_ratio), float(nu_nubar_ratio), float(delta_index),
        float(Barr_uphor_ratio), float(Barr_nu_nubar_ratio), _stream()))


def barr_simple(true_energy, true_coszen, nu_flux_nominal, nubar_flux_nominal, nubar,
                nue_numu_ratio, nu_nubar_ratio, delta_index, Barr_uphor_ratio,
                Barr_nu_nubar_ratio, out=None):
    """`apply_sys_vectorized` (flux/barr_simple.py:207-233)."""
    lib = _lib.lib()
    n = true_energy.numel()
    if out is None:
        out = torch.empty((n, 2), dtype=F8, device=true_energy.device)
    _lib.check(lib.pisa_hip_barr_simple(
        _ptr(true_energy), _ptr(true_coszen), _ptr(nu_flux_nominal), _ptr(nubar_flux_nominal),
        int(nubar), float(nue_numu_ratio), float(nu_nubar_ratio), float(delta_index),
        float(Barr_uphor_ratio), float(Barr_nu_nubar_ratio), n, _ptr(out), _stream()))
    return out
