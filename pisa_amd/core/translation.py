"""Translations between event arrays and binned data as free functions: the interface of
pisa/core/translation.py (`histogram` :90-129, `lookup` :228-414, `resample` :49-85, `find_index` :504-553)
for stages that call it directly instead of going through a `Container`.

The arithmetic runs on the GPU (`pisa_hip_histogram_regular`, `pisa_hip_lookup_regular`); the reference's two
regimes are kept:
  * every dimension linear and regular: fast_histogram's rule -- bin = int((x - min) * n / (max - min)), the range
    half open, an event ON the last edge is outside (translation.py:171-205);
  * anything else (a logarithmic or an irregular dimension): numpy's rule for ALL dimensions -- bins found by
    comparison with the edges, the last edge included (`np.histogramdd`, translation.py:207-225; `find_index`).
    The events are digitised against the edges first and the kernels then run over the bin numbers.
Samples and weights may be host arrays (the result is a host array, like the reference's) or device tensors (the
result stays on the device).  There is no host implementation: without the HIP library these functions raise.
"""
import numpy as np

from pisa_amd import FTYPE
from pisa_amd.core.binning import MultiDimBinning

__all__ = ["histogram", "lookup", "resample", "find_index"]


def find_index(val, bin_edges):
    """bin number of `val` (scalar or array) among `bin_edges`: [ bin 0 ) [ bin 1 ) ... [ last bin ]; -1 below
    the first edge or for NaN, `num_bins` above the last (translation.py:504-553)"""
    edges = np.asarray(bin_edges, dtype=FTYPE)
    assert edges.ndim == 1 and len(edges) >= 2, "bin_edges must define at least one bin"
    n = len(edges) - 1
    v = np.asarray(val, dtype=FTYPE)
    with np.errstate(invalid="ignore"):
        idx = np.searchsorted(edges, v, side="right") - 1
        idx = np.where(v == edges[-1], n - 1, idx)          # the last edge belongs to the last bin
        idx = np.clip(idx, 0, n - 1)
        idx = np.where(v > edges[-1], n, idx)
        idx = np.where(v >= edges[0], idx, -1)               # below the range, or NaN
    return int(idx) if np.ndim(val) == 0 else idx.astype(np.int64)


def _columns(sample):
    if isinstance(sample, np.ndarray):
        return [sample] if sample.ndim == 1 else [np.ascontiguousarray(c) for c in sample.T]
    cols = list(sample)
    if not cols:
        raise ValueError("Sample should be either an (N, D) array, or an (N,) array, or a (D, N) array-like.")
    return cols


def _on_device(x):
    import torch

    from pisa_amd import kernels as K

    if isinstance(x, torch.Tensor):
        return x.to(dtype=torch.float64).contiguous()
    return K.to_device(np.ascontiguousarray(x, dtype=FTYPE))


def _is_tensor(x):
    return type(x).__module__.startswith("torch")


def _kernel_form(sample, binning, by_edges=False):
    """(argument block of the binning the kernels see, device columns): the coordinates themselves for a
    linear regular binning, else bin numbers in [0, n) with -1 outside"""
    from pisa_amd import _lib

    cols = _columns(sample)
    if len(cols) != binning.num_dims:
        raise ValueError("%d sample column(s) for a %d-dimensional binning" % (len(cols), binning.num_dims))
    if binning.is_lin and not binning.is_irregular and not by_edges:
        doms = [d.domain.magnitude for d in binning]
        b = _lib.make_binning([float(d[0]) for d in doms], [float(d[1]) for d in doms], [d.num_bins for d in binning])
        return b, [_on_device(c) for c in cols]
    import torch

    out = []
    for c, d in zip(cols, binning):
        edges = d.edge_magnitudes
        if _is_tensor(c):
            e = torch.tensor(np.array(edges), device=c.device)
            c = c.to(torch.float64)
            idx = torch.bucketize(c, e, right=True) - 1
            idx = torch.where(c == e[-1], torch.full_like(idx, d.num_bins - 1), idx)
            idx = torch.where((c >= e[0]) & (c <= e[-1]), idx, torch.full_like(idx, -1))
            out.append(idx.to(torch.float64).contiguous())
        else:
            idx = find_index(np.asarray(c, dtype=FTYPE), edges).astype(FTYPE)
            idx[idx == d.num_bins] = -1.0
            out.append(_on_device(idx))
    b = _lib.make_binning([0.0] * binning.num_dims, [float(d.num_bins) for d in binning], [d.num_bins for d in binning])
    return b, out


def _histogram(sample, weights, binning, averaged, count, by_edges=False):
    import torch

    from pisa_amd import kernels as K

    b, cols = _kernel_form(sample, binning, by_edges)
    to_host = not (_is_tensor(weights) or (weights is None and _is_tensor(_columns(sample)[0])))
    width = None if weights is None or np.ndim(weights) == 1 else int(np.shape(weights)[1])
    if weights is None or count:
        flat = K.histogram_regular(cols, None, b, averaged=False)
        if averaged:                         # count / count: 1 where there are events, 0 elsewhere
            flat = (flat > 0).to(torch.float64)
        if width is not None:
            flat = flat[:, None].expand(-1, width).contiguous()
    else:
        w = _on_device(weights)
        if width is not None:
            flat = torch.stack([K.histogram_regular(cols, w[:, i].contiguous(), b, averaged=averaged)
                                for i in range(width)], dim=1)
        else:
            flat = K.histogram_regular(cols, w, b, averaged=averaged)
    return flat.cpu().numpy().astype(FTYPE, copy=False) if to_host else flat


def histogram(sample, weights, binning, averaged, apply_weights=True):
    """`weights` of the events at `sample` summed (or, `averaged`, averaged: empty bins 0) per bin of `binning`;
    flat, C order; a [N, d] weight array gives [n_bins, d]; `weights=None` counts the events.  `apply_weights` is
    accepted and, as in the reference -- whose `histogram` passes `apply_weights=True` to its helpers whatever it
    was given (translation.py:110-113) --, does not change the result."""
    if not isinstance(binning, MultiDimBinning):
        raise ValueError("Binning should be a PISA MultiDimBinning")
    return _histogram(sample, weights, binning, averaged, count=False)


def lookup(sample, flat_hist, binning):
    """the histogram's value at each sample point, 0 outside the binning (up to three dimensions); a
    [n_bins, d] histogram gives [N, d]"""
    if not isinstance(binning, MultiDimBinning):
        raise ValueError("Binning should be a PISA MultiDimBinning")
    assert binning.num_dims <= 3, "can only do up to 3D at the moment"
    from pisa_amd import kernels as K

    b, cols = _kernel_form(sample, binning)
    to_host = not _is_tensor(flat_hist)
    h = _on_device(flat_hist)
    if h.shape[0] != binning.size:
        h = h.reshape((binning.size,) + tuple(h.shape[binning.num_dims:]))
    out = K.lookup_regular(cols, h, b)
    return out.cpu().numpy() if to_host else out


def resample(weights, old_sample, old_binning, new_sample, new_binning):
    """binned `weights` on `old_binning` moved to `new_binning` (same dimension names): bins of the new binning that
    hold more than one of the old bin centres take the average of those, the others the value of the old bin their own
    centre lies in (translation.py:49-85)"""
    if old_binning.names != new_binning.names:
        raise ValueError("cannot translate betwen %s and %s" % (old_binning, new_binning))
    # numpy's rule whatever the binning, as the reference (it calls its `histogram_np` here)
    averaged = _histogram(old_sample, weights, new_binning, averaged=True, count=False, by_edges=True)
    counts = _histogram(old_sample, weights, new_binning, averaged=False, count=True, by_edges=True)
    values = lookup(new_sample, weights, old_binning)
    if _is_tensor(values):
        import torch

        return torch.where(counts > 1, averaged, values)
    many = counts > 1
    values[many] = averaged[many]
    return values
