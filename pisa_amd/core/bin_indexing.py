"""Bin number of every event in a binning given by its EDGES (counterpart of pisa/core/bin_indexing.py:104-158):
C-order flat index, -1 when any coordinate is below its first edge (or NaN), else `binning.size` when any is above
its last edge; the last edge belongs to the last bin.  One launch of `pisa_hip_lookup_indices` over device columns;
host arrays are uploaded and the answer comes back as a numpy int64 array, device tensors stay on the device."""
import numpy as np

from pisa_amd import FTYPE
from pisa_amd.core.binning import MultiDimBinning

__all__ = ["lookup_indices"]


def lookup_indices(sample, binning):
    from pisa_amd import kernels as K
    from pisa_amd.core.translation import _is_tensor, _on_device

    binning = MultiDimBinning(binning)
    if len(sample) != binning.num_dims:
        raise ValueError("`binning` has %d dimension(s), but `sample` contains %d arrays (so represents %d dimensions)"
                         % (binning.num_dims, len(sample), len(sample)))
    if binning.num_dims not in (1, 2, 3):
        raise NotImplementedError("binning must have num_dims in [1, 2, 3]; got %d" % binning.num_dims)
    on_device = all(_is_tensor(s) for s in sample)
    cols = [_on_device(s) for s in sample]
    edges = [K.to_device(np.ascontiguousarray(d.edge_magnitudes, dtype=FTYPE)) for d in binning]
    idx = K.lookup_indices(cols, edges)
    return idx if on_device else idx.cpu().numpy()
