"""`Container`, `ContainerSet`, `VirtualContainer`: the data contract between
stages (counterpart of pisa/core/container.py:199-1012).

Same public behaviour as the reference -- variables stored per representation
("events", "log_events", or a `MultiDimBinning`), validity bits, automatic
translation between representations on access, `mark_changed`, aux data,
linking -- but every array is a `DualArray` that can live in host memory, in
HBM, or both:

    container[key]            -> numpy array (copied back from HBM if the device
                                 copy is newer): what third-party services see
    container.device(key)     -> torch device tensor (uploaded if the host copy
                                 is newer): what the HIP stages use
    container[key] = array    -> accepts numpy arrays or device tensors
    container.mark_changed(k) -> host copy is authoritative again

The translations themselves run on the GPU:
    binned -> events   `pisa_hip_lookup_regular`     (container.py:981-1012)
    events -> binned   `pisa_hip_histogram_regular`  (container.py:933-979)
Log dimensions are looked up / histogrammed in ln-space exactly as the
reference regularises them (container.py:950-958, 992-1005); irregular
dimensions are digitised on the host once (np.searchsorted) and binned in index
space.
"""
import re
from collections import defaultdict
from collections.abc import Sequence

import numpy as np

from pisa_amd import FTYPE, _lib
from pisa_amd.core.binning import MultiDimBinning, OneDimBinning
from pisa_amd.core.map import Map, MapSet

__all__ = ["Container", "ContainerSet", "VirtualContainer", "DualArray", "BlockRow"]


def _is_tensor(x):
    return hasattr(x, "is_cuda") and hasattr(x, "data_ptr")


class DualArray:
    """A variable's storage: host numpy and/or device tensor with validity."""

    __slots__ = ("host", "dev", "host_valid", "dev_valid")

    def __init__(self, data, host=None):
        """`data`: numpy array or device tensor.  A strided device view is kept
        as is and only compacted when somebody asks for it (`get_dev`), so
        stages can publish views of one shared table at no cost.  `host`:
        optional host mirror of a device tensor that is already up to date
        (saves a D2H per variable when a stage copied a whole block back)."""
        if _is_tensor(data):
            self.host, self.dev = host, data
            self.host_valid, self.dev_valid = host is not None, True
        else:
            self.host, self.dev = np.asarray(data), None
            self.host_valid, self.dev_valid = True, False

    @property
    def shape(self):
        return tuple(self.dev.shape) if (self.dev_valid and not self.host_valid) else self.host.shape

    def get_host(self):
        if not self.host_valid:
            self.host = self.dev.cpu().numpy()
            self.host_valid = True
        return self.host

    def get_dev(self):
        if not self.dev_valid:
            from pisa_amd import kernels as K

            h = self.host
            self.dev = K.to_device(h, dtype=h.dtype if h.dtype in (np.int32, np.int64) else np.float64)
            self.dev_valid = True
        elif not self.dev.is_contiguous():
            self.dev = self.dev.contiguous()
        return self.dev

    def host_changed(self):
        self.dev_valid = False

    def dev_changed(self):
        self.host_valid, self.dev_valid = False, True


class BlockRow(DualArray):
    """One container's binned output (`weights`, `errors` or `bin_unc2`) as a row of the table of maps the
    fused histogram kernel left in HBM (`core/fastplan.py:DeviceMapBlock`): nothing is computed or copied
    until somebody reads it -- `Pipeline.get_outputs()` turns rows that nobody touched into device-backed Maps
    (`ContainerSet.get_mapset`), whose sum and metric run on the device.  Read through `container[key]` it is
    an ordinary host array (all rows of the block come home in one transfer), through `container.device(key)`
    a device tensor; after that it behaves like any DualArray (in-place edits + `mark_changed` included)."""

    __slots__ = ("block", "row", "which", "n")

    def __init__(self, block, row, which, n):
        self.block, self.row, self.which, self.n = block, row, which, n
        self.host = self.dev = None
        self.host_valid = self.dev_valid = False      # neither materialised yet

    @property
    def pristine(self):
        return not (self.host_valid or self.dev_valid)

    @property
    def shape(self):
        return (self.n,)

    def _source(self):
        if self.block is None:
            raise RuntimeError("this binned output belongs to an earlier evaluation whose maps have been overwritten")
        return self.block

    def get_host(self):
        if not self.host_valid:
            self.host = self.dev.cpu().numpy() if self.dev_valid else self._source().row_host(self.row, self.which)
            self.host_valid = True
        return self.host

    def get_dev(self):
        if not self.dev_valid:
            if self.host_valid:
                return DualArray.get_dev(self)
            self.dev = self._source().row_dev(self.row, self.which)
            self.dev_valid = True
        return self.dev

    def host_changed(self):
        if self.pristine:
            self.get_host()
        self.dev_valid = False

    def dev_changed(self):
        if self.pristine:
            self.get_dev()
        self.host_valid, self.dev_valid = False, True


def regularized(binning, get_column):
    """Linear, regular stand-in for `binning` plus matching sample columns.

    get_column(name, log) must return the (host) event column `name`, or its
    natural logarithm if `log`.  Log-uniform dimensions become linear in ln(x)
    (container.py:950-958); irregular ones are digitised to a bin index
    (utils/hist.py:100-113 does the same for the hist stage)."""
    mins, maxs, nbins, cols = [], [], [], []
    for d in binning:
        if d.is_irregular:
            edges = d.edge_magnitudes
            x = np.asarray(get_column(d.name, False), dtype=FTYPE)
            idx = (np.searchsorted(edges, x, side="right") - 1).astype(FTYPE)
            idx[x == edges[-1]] -= 1  # numpy's closed last bin
            idx[np.isnan(x)] = -1.0
            cols.append(idx)
            mins.append(0.0); maxs.append(float(d.num_bins)); nbins.append(d.num_bins)
        elif d.is_log:
            dom = np.log(d.domain.magnitude)
            cols.append(get_column(d.name, True))
            mins.append(dom[0]); maxs.append(dom[1]); nbins.append(d.num_bins)
        else:
            dom = d.domain.magnitude
            cols.append(get_column(d.name, False))
            mins.append(dom[0]); maxs.append(dom[1]); nbins.append(d.num_bins)
    return _lib.make_binning(mins, maxs, nbins), cols


class Container:
    # counts every store / mark_changed of every container of the process: lets an evaluation
    # plan (core/fastplan.py) notice with one comparison that SOMEBODY wrote a container
    clock = 0
    valid_translation_modes = ("average", "sum")
    sum_mode_keys = ()
    array_representations = ("events", "log_events")

    _unrolled = {}   # (hash(binning), dimension name) -> DualArray of the unrolled bin centres
    writes = 0       # per container: stores / mark_changed of THIS container (class default: none yet)

    def __init__(self, name, representation="events"):
        self.name = name
        self._representation = None
        self._rep_hash = hash(None)
        self._is_map = False
        self.linked = False
        self._aux_data = {}
        self.validity = defaultdict(dict)
        self.translation_modes = {}
        self.data = defaultdict(dict)
        self._representations = {}
        self.precedence = defaultdict(int)
        # deferred operations per key (see pisa_amd/stages/deferred.py)
        self.pending = {}
        self._pending_rep = {}
        self._pending_hash = {}
        # change counter per key: bumped by every store and by `mark_changed`, so that a consumer
        # holding a derived copy (the fused engine's folded flux column) can tell when its
        # source moved -- object identity cannot (in-place edits + mark_changed keep the object)
        self._version = defaultdict(int)
        # operation chains a fused kernel consumed without writing the event-wise result
        # (key -> (ops, representation)); materialised only if somebody reads it there
        self._lazy = {}
        self.representation = representation

    def __repr__(self):
        return "Container containing keys %s" % self.all_keys

    # -- deferred operations ------------------------------------------------
    def touch_pending(self, key):
        """`key` now has pending operations defined in the current representation"""
        self._pending_rep[key] = self._representation
        self._pending_hash[key] = self._rep_hash
        if key not in self.current_data:
            self.current_data[key] = DualArray(np.empty(0, dtype=FTYPE))
        if key not in self.translation_modes:
            self.translation_modes[key] = "average"
        self._invalidate_others(key)

    def _store(self, key, data):
        """(re)place the array of `key` in the current representation, keep pending ops"""
        self.current_data[key] = DualArray(data)
        self._invalidate_others(key)

    def device_raw(self, key):
        return self.current_data[key].get_dev()

    def keep_lazy(self, key, ops, representation):
        """`key` in `representation` is the result of `ops` (a deferred chain that a fused kernel
        has consumed): valid there, computed on first access."""
        self._lazy[key] = (ops, representation)
        self.validity[key][hash(representation)] = True

    def _materialize_lazy(self, key):
        ops, rep = self._lazy.pop(key)
        others = {h: ok for h, ok in self.validity[key].items() if h != hash(rep)}
        version = self._version[key]
        self.pending[key] = list(ops)
        self._pending_rep[key] = rep
        self._flush_pending(key)
        self.validity[key].update(others)   # the values of the other representations still hold
        self._version[key] = version

    def _flush_pending(self, key):
        if self.pending.get(key):
            from pisa_amd.stages import deferred

            keep = self._representation
            try:
                self.representation = self._pending_rep.get(key, keep)
                deferred.materialize(self, key)
            finally:
                self.representation = keep

    # -- representation ------------------------------------------------------
    @property
    def representation(self):
        return self._representation

    @representation.setter
    def representation(self, representation):
        if representation is self._representation and representation is not None:
            return      # (stages switch every container back and forth between the same few objects)
        key = hash(representation)
        if key not in self._representations:
            self._representations[key] = representation
            if isinstance(representation, MultiDimBinning):
                for name in representation.names:
                    self.validity[name][key] = True
            elif isinstance(representation, str):
                if representation not in self.array_representations:
                    raise ValueError("Unknown representation '%s'" % representation)
        self._representation = representation
        self._rep_hash = key      # (hashing a binning walks its dimensions: the stores and look-ups below use this)
        self._is_map = isinstance(representation, MultiDimBinning)
        self.current_data = self.data[key]

    representations = property(lambda self: tuple(self._representations.values()))
    representation_keys = property(lambda self: tuple(self._representations.keys()))

    @property
    def is_map(self):
        return self._is_map

    @property
    def shape(self):
        if self.is_map:
            return self._representation.shape
        if len(self.keys) == 0:
            return None
        return self.current_data[self.keys[0]].shape[0:1]

    @property
    def size(self):
        return int(np.prod(self.shape))

    @property
    def num_dims(self):
        return self._representation.num_dims if self.is_map else 1

    @property
    def keys(self):
        keys = tuple(self.current_data.keys())
        if self.is_map:
            keys += tuple(self._representation.names)
        return keys

    keys_incl_aux_data = property(lambda self: list(self.keys) + list(self._aux_data.keys()))
    all_keys = property(lambda self: list(self.validity.keys()))
    all_keys_incl_aux_data = property(lambda self: self.all_keys + list(self._aux_data.keys()))

    def set_aux_data(self, key, val):
        if key in self.all_keys:
            raise KeyError("Key %s already exsits" % key)
        self._aux_data[key] = val

    # -- validity --------------------------------------------------------------
    def version(self, key):
        """number of times `key` was stored or marked changed (any representation)"""
        return self._version[key]

    def mark_changed(self, key):
        self._version[key] += 1
        Container.clock += 1
        self.writes += 1
        self._lazy.pop(key, None)
        for rep in self.validity[key]:
            self.validity[key][rep] = False
        if key in self.current_data:
            self.mark_valid(key)
            self.current_data[key].host_changed()

    def mark_dev_changed(self, key):
        """`mark_changed` for a kernel that rewrote the DEVICE array of `key` in place (the host
        mirror, if any, is what went stale)"""
        self._version[key] += 1
        Container.clock += 1
        self.writes += 1
        self._lazy.pop(key, None)
        for rep in self.validity[key]:
            self.validity[key][rep] = False
        self.mark_valid(key)
        self.current_data[key].dev_changed()

    def mark_valid(self, key):
        self.validity[key][self._rep_hash] = True

    def _invalidate_others(self, key):
        self._version[key] += 1
        Container.clock += 1
        self.writes += 1
        if self._lazy:
            self._lazy.pop(key, None)
        v = self.validity[key]
        if len(v) > 1 or self._rep_hash not in v:      # (a single entry that IS the current one: nothing to do)
            for rep in v:
                v[rep] = False
        v[self._rep_hash] = True

    # -- access ------------------------------------------------------------------
    def __getitem__(self, key):
        arr = self._get(key)
        return arr.get_host() if isinstance(arr, DualArray) else arr

    def device(self, key):
        """current-representation data as a device tensor"""
        arr = self._get(key)
        if isinstance(arr, DualArray):
            return arr.get_dev()
        from pisa_amd import kernels as K

        return K.to_device(np.asarray(arr, dtype=FTYPE))

    def device_view(self, key):
        """`device(key)` without compacting a strided view that a stage published (for kernels that take a stride)"""
        arr = self._get(key)
        if isinstance(arr, DualArray) and arr.dev_valid:
            return arr.dev
        return self.device(key)

    def _get(self, key):
        if key in self.pending:
            self._flush_pending(key)
        elif key in self._lazy and hash(self._lazy[key][1]) == self._rep_hash:
            self._materialize_lazy(key)
        if self._is_map and key in self._representation._name_set:
            # the bin centres of a binning never change: one host array and one device copy per
            # (binning, dimension) for all containers (a binned flux stage asks for them every evaluation)
            ck = (self._rep_hash, key)
            arr = Container._unrolled.get(ck)
            if arr is None:
                arr = Container._unrolled[ck] = DualArray(self.unroll_binning(key, self._representation))
            return arr
        if key not in self.current_data:
            if key in self.validity:
                self.auto_translate(key)
            elif key in self._aux_data:
                return self._aux_data[key]
            else:
                raise KeyError('Key "%s" not present in Container "%s"' % (key, self.name))
        if not self.validity[key].get(self._rep_hash, False):
            self.auto_translate(key)
        return self.current_data[key]

    def __setitem__(self, key, data):
        if self._is_map and key in self._representation._name_set:
            raise Exception("Cannot add variable %s, as it is a binning dimension" % key)
        self.pending.pop(key, None)  # overwritten
        self._add_data(key, data)
        if key not in self.translation_modes:
            self.translation_modes[key] = "sum" if key in self.sum_mode_keys else "average"
        self._invalidate_others(key)

    def publish(self, key, arr):
        """`container[key] = ...` for a stage that hands over a ready DualArray in the current representation
        (no format checks, no reshaping): the store, the change counters, the validity bits"""
        self.pending.pop(key, None)
        self.current_data[key] = arr
        if key not in self.translation_modes:
            self.translation_modes[key] = "sum" if key in self.sum_mode_keys else "average"
        self._invalidate_others(key)

    def refresh_dev(self, key, dev):
        """`container[key] = dev` where `dev` is the very tensor (or view) published last time and a kernel
        has rewritten it in place: new values, same object -- only the bookkeeping runs"""
        arr = self.current_data.get(key)
        if arr is not None and arr.dev is dev and type(arr) is DualArray:
            self.pending.pop(key, None)
            arr.host_valid, arr.dev_valid = False, True
            self._invalidate_others(key)
        else:
            self[key] = dev

    def set_mirrored(self, key, dev, host):
        """`container[key] = dev` where `host` already holds the same values
        (a stage copied a whole block back in one transfer)."""
        self[key] = dev
        arr = self.current_data[key]
        arr.host, arr.host_valid = np.asarray(host).reshape(tuple(arr.dev.shape)), True

    def _add_data(self, key, data):
        if isinstance(data, Map):
            assert self._rep_hash == hash(data.binning)
            self.current_data[key] = DualArray(data.hist.ravel())
            return
        if isinstance(data, Sequence) and not isinstance(data, np.ndarray) and len(data) == 2 \
                and isinstance(data[0], MultiDimBinning):
            binning, data = data
            assert self._rep_hash == hash(binning)
        if not (isinstance(data, np.ndarray) or _is_tensor(data)):
            raise TypeError("unknown dataformat")
        if self.is_map:
            b = self._representation
            shape = tuple(data.shape)
            if shape[0] != b.size:
                assert shape[: b.num_dims] == b.shape, "Incompatible dimensions"
                tail = shape[b.num_dims:]
                data = data.reshape((b.size,) + tuple(tail))
        else:
            cur = self.shape
            if cur is not None:
                assert tuple(data.shape[:1]) == tuple(cur), "Incompatible dimensions"
        self.current_data[key] = DualArray(data)

    @staticmethod
    def unroll_binning(key, binning):
        grid = binning.meshgrid(entity="weighted_centers", attach_units=False)
        return grid[binning.index(key)].ravel()

    def get_hist(self, key):
        assert self.is_map, "Cannot retrieve hists from non-map data"
        data = self[key]
        binning = self._representation
        full = list(binning.shape) + ([-1] if data.ndim > 1 else [])
        return data.reshape(full), binning

    def get_map(self, key, error=None):
        hist, binning = self.get_hist(key)
        error_hist = np.abs(self.get_hist(error)[0]) if error is not None else None
        assert hist.ndim == binning.num_dims
        return Map(name=self.name, hist=hist, error_hist=error_hist, binning=binning)

    # -- translation ------------------------------------------------------------
    def find_valid_representation(self, key):
        best, prec = None, np.inf
        for h, ok in self.validity[key].items():
            if ok and self.precedence[h] < prec:
                prec, best = self.precedence[h], self._representations[h]
        return best

    def auto_translate(self, key):
        src = self.find_valid_representation(key)
        if src is None:
            raise Exception("No valid representation for %s in container" % key)
        self.translate(key, src)

    def _column(self, name, log):
        """event column (or its log) as a DEVICE tensor; logs are taken on the host
        with numpy like the reference's cached 'log_events' representation"""
        keep = self._representation
        try:
            self.representation = "log_events" if log else "events"
            return self.device(name)
        finally:
            self.representation = keep

    def _host_column(self, name, log):
        keep = self._representation
        try:
            self.representation = "log_events" if log else "events"
            return self[name]
        finally:
            self.representation = keep

    def translate(self, key, src_representation):
        assert hash(src_representation) in self._representations
        dest = self._representation
        if hash(src_representation) == hash(dest):
            return
        from_map = isinstance(src_representation, MultiDimBinning)
        to_map = isinstance(dest, MultiDimBinning)
        mode = self.translation_modes[key]
        if mode not in self.valid_translation_modes:
            raise ValueError("Unknown translation mode for variable '%s': '%s'!" % (key, mode))
        if from_map and to_map:
            if mode == "sum":
                raise NotImplementedError("Map to Map in sum mode needs to integrate over bins.")
            out = self.resample(key, src_representation, dest)
        elif to_map:
            out = self.array_to_binned(key, src_representation, dest, averaged=(mode == "average"))
        elif mode == "sum":
            raise NotImplementedError("Translating %s to %s in 'sum' mode!" % (src_representation, dest))
        elif from_map:
            out = self.binned_to_array(key, src_representation, dest)
        elif src_representation == "events" and dest == "log_events":
            self.representation = "events"
            out = np.log(self[key])
        elif src_representation == "log_events" and dest == "events":
            self.representation = "log_events"
            out = np.exp(self[key])
        else:
            raise NotImplementedError("Translating %s to %s" % (src_representation, dest))
        self.representation = dest
        self._add_data(key, out)
        self.validity[key][hash(dest)] = True
        self.validity[key][hash(src_representation)] = True

    def resample(self, key, src_representation, dest_representation):
        """map -> map (container.py:906-931 -> translation.resample, translation.py:49-85) on the
        GPU: the old bins' centres are histogrammed, weighted with the old values, into the new
        binning; a new bin that received more than one old bin takes their average, any other
        one the old value looked up at its own centre."""
        import torch

        from pisa_amd import kernels as K

        if src_representation.names != dest_representation.names:
            raise ValueError("cannot translate betwen %s and %s" % (src_representation, dest_representation))
        self.representation = src_representation
        weights = self.device(key)
        if weights.dim() != 1:
            raise NotImplementedError("resampling of vector-valued binned data")
        old_centres = {n: self.unroll_binning(n, src_representation) for n in src_representation.names}
        b_new, cols = regularized(dest_representation, lambda n, log: (np.log(old_centres[n]) if log else old_centres[n]))
        cols = [K.to_device(np.asarray(c, dtype=FTYPE)) for c in cols]
        flat = K.histogram_regular(cols, weights, b_new)
        counts = K.histogram_regular(cols, None, b_new)
        new_centres = {n: self.unroll_binning(n, dest_representation) for n in dest_representation.names}
        b_old, cols_new = regularized(src_representation, lambda n, log: (np.log(new_centres[n]) if log else new_centres[n]))
        looked_up = K.lookup_regular([K.to_device(np.asarray(c, dtype=FTYPE)) for c in cols_new], weights, b_old)
        self.representation = dest_representation
        return torch.where(counts > 1, torch.nan_to_num(flat / counts), looked_up)

    def _kernel_form(self, binning):
        """(argument block, device sample columns) with which the kernels histogram into / look up from `binning`, by the
        reference's rule (container.py:948-973, 985-1011): NO irregular dimension -- every logarithmic dimension in ln x, the
        others as they are, fast_histogram's arithmetic on half-open ranges; ANY irregular dimension -- ALL dimensions by
        comparison with their edges in the original coordinates, last edge included (numpy's `histogramdd`)"""
        if binning.is_irregular:
            from pisa_amd.core import translation as T

            return T._kernel_form([self._column(d.name, False) for d in binning], binning, by_edges=True)
        return regularized(binning, self._column)

    def array_to_binned(self, key, src_representation, dest_representation, averaged=True):
        """events -> map (container.py:933-979) on the GPU"""
        from pisa_amd import kernels as K

        assert src_representation in self.array_representations
        self.representation = src_representation
        import torch

        weights = self.device(key)
        if weights.dtype != torch.float64:        # an integer column (`bin_indices`): the kernels read doubles
            weights = weights.to(torch.float64)
        b, cols = self._kernel_form(dest_representation)
        if weights.dim() == 2:

            outs = [K.histogram_regular(cols, weights[:, i].contiguous(), b, averaged=averaged)
                    for i in range(weights.shape[1])]
            return torch.stack(outs, dim=1)
        return K.histogram_regular(cols, weights, b, averaged=averaged)

    def binned_to_array(self, key, src_representation, dest_representation):
        """map -> events nearest-bin lookup (container.py:981-1012) on the GPU"""
        from pisa_amd import kernels as K

        import torch

        self.representation = src_representation
        flat = self.device(key)
        if flat.dtype != torch.float64:
            flat = flat.to(torch.float64)
        self.representation = dest_representation
        b, cols = self._kernel_form(src_representation)
        return K.lookup_regular(cols, flat, b)

    def get_keep_mask(self, keep_criteria):
        assert isinstance(keep_criteria, str)
        for var in self.keys:
            keep_criteria = re.sub(r"\b%s\b" % var, 'self["%s"]' % var, keep_criteria)
        return eval(keep_criteria)  # pylint: disable=eval-used


class VirtualContainer:
    """Linked containers behave as one for binned data (container.py:363-448)."""

    def __init__(self, name, containers):
        self.name = name
        for c in containers:
            if c.linked:
                raise ValueError("Cannot link container %s since it is already linked" % c.name)
            c.linked = True
        self.containers = containers

    def __repr__(self):
        return "VirtualContainer containing %s" % [c.name for c in self]

    def unlink(self):
        for c in self:
            c.linked = False

    def __iter__(self):
        return iter(self.containers)

    def __getitem__(self, key):
        return self.containers[0][key]

    def device(self, key):
        return self.containers[0].device(key)

    def device_view(self, key):
        return self.containers[0].device_view(key)

    def __setitem__(self, key, value):
        for c in self:
            c[key] = value

    def set_aux_data(self, key, val):
        for c in self:
            c.set_aux_data(key, val)

    def mark_changed(self, key):
        # one array is shared by all linked containers (the reference copies it
        # 5 times, container.py:415-423; sharing is equivalent for readers)
        src = self.containers[0]
        arr = src.current_data.get(key)
        for c in self.containers[1:]:
            if arr is not None:
                c.current_data[key] = arr
                if key not in c.translation_modes:
                    c.translation_modes[key] = src.translation_modes.get(key, "average")
        for c in self:
            c.mark_changed(key)

    def mark_valid(self, key):
        for c in self:
            c.mark_valid(key)

    @property
    def representation(self):
        return self.containers[0].representation

    @representation.setter
    def representation(self, representation):
        for c in self:
            c.representation = representation

    shape = property(lambda self: self.containers[0].shape)
    size = property(lambda self: int(np.prod(self.shape)))


class ContainerSet:
    def __init__(self, name, containers=None, representation=None):
        self.name = name
        self.linked_containers = []
        self.containers = []
        for c in containers or []:
            self.add_container(c)
        self._glob_aux_data = {}
        self.representation = representation

    def __repr__(self):
        return "ContainerSet containing %s" % [c.name for c in self]

    @property
    def is_map(self):
        if len(self.containers):
            return self.containers[0].is_map
        return None

    def add_container(self, container):
        if container.name in self.names:
            raise ValueError("container with name %s already exists" % container.name)
        self.containers.append(container)

    @property
    def representation(self):
        return self._representation

    @representation.setter
    def representation(self, representation):
        self._representation = representation
        for c in self:
            c.representation = representation

    names = property(lambda self: [c.name for c in self.containers])

    def get_shared_keys(self, rep_indep=True):
        if len(self.containers) == 0:
            return ()
        return tuple(set.intersection(*[
            set(c.all_keys_incl_aux_data if rep_indep else c.keys_incl_aux_data)
            for c in self.containers]))

    def link_containers(self, key, names):
        link_names = [n for n in names if n in self.names]
        containers = [self[n] for n in link_names]
        if containers:
            self.linked_containers.append(VirtualContainer(key, containers))

    def unlink_containers(self):
        for c in self.linked_containers:
            c.unlink()
        self.linked_containers = []

    def __getitem__(self, key):
        if key in self.names:
            return self.containers[self.names.index(key)]
        for c in self.linked_containers:
            if c.name == key:
                return c
        if key in self._glob_aux_data:
            return self._glob_aux_data[key]
        raise KeyError("No name `%s` in container" % key)

    def __setitem__(self, key, data):
        if key in self.names:
            raise KeyError("`%s` is a container name." % key)
        if key in [c.name for c in self.linked_containers]:
            raise KeyError("`%s` is a linked container name and can't be overwritten." % key)
        self._glob_aux_data[key] = data

    def __iter__(self):
        return iter([c for c in self.containers if not c.linked] + self.linked_containers)

    def _device_mapset(self, key, error):
        """the containers' `key` (and `error`) in the current representation are untouched rows of ONE table
        of maps still in HBM (published by the fused utils.hist stage): a MapSet of device-backed Maps over
        that table -- no copy, no host arrays; None otherwise"""
        if key != "weights" or error not in (None, "errors") or self.linked_containers:
            return None
        block = None
        for i, c in enumerate(self.containers):
            if c.pending or not c._is_map:
                return None
            for k, which in ((key, 0), (error, 1)):
                if k is None:
                    continue
                arr = c.current_data.get(k)
                if (type(arr) is not BlockRow or not arr.pristine or arr.row != i or arr.which != which
                        or not c.validity[k].get(c._rep_hash, False)):
                    return None
                if block is None:
                    block = arr.block
                elif arr.block is not block:
                    return None
        if block is None or not block._live or block._host is not None or len(self.containers) != block.n_rows:
            return None
        from pisa_amd.core.fastplan import DeviceMapSet

        return DeviceMapSet([c.name for c in self.containers], self.containers[0]._representation,
                            block.view(error is not None), self.name)

    def _run_weight_chains(self):
        """Every container carries a pending reset [-> osc] [-> aeff] chain on `weights` in the current
        representation (a pipeline whose stages apply on maps, e.g. osc_example.cfg): ONE launch for all of
        them (`pisa_hip_weight_chain_multi`) instead of up to three per container.  Returns the flat device block
        and the containers' new arrays (for `_prefetch_to_host`), or None if the shape does not apply -- the
        chains are then materialised container by container on access, same values."""
        from pisa_amd import kernels as K
        from pisa_amd.stages import deferred

        if len(self.containers) < 2 or self.linked_containers:
            return None
        chains = [deferred.batch_chain(c) for c in self.containers]
        if any(ch is None for ch in chains):
            return None
        items = []
        for c, (flux, scale) in zip(self.containers, chains):
            items.append((c.device("initial_weights"), None if flux is None else c.device(flux),
                          None if flux is None else c.device_view("prob_e"),
                          None if flux is None else c.device_view("prob_mu"),
                          None if scale is None else c.device("weighted_aeff"), scale))
        block, views = K.weight_chain_multi(items)
        arrays = []
        for c, v in zip(self.containers, views):
            c.pending.pop(deferred.KEY, None)
            c._store(deferred.KEY, v)
            arrays.append(c.current_data[deferred.KEY])
        return block, arrays

    def get_mapset(self, key, error=None):
        fast = self._device_mapset(key, error)
        if fast is not None:
            return fast
        batch = self._run_weight_chains() if key == "weights" else None
        if batch is not None and error is None and (batch[0].numel() * 8) <= (64 << 20):
            # the weights of all containers are one contiguous block: one copy into page-locked memory
            import torch

            block, arrays = batch
            host = torch.empty(block.shape, dtype=block.dtype, pin_memory=True)
            host.copy_(block, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            flat, off = host.numpy(), 0
            for a in arrays:
                n = int(a.dev.numel())
                a.host, a.host_valid = flat[off:off + n], True
                off += n
        self._prefetch_to_host([k for k in (key, error) if k is not None])
        return MapSet(name=self.name, maps=[c.get_map(key, error=error) for c in self])

    def _prefetch_to_host(self, keys):
        """One device->host transfer for all containers' arrays of `keys` that only live in
        HBM (instead of one synchronising copy per container and key)."""
        todo = []
        for c in self.containers:
            for k in keys:
                try:
                    arr = c._get(k)
                except Exception:  # missing key: let get_map raise the proper error
                    return
                if isinstance(arr, DualArray) and not arr.host_valid and arr.dev_valid:
                    todo.append(arr)
        if len(todo) < 2:
            return
        import torch

        # into PINNED host memory (a pageable destination goes through the runtime's staging copy: 0.57 ms for the
        # twelve 200 x 200 maps of osc_example.cfg against 0.15 ms); the block comes from torch's caching host
        # allocator and belongs to the arrays handed out -- maps of an earlier evaluation that somebody still
        # holds keep theirs
        dev = torch.cat([a.get_dev().reshape(-1) for a in todo])
        if dev.numel() * dev.element_size() <= (64 << 20):
            host = torch.empty(dev.shape, dtype=dev.dtype, pin_memory=True)
            host.copy_(dev, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            flat = host.numpy()
        else:   # (page-locked memory is not for map sets of that size)
            flat = dev.cpu().numpy()
        off = 0
        for a in todo:
            n = int(np.prod(a.dev.shape))
            a.host = flat[off:off + n].reshape(tuple(a.dev.shape))
            a.host_valid = True
            off += n

    glob_aux_data_keys = property(lambda self: self._glob_aux_data.keys())
