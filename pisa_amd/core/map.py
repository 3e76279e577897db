"""`Map` / `MapSet`: binned outputs of a pipeline.

Counterparts of pisa/core/map.py restricted to what sits on or next to the hot
path: holding (hist, error_hist) on a `MultiDimBinning`, summing maps with
variance propagation (map.py:1811-1838; the reference does this through
`uncertainties` object arrays, here variances are a second fp64 array), and
`metric` / `metric_total` (map.py:1572-1604, 2956-2978), which run on the GPU
through `pisa_hip_metric` -- there is no host implementation of the metrics in
this package.
"""
from collections import OrderedDict

import numpy as np

from pisa_amd import FTYPE
from pisa_amd.core.binning import MultiDimBinning

__all__ = ["Map", "MapSet", "ALL_METRICS"]

ALL_METRICS = ("llh", "poisson_llh", "chi2", "mod_chi2")


class Map:
    """A Map may be *device backed*: `_lazy = (block, mask)` names rows of a table of maps that
    still lives in HBM (`core/fastplan.py:DeviceMapBlock`); its host arrays are fetched -- for
    all maps of the block in ONE transfer -- the first time anything asks for them.  Sums of
    such maps stay device backed, and `metric` of a device-backed total runs on the device
    without the maps ever travelling to the host."""

    def __init__(self, name, hist, binning, error_hist=None, hash=None, parent_indexer=None,
                 tex=None, full_comparison=False):
        if not isinstance(binning, MultiDimBinning):
            binning = MultiDimBinning(binning)
        hist = np.asarray(hist, dtype=FTYPE)
        if hist.shape != binning.shape:
            raise ValueError("hist shape %s incompatible with binning shape %s"
                             % (hist.shape, binning.shape))
        self.name = name
        self.tex = tex
        self.binning = binning
        self._lazy = None
        self._extra = None
        self._h = hist
        self._v = None
        if error_hist is not None:
            self.set_errors(error_hist)

    @classmethod
    def device_backed(cls, name, binning, block, mask, extra=None):
        """`extra` = (hist, variances or None): host maps added AFTER the device rows (the other
        pipelines of a DistributionMaker); the sum stays device backed and `metric` hands the
        addend to the device tail"""
        m = cls.__new__(cls)
        m.name, m.tex, m.binning = name, None, binning
        m._lazy, m._h, m._v = (block, mask), None, None
        m._extra = extra
        return m

    def _fetch(self):
        block, mask = self._lazy
        self._h, self._v = block.host_sum(mask, self.binning.shape)
        extra = getattr(self, "_extra", None)
        if extra is not None:
            self._h = self._h + extra[0]
            if self._v is not None or extra[1] is not None:
                self._v = (np.zeros_like(self._h) if self._v is None else self._v) + \
                          (np.zeros_like(self._h) if extra[1] is None else extra[1])
            self._extra = None
        self._lazy = None

    # -- copying / pickling: host arrays only.  A device-backed map references the evaluation
    # engine (HBM tensors, ctypes argument blocks): `deepcopy(maker.get_outputs(...))`, a standard
    # pattern with the reference, must neither clone that nor trip over its raw pointers, so the maps
    # are brought to the host first and the copy is an ordinary host map.
    def __getstate__(self):
        if self._lazy is not None:
            self._fetch()
        state = dict(self.__dict__)
        state["_lazy"] = state["_extra"] = None
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)

    def __deepcopy__(self, memo):
        import copy

        m = Map.__new__(Map)
        memo[id(self)] = m
        m.__dict__.update(copy.deepcopy(self.__getstate__(), memo))
        return m

    @property
    def _hist(self):
        if self._lazy is not None:
            self._fetch()
        return self._h

    @_hist.setter
    def _hist(self, h):
        self._h = h

    @property
    def _var(self):
        if self._lazy is not None:
            self._fetch()
        return self._v

    @_var.setter
    def _var(self, v):
        if self._lazy is not None:
            self._fetch()
        self._v = v

    # -- values -------------------------------------------------------------
    hist = property(lambda self: self._hist)
    nominal_values = hist
    shape = property(lambda self: self.binning.shape)

    @property
    def std_devs(self):
        return np.zeros_like(self._hist) if self._var is None else np.sqrt(self._var)

    @property
    def variances(self):
        return np.zeros_like(self._hist) if self._var is None else self._var

    def set_errors(self, error_hist):
        if error_hist is None:
            self._var = None
            return
        e = np.abs(np.asarray(error_hist, dtype=FTYPE))
        assert e.shape == self._hist.shape
        self._var = np.square(e)

    def set_poisson_errors(self):
        self._var = self._hist.copy()

    # -- arithmetic (linear error propagation, uncorrelated) ------------------
    def _new(self, hist, var, name=None):
        m = Map(name or self.name, hist, self.binning, tex=self.tex)
        m._var = var
        return m

    def __add__(self, other):
        if isinstance(other, Map):
            if (self._lazy is not None and other._lazy is not None and self._lazy[0] is other._lazy[0]
                    and not (self._lazy[1] & other._lazy[1])):
                # rows of one device table: the sum stays on the device
                if getattr(self, "_extra", None) is None and getattr(other, "_extra", None) is None:
                    return Map.device_backed("(%s + %s)" % (self.name, other.name), self.binning,
                                             self._lazy[0], self._lazy[1] | other._lazy[1])
            assert other.binning == self.binning
            if self._lazy is not None and other._lazy is None and self._lazy[1] == self._lazy[0].full:
                # complete device template + a host map: the sum stays on the device, the host map is
                # added behind the device rows (same order as the host sum below)
                prev = getattr(self, "_extra", None)
                oh, ov = other._h, other._v
                if prev is not None:
                    ph, pv = prev
                    oh = ph + oh
                    if pv is not None or ov is not None:
                        ov = (np.zeros_like(oh) if pv is None else pv) + (np.zeros_like(oh) if ov is None else ov)
                return Map.device_backed("(%s + %s)" % (self.name, other.name), self.binning,
                                         self._lazy[0], self._lazy[1], extra=(oh, ov))
            var = None
            if self._var is not None or other._var is not None:
                var = self.variances + other.variances
            return self._new(self._hist + other._hist, var, name="(%s + %s)" % (self.name, other.name))
        if np.isscalar(other) and other == 0:  # sum() starts from 0
            return self
        return self._new(self._hist + other, self._var)

    __radd__ = __add__

    def __sub__(self, other):
        if isinstance(other, Map):
            var = None
            if self._var is not None or other._var is not None:
                var = self.variances + other.variances
            return self._new(self._hist - other._hist, var)
        return self._new(self._hist - other, self._var)

    def __mul__(self, other):
        if isinstance(other, Map):
            var = None
            if self._var is not None or other._var is not None:
                var = self.variances * other._hist ** 2 + other.variances * self._hist ** 2
            return self._new(self._hist * other._hist, var)
        other = np.asarray(other, dtype=FTYPE)
        return self._new(self._hist * other, None if self._var is None else self._var * other ** 2)

    __rmul__ = __mul__

    def __truediv__(self, other):
        if isinstance(other, Map):
            with np.errstate(divide="ignore", invalid="ignore"):
                h = self._hist / other._hist
                var = None
                if self._var is not None or other._var is not None:
                    var = (self.variances / other._hist ** 2
                           + other.variances * self._hist ** 2 / other._hist ** 4)
            return self._new(h, var)
        other = np.asarray(other, dtype=FTYPE)
        return self._new(self._hist / other, None if self._var is None else self._var / other ** 2)

    def sum(self, *args, **kwargs):
        return self._hist.sum(*args, **kwargs)

    def fluctuate(self, method, random_state=None):
        """Poisson pseudo-data (map.py:1214-1321; only method='poisson')."""
        if method in (None, "", "none", "asimov"):
            return self._new(self._hist.copy(), self._var)
        if method != "poisson":
            raise ValueError("fluctuate method '%s' not supported" % method)
        rs = random_state if isinstance(random_state, np.random.RandomState) \
            else np.random.RandomState(random_state)
        h = rs.poisson(np.clip(self._hist, 0, None)).astype(FTYPE)
        return self._new(h, h.copy())

    # -- metrics: GPU ---------------------------------------------------------
    def metric(self, expected_values, metric, binned=False):
        if metric not in ALL_METRICS:
            raise ValueError('`metric` "%s" not recognized; use one of %s.' % (metric, ALL_METRICS))
        from pisa_amd import kernels as K

        if isinstance(expected_values, MapSet):
            expected_values = sum(expected_values)
        if isinstance(expected_values, Map) and expected_values._lazy is not None and not binned:
            block, mask = expected_values._lazy
            if expected_values.binning.shape != self._hist.shape:
                raise ValueError("Shape mismatch: actual %s, expected %s"
                                 % (self._hist.shape, expected_values.binning.shape))
            val = block.metric(mask, metric, self._hist, getattr(expected_values, "_extra", None))
            if val is not None:
                return val
        if isinstance(expected_values, Map):
            exp_hist, exp_var = expected_values.hist, expected_values._var
        else:
            exp_hist, exp_var = np.asarray(expected_values, dtype=FTYPE), None
        if exp_hist.shape != self._hist.shape:
            raise ValueError("Shape mismatch: actual %s, expected %s" % (self._hist.shape, exp_hist.shape))
        a = K.to_device(self._hist.ravel())
        e = K.to_device(exp_hist.ravel())
        s2 = K.to_device(exp_var.ravel()) if (exp_var is not None and metric == "mod_chi2") else None
        total, per_bin = K.metric(metric, a, e, s2, per_bin=True)
        if binned:
            return per_bin.cpu().numpy().reshape(self._hist.shape)
        return float(total.item())

    def metric_total(self, expected_values, metric, metric_kwargs=None):
        return self.metric(expected_values, metric)

    def llh(self, expected_values, binned=False):
        return self.metric(expected_values, "llh", binned)

    def poisson_llh(self, expected_values, binned=False):
        return self.metric(expected_values, "poisson_llh", binned)

    def chi2(self, expected_values, binned=False):
        return self.metric(expected_values, "chi2", binned)

    def mod_chi2(self, expected_values, binned=False):
        return self.metric(expected_values, "mod_chi2", binned)

    def __repr__(self):
        if self._lazy is not None:
            return "Map(name=%r, shape=%s, on device)" % (self.name, self.shape)
        return "Map(name=%r, shape=%s, sum=%.6g)" % (self.name, self.shape, self._hist.sum())


class MapSet:
    def __init__(self, maps, name=None, tex=None, hash=None, collate_by_name=True):
        self.maps = list(maps)
        self.name = name
        self.tex = tex
        self.collate_by_name = collate_by_name

    names = property(lambda self: [m.name for m in self.maps])

    def __iter__(self):
        return iter(self.maps)

    def __len__(self):
        return len(self.maps)

    def __getitem__(self, item):
        if isinstance(item, str):
            return self.maps[self.names.index(item)]
        return self.maps[item]

    def __contains__(self, name):
        return name in self.names

    def total(self, name="total"):
        """sum of all maps (what `sum(mapset)` gives), named"""
        out = sum(self.maps)
        if out is self.maps[0] and len(self.maps) == 1:
            out = out._new(out._hist, out._var)
        out.name = name
        return out

    def combine_wildcard(self, expr):
        import fnmatch

        sel = [m for m in self.maps if fnmatch.fnmatch(m.name, expr)]
        if not sel:
            raise ValueError("no map matches '%s'" % expr)
        out = sum(sel)
        out.name = expr
        return out

    def __add__(self, other):
        if isinstance(other, MapSet):
            if self.collate_by_name:
                return MapSet([m + other[m.name] for m in self.maps], name=self.name)
            return MapSet([a + b for a, b in zip(self.maps, other.maps)], name=self.name)
        if np.isscalar(other) and other == 0:
            return self
        return MapSet([m + other for m in self.maps], name=self.name)

    __radd__ = __add__

    def __mul__(self, other):
        return MapSet([m * other for m in self.maps], name=self.name)

    __rmul__ = __mul__

    def fluctuate(self, method, random_state=None):
        rs = random_state if isinstance(random_state, np.random.RandomState) \
            else np.random.RandomState(random_state)
        return MapSet([m.fluctuate(method, rs) for m in self.maps], name=self.name)

    def metric_per_map(self, expected_values, metric):
        out = OrderedDict()
        for m in self.maps:
            exp = expected_values[m.name] if isinstance(expected_values, MapSet) else expected_values
            out[m.name] = m.metric(exp, metric)
        return out

    def metric_total(self, expected_values, metric, metric_kwargs=None):
        return float(np.sum(list(self.metric_per_map(expected_values, metric).values())))

    def __repr__(self):
        return "MapSet(name=%r, maps=%s)" % (self.name, self.names)
