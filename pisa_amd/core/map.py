"""`Map` / `MapSet`: binned outputs of a pipeline.

Counterparts of pisa/core/map.py restricted to what sits on or next to the hot
path: holding (hist, error_hist) on a `MultiDimBinning`, summing maps with
variance propagation (map.py:1811-1838; the reference does this through
`uncertainties` object arrays, here variances are a second fp64 array), and
`metric` / `metric_total` (map.py:1572-1604, 2956-2978), which run on the GPU
through `pisa_hip_metric` -- there is no host implementation of the metrics in
this package.
"""
import fnmatch
import re
from collections import OrderedDict
from collections.abc import Iterable, Mapping, Sequence

import numpy as np

from pisa_amd import FTYPE, HASH_SIGFIGS
from pisa_amd.core.binning import MultiDimBinning, OneDimBinning, _round_sig

__all__ = ["Map", "MapSet", "ALL_METRICS", "rebin"]

# stats.py:43-51 without barlow_llh, generalized_poisson_llh and weighted_chi2 (not built)
ALL_METRICS = ("llh", "poisson_llh", "conv_llh", "mcllh_mean", "mcllh_eff", "chi2", "mod_chi2", "correct_chi2",
               "signed_sqrt_mod_chi2")
FLUCTUATE_METHODS = ("poisson", "scaled_poisson", "gauss", "gauss+poisson")
_ALLCLOSE = dict(rtol=1e-12, atol=np.finfo(FTYPE).eps, equal_nan=True)


def rebin(hist, orig_binning, new_binning, normalize_values=True):
    """`hist` summed into `new_binning`, whose edges are a subset of `orig_binning`'s; the dimensions may come
    in another order (map.py:115-184)."""
    if set(new_binning.basenames) != set(orig_binning.basenames):
        raise ValueError("`new_binning` dimensions' basenames %s do not have 1:1 correspondence (modulo pre/suffixes)"
                         " to original binning dimensions' basenames %s" % (new_binning.basenames, orig_binning.basenames))
    if orig_binning.edges_hash == new_binning.edges_hash:
        return hist
    src, dst = [], []
    for new_idx, new_dim in enumerate(new_binning):
        orig_idx = orig_binning.index(new_dim.name)
        dst.append(new_idx)
        src.append(orig_idx)
        orig_dim = orig_binning.dimensions[orig_idx]
        oe, ne = orig_dim.edge_magnitudes * orig_dim.units.scale, new_dim.edge_magnitudes * new_dim.units.scale
        if normalize_values:
            oe, ne = _round_sig(oe, HASH_SIGFIGS), _round_sig(ne, HASH_SIGFIGS)
        if len(ne) != len(oe) or not np.allclose(ne, oe, **_ALLCLOSE):
            at = np.searchsorted(oe, ne)
            if np.any(at >= len(oe)) or np.any(oe[np.minimum(at, len(oe) - 1)] != ne):
                raise ValueError("the edges of '%s' in the new binning are not a subset of the original edges"
                                 % new_dim.name)
            inside = [slice(None)] * hist.ndim        # a new binning may cover part of the original range only
            inside[orig_idx] = slice(at[0], at[-1])
            hist = np.add.reduceat(hist[tuple(inside)], at[:-1] - at[0], axis=orig_idx)
    return np.moveaxis(hist, source=src, destination=dst)


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
        self.hash = hash
        self.full_comparison = bool(full_comparison)
        self.parent_indexer = parent_indexer
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
        m.hash, m.full_comparison, m.parent_indexer = None, False, None
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

    def __rsub__(self, other):
        return self._new(other - self._hist, self._var)

    def __rtruediv__(self, other):
        other = np.asarray(other, dtype=FTYPE)
        with np.errstate(divide="ignore", invalid="ignore"):
            h = other / self._hist
            var = None if self._var is None else self._var * other ** 2 / self._hist ** 4
        return self._new(h, var)

    def __neg__(self):
        return self._new(-self._hist, self._var)

    def __abs__(self):
        return self._new(np.abs(self._hist), self._var)

    def __pow__(self, other):
        """map ** number or map ** map (map.py:1755-1779); d(a^b) = a^b (b/a da + ln a db)"""
        if isinstance(other, Map):
            b, vb = other._hist, other._var
        else:
            b, vb = np.asarray(other, dtype=FTYPE), None
        with np.errstate(divide="ignore", invalid="ignore"):
            h = self._hist ** b
            var = None
            if self._var is not None or vb is not None:
                var = np.zeros_like(h)
                if self._var is not None:
                    var = var + (b * self._hist ** (b - 1)) ** 2 * self._var
                if vb is not None:
                    var = var + (h * np.log(self._hist)) ** 2 * vb
        return self._new(h, var)

    def _unary(self, f, dfdx):
        with np.errstate(divide="ignore", invalid="ignore"):
            h = f(self._hist)
            var = None if self._var is None else dfdx(self._hist, h) ** 2 * self._var
        return self._new(h, var)

    def sqrt(self):
        return self._unary(np.sqrt, lambda x, y: 0.5 / y)

    def log(self):
        return self._unary(np.log, lambda x, y: 1.0 / x)

    def log10(self):
        return self._unary(np.log10, lambda x, y: 1.0 / (x * np.log(10.0)))

    def round2int(self):
        return self._new(np.rint(self._hist), self._var)

    # -- shape: sums, projections, rebinning, bins ------------------------------
    size = property(lambda self: self.binning.size)

    @property
    def num_entries(self):
        return float(np.nansum(self._hist))

    def _rebuilt(self, hist, var, binning, name=None):
        m = Map(self.name if name is None else name, hist, binning, tex=self.tex,
                full_comparison=self.full_comparison)
        m._var = var
        return m

    def sum(self, axis=None, keepdims=False):
        """NaN-ignoring sum over the dimensions `axis` names (a name, a number, or several); all of them and
        `keepdims=False`: a number, otherwise a Map whose summed dimensions are gone, or -- `keepdims` -- one
        bin wide (map.py:785-827).  Variances add."""
        if axis is None:
            axis = self.binning.names
        if isinstance(axis, (str, int, np.integer, OneDimBinning)):
            axis = [axis]
        idx = tuple(sorted({self.binning.index(d) for d in axis}))
        h = np.nansum(self._hist, axis=idx, keepdims=keepdims)
        v = None if self._var is None else np.nansum(self._var, axis=idx, keepdims=keepdims)
        if len(idx) == self.binning.num_dims and not keepdims:
            return float(h)
        dims = []
        for i, d in enumerate(self.binning.dims):
            if i not in idx:
                dims.append(d)
            elif keepdims:
                dims.append(d.downsample(len(d)))
        return self._rebuilt(h, v, MultiDimBinning(dims))

    def project(self, axis, keepdims=False):
        """everything summed onto the one dimension `axis` (map.py:829-853)"""
        keep = self.binning.index(axis)
        return self.sum(axis=[i for i in range(self.binning.num_dims) if i != keep], keepdims=keepdims)

    def rebin(self, new_binning):
        """the contents summed into `new_binning` (edges a subset of this map's; map.py:856-884)"""
        new_binning = MultiDimBinning(new_binning)
        h = rebin(self._hist, self.binning, new_binning, normalize_values=self.binning.normalize_values)
        v = None if self._var is None else rebin(self._var, self.binning, new_binning,
                                                 normalize_values=self.binning.normalize_values)
        return self._rebuilt(np.ascontiguousarray(h), None if v is None else np.ascontiguousarray(v), new_binning)

    def downsample(self, *args, **kwargs):
        return self.rebin(self.binning.downsample(*args, **kwargs))

    def reorder_dimensions(self, order):
        new_binning = self.binning.reorder_dimensions(order)
        src = [self.binning.index(n) for n in new_binning.names]
        h = np.ascontiguousarray(np.transpose(self._hist, src))
        v = None if self._var is None else np.ascontiguousarray(np.transpose(self._var, src))
        return self._rebuilt(h, v, new_binning)

    def squeeze(self):
        keep = tuple(n for n in self.binning.squeeze().shape)
        return self._rebuilt(self._hist.reshape(keep), None if self._var is None else self._var.reshape(keep),
                             self.binning.squeeze())

    def __getitem__(self, idx):
        """bins by position (ints / slices, one per dimension: the map of those bins); a dimension's bin NAME
        when only one dimension has such a bin (map.py:1203-1227)"""
        if isinstance(idx, str):
            hits = [d.name for d in self.binning if d.bin_names is not None and idx in d.bin_names]
            if len(hits) != 1:
                raise ValueError("bin name '%s' identifies %d dimensions" % (idx, len(hits)))
            return self.slice(**{hits[0]: self.binning[hits[0]].index(idx)})
        new_binning = self.binning[idx]
        if not isinstance(idx, Sequence):
            idx = [idx]
        sel = tuple(slice(i, i + 1 if i != -1 else None) if isinstance(i, (int, np.integer)) else i for i in idx)
        m = self._rebuilt(self._hist[sel], None if self._var is None else self._var[sel], new_binning)
        m.parent_indexer = sel
        return m

    def slice(self, **kwargs):
        """`m.slice(energy=slice(0, 3), pid='track')`: the named dimensions indexed, the others whole (map.py:177-244)"""
        sel = {n: (self.binning[n].index(v) if isinstance(v, str) else v) for n, v in kwargs.items()}
        return self[self.binning.indexer(**sel)]

    def item(self, *args):
        return self._hist.item(*args)

    def iterbins(self):
        """one single-bin Map per bin, in C order"""
        for coord in self.binning.itercoords():
            yield self[tuple(coord)]

    def itercoords(self):
        return self.binning.itercoords()

    def split(self, dim, bin=None, use_basenames=False):  # noqa: A002 (the reference's argument name)
        """the maps of the bins of `dim` (that dimension removed): a MapSet named after the bin names, or the one
        map of `bin` (map.py:1229-1350)"""
        i = self.binning.index(dim, use_basenames=use_basenames)
        d = self.binning.dims[i]
        names = d.bin_names if d.bin_names is not None else ["%s_bin%d" % (d.name, k) for k in range(len(d))]
        rest = self.binning.remove(i)

        def one(k):
            sel = [slice(None)] * self.binning.num_dims
            sel[i] = k
            return self._rebuilt(self._hist[tuple(sel)], None if self._var is None else self._var[tuple(sel)], rest,
                                 name=names[k])
        if bin is not None:
            return one(d.index(bin))
        return MapSet([one(k) for k in range(len(d))], name=self.name)

    # -- comparison -------------------------------------------------------------
    def assert_compat(self, other):
        if not isinstance(other, Map):
            return
        if self.binning != other.binning:
            raise ValueError("Map '%s' and '%s' have different binnings" % (self.name, other.name))

    @property
    def hashable_state(self):
        return OrderedDict([("name", self.name), ("hist", _round_sig(self._hist, HASH_SIGFIGS).tobytes()),
                            ("errors", None if self._var is None else _round_sig(self._var, HASH_SIGFIGS).tobytes()),
                            ("binning", self.binning.hash)])

    def __hash__(self):
        if self.hash is not None:
            return self.hash if isinstance(self.hash, int) else hash(self.hash)
        return hash(tuple(self.hashable_state.values()))

    def __eq__(self, other):
        """a number or an array: every bin equals it; a Map: same binning, values and errors -- or, when both
        carry a `hash` and neither asks for `full_comparison`, the same hash (map.py:1654-1682)"""
        if np.isscalar(other):
            return bool(np.all(self._hist == other))
        if isinstance(other, np.ndarray):
            return bool(np.all(self._hist == other))
        if not isinstance(other, Map):
            return False
        if self.full_comparison != other.full_comparison:
            return False
        if not self.full_comparison and self.hash is not None and other.hash is not None:
            return self.hash == other.hash
        if self.name != other.name or self.binning != other.binning:
            return False
        return bool(np.array_equal(self._hist, other._hist, equal_nan=True)
                    and np.array_equal(self.variances, other.variances, equal_nan=True))

    def __ne__(self, other):
        return not self == other

    def allclose(self, other):
        """values and errors agree to the reference's `ALLCLOSE_KW` (rtol 1e-12; map.py:1876-1890)"""
        if isinstance(other, Map):
            return bool(self.binning == other.binning and np.allclose(self._hist, other._hist, **_ALLCLOSE)
                        and np.allclose(self.std_devs, other.std_devs, **_ALLCLOSE))
        return bool(np.allclose(self._hist, other, **_ALLCLOSE))

    def compare(self, ref):
        """summary numbers of this map against `ref` (map.py:279-351): differences, ratios and fractional
        differences, NaN-aware"""
        assert isinstance(ref, Map) and ref.binning == self.binning
        with np.errstate(divide="ignore", invalid="ignore"):
            diff = self._hist - ref._hist
            ratio = self._hist / ref._hist
            frac = diff / ref._hist
        finite = np.isfinite(frac)
        out = OrderedDict()
        out["diff"], out["fract_diff"], out["ratio"] = diff, frac, ratio
        out["max_abs_diff"] = float(np.nanmax(np.abs(diff))) if diff.size else 0.0
        out["max_abs_fract_diff"] = float(np.max(np.abs(frac[finite]))) if finite.any() else np.nan
        out["total_abs_diff"] = float(np.nansum(np.abs(diff)))
        out["nanmatch"] = bool(np.array_equal(np.isnan(self._hist), np.isnan(ref._hist)))
        out["infmatch"] = bool(np.array_equal(np.isinf(self._hist), np.isinf(ref._hist)))
        return out

    # -- files ------------------------------------------------------------------
    @property
    def serializable_state(self):
        return OrderedDict([("name", self.name), ("hist", self._hist), ("binning", self.binning.serializable_state),
                            ("error_hist", None if self._var is None else self.std_devs), ("hash", self.hash),
                            ("tex", self.tex), ("full_comparison", self.full_comparison)])

    def to_json(self, filename, **kwargs):
        from pisa_amd.utils import jsons

        jsons.to_json(self.serializable_state, filename, **kwargs)

    @classmethod
    def from_json(cls, resource):
        from pisa_amd.utils import jsons

        state = resource if isinstance(resource, Mapping) else jsons.from_json(resource)
        state = dict(state)
        state["binning"] = MultiDimBinning(**state["binning"])
        state["hist"] = np.asarray(state["hist"], dtype=FTYPE)
        return cls(**state)

    def fluctuate(self, method, random_state=None, jumpahead=None):
        """Pseudo-data (map.py:1098-1254): 'poisson', 'scaled_poisson' (Bohm & Zech: same mean and standard
        deviation as this map), 'gauss', 'gauss+poisson'; '' / 'none' / None: a copy.  The draws are scipy's
        `poisson.rvs` / `norm.rvs` on the `RandomState`, over the non-NaN bins in C order: for a given seed the
        numbers are the reference's.  Except for 'scaled_poisson' the errors of the new map are sqrt(this map)."""
        method = "" if method is None else str(method).strip().lower().replace(" ", "")
        if method in ("", "none", "asimov"):
            return self._new(self._hist.copy(), None if self._var is None else self._var.copy())
        if method not in FLUCTUATE_METHODS:
            raise ValueError('Map fluctuation method "%s" not recognized! Valid choices are: %s.'
                             % (method, FLUCTUATE_METHODS))
        if jumpahead is not None:
            raise DeprecationWarning("`jumpahead` is deprecated since it does not result in an independent random"
                                     " sequence, simply use a different seed")
        from scipy.stats import norm, poisson

        rs = random_state if isinstance(random_state, np.random.RandomState) else np.random.RandomState(random_state)
        orig = self._hist
        ok = ~np.isnan(orig)
        vals = np.full(orig.shape, np.nan, dtype=FTYPE)
        with np.errstate(invalid="ignore"):
            var = np.where(ok, orig, np.nan)
            if method == "poisson":
                vals[ok] = poisson.rvs(orig[ok], random_state=rs)
            elif method == "scaled_poisson":
                sigma = self.std_devs.copy()
                zero = orig == 0.0
                if np.any(sigma[ok & ~zero] == 0.0):    # counts without errors: their Poisson expectation
                    sigma[ok] = np.sqrt(orig[ok])
                v = sigma[ok] ** 2
                scale = 1.0 if np.allclose(v, orig[ok], **_ALLCLOSE) else v / orig[ok]
                vals[ok] = poisson.rvs(orig[ok] / scale, random_state=rs)
                vals[ok] *= scale
                vals[zero] = 0.0
                var = np.where(ok, sigma ** 2, np.nan)
            elif method == "gauss+poisson":
                g = np.full(orig.shape, np.nan, dtype=FTYPE)
                g[ok] = norm.rvs(loc=orig[ok], scale=self.std_devs[ok], random_state=rs)
                g = np.clip(g, 0, None)
                vals[ok] = poisson.rvs(g[ok], random_state=rs)
            else:
                vals[ok] = norm.rvs(loc=orig[ok], scale=self.std_devs[ok], random_state=rs)
        return self._new(vals, var)

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
        s2 = None
        if metric in K.VARIANCE_METRICS and (exp_var is not None or metric != "mod_chi2"):
            # (a map without errors has sigma = 0, as `unp.std_devs` of plain numbers)
            s2 = K.to_device(np.zeros(exp_hist.size, dtype=FTYPE) if exp_var is None else exp_var.ravel())
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

    def correct_chi2(self, expected_values, binned=False):
        return self.metric(expected_values, "correct_chi2", binned)

    def signed_sqrt_mod_chi2(self, expected_values, binned=False):
        return self.metric(expected_values, "signed_sqrt_mod_chi2", binned)

    def mcllh_mean(self, expected_values, binned=False):
        return self.metric(expected_values, "mcllh_mean", binned)

    def mcllh_eff(self, expected_values, binned=False):
        return self.metric(expected_values, "mcllh_eff", binned)

    def conv_llh(self, expected_values, binned=False):
        return self.metric(expected_values, "conv_llh", binned)

    def __repr__(self):
        if self._lazy is not None:
            return "Map(name=%r, shape=%s, on device)" % (self.name, self.shape)
        return "Map(name=%r, shape=%s, sum=%.6g)" % (self.name, self.shape, self._hist.sum())


def _python_name(text):
    """a string turned into a valid identifier (utils/format.py make_valid_python_name)"""
    name = re.sub(r"[^0-9a-zA-Z_]", "_", str(text))
    name = re.sub(r"_+", "_", name).strip("_")
    return ("_" + name) if name[:1].isdigit() else name


class MapSet:
    """Ordered set of `Map`s (map.py:1898-2838).  Operations on the set are the operation on every map:
    arithmetic, `sum`, `rebin`, ... and any other Map method or attribute reached through the set
    (`apply_to_maps`); an argument that is itself a MapSet hands each map its partner, by name
    (`collate_by_name`, the default) or by position."""

    def __init__(self, maps, name=None, tex=None, hash=None, collate_by_name=True):
        if isinstance(maps, MapSet):
            name = maps.name if name is None else name
            tex = maps.tex if tex is None else tex
            maps = maps.maps
        made = []
        for m in maps:
            if isinstance(m, Mapping):
                m = Map.from_json(m)
            if not isinstance(m, Map):
                raise TypeError("a MapSet holds Maps; got %s" % type(m).__name__)
            made.append(m)
        self.__dict__["maps"] = made
        self.name = name
        self.tex = tex
        self.collate_by_name = collate_by_name
        self.collate_by_num = not collate_by_name
        if hash is not None:
            self.hash = hash

    names = property(lambda self: [m.name for m in self.maps])
    hashes = property(lambda self: [m.hash for m in self.maps])

    @property
    def hash(self):
        """one number for the set when every map carries a hash, else None (map.py:2378-2394)"""
        hashes = self.hashes
        if not hashes or any(h is None for h in hashes):
            return None
        return hashes[0] if all(h == hashes[0] for h in hashes) else hash(tuple(hashes))

    @hash.setter
    def hash(self, val):
        for m in self.maps:
            m.hash = val

    def hash_maps(self, map_names=None):
        return [hash(m) for m in self.maps if map_names is None or m.name in map_names]

    def __iter__(self):
        return iter(self.maps)

    def __len__(self):
        return len(self.maps)

    def __contains__(self, name):
        return name in self.names

    def index(self, x):
        """position of a map given by name, by number or as the Map itself (map.py:2054-2085)"""
        if isinstance(x, (int, np.integer)) and not isinstance(x, bool):
            if -len(self) <= x < len(self):
                return int(x) % len(self)
            raise ValueError("Map index %d is out of range (%d maps)" % (x, len(self)))
        if isinstance(x, Map):
            for i, m in enumerate(self.maps):
                if m is x:
                    return i
            x = x.name
        if isinstance(x, str) and x in self.names:
            return self.names.index(x)
        raise ValueError('Could not find map "%s" among maps %s' % (x, self.names))

    def find_map(self, value):
        return self.maps[self.index(value)]

    def pop(self, *args):
        """remove and return a map by name / number / itself (the last without argument)"""
        if len(args) > 1:
            raise ValueError("`pop` takes 0 or 1 argument; %d passed" % len(args))
        return self.maps.pop(self.index(args[0]) if args else -1)

    def collate_with_names(self, vals):
        return OrderedDict(zip(self.names, vals))

    def __getitem__(self, item):
        """a map by name or position, several by a slice (a MapSet), or -- a tuple with one entry per dimension --
        those bins of every map (map.py:2572-2608)"""
        if isinstance(item, str):
            return self.find_map(item)
        if isinstance(item, (int, np.integer)):
            return self.maps[item]
        if isinstance(item, slice):
            return self._like(self.maps[item])
        if isinstance(item, Iterable):
            return self._like([m[tuple(item)] for m in self.maps])
        raise TypeError("getitem does not support `item` of type %s" % type(item))

    def _like(self, maps):
        return MapSet(maps, name=self.name, tex=self.tex, collate_by_name=self.collate_by_name)

    def __getattr__(self, attr):
        if attr.startswith("__") or "maps" not in self.__dict__:
            raise AttributeError(attr)
        if attr in self.names:
            return self[attr]
        return self.apply_to_maps(attr)

    def apply_to_maps(self, attr, *args):
        """attribute `attr` of every map; if it is a method, called with `args`, a MapSet among them replaced
        by the called map's partner.  All results Maps: a MapSet; all None: None; else {name: result}
        (map.py:2462-2545)"""
        name = getattr(attr, "__name__", attr)
        missing = [m.name for m in self.maps if not hasattr(m, name)]
        if missing:
            raise AttributeError('Maps %s (%d of %d maps in set) do not have attribute "%s"'
                                 % (", ".join(missing), len(missing), len(self), name))
        vals = [getattr(m, name) for m in self.maps]
        if all(callable(v) for v in vals):
            def partner(arg, num, m):
                if isinstance(arg, MapSet):
                    return arg[m.name] if self.collate_by_name else arg[num]
                if isinstance(arg, (list, tuple)) and any(isinstance(a, MapSet) for a in arg):
                    return [partner(a, num, m) for a in arg]
                return arg
            vals = [v(*[partner(a, i, m) for a in args]) for i, (v, m) in enumerate(zip(vals, self.maps))]
        if vals and all(isinstance(v, Map) for v in vals):
            return self._like(vals)
        if all(v is None for v in vals):
            return None
        return self.collate_with_names(vals)

    # -- combinations -------------------------------------------------------------
    def total(self, name="total"):
        """sum of all maps (what `sum(mapset)` gives), named"""
        out = sum(self.maps)
        if out is self.maps[0] and len(self.maps) == 1:
            out = out._new(out._hist, out._var)
        out.name = name
        return out

    def _combine(self, exprs, matches):
        scalar = isinstance(exprs, str) or hasattr(exprs, "pattern")
        out = []
        for expr in ([exprs] if scalar else exprs):
            sel = [m for m in self.maps if matches(expr, m.name)]
            if not sel:
                raise ValueError('No map names match "%s"' % getattr(expr, "pattern", expr))
            if len(sel) > 1:
                m = sum(sel[1:], sel[0])
                m.name = _python_name(getattr(expr, "pattern", expr)) or "combined"
                m.tex = None
            else:
                m = sel[0]._new(sel[0]._hist, sel[0]._var)
            out.append(m)
        return out[0] if scalar else self._like(out)

    def combine_re(self, regexes):
        """maps whose names match (`re.match`) a regex, added: a Map for one regex, a MapSet for several
        (map.py:2116-2234).  A sum is named after its expression (the reference names sums of flavour /
        interaction maps after the `NuFlavIntGroup` they form; that algebra is not part of this build)."""
        return self._combine(regexes, lambda rx, name: re.match(rx, name) is not None)

    def combine_wildcard(self, expressions):
        """the same with shell wildcards (`fnmatch`; map.py:2236-2331): `combine_wildcard('*_cc')`"""
        return self._combine(expressions, lambda ex, name: fnmatch.fnmatch(name, ex))

    # -- arithmetic: the operation on every map ----------------------------------------
    def __add__(self, other):
        if np.isscalar(other) and other == 0:
            return self
        return self.apply_to_maps("__add__", other)

    __radd__ = __add__

    def __sub__(self, other):
        return self.apply_to_maps("__sub__", other)

    def __rsub__(self, other):
        return self.apply_to_maps("__rsub__", other)

    def __mul__(self, other):
        return self.apply_to_maps("__mul__", other)

    __rmul__ = __mul__

    def __truediv__(self, other):
        return self.apply_to_maps("__truediv__", other)

    def __rtruediv__(self, other):
        return self.apply_to_maps("__rtruediv__", other)

    def __pow__(self, other):
        return self.apply_to_maps("__pow__", other)

    def __neg__(self):
        return self.apply_to_maps("__neg__")

    def __abs__(self):
        return self.apply_to_maps("__abs__")

    def sqrt(self):
        return self.apply_to_maps("sqrt")

    def log(self):
        return self.apply_to_maps("log")

    def log10(self):
        return self.apply_to_maps("log10")

    def sum(self, *args, **kwargs):
        return self._per_map("sum", *args, **kwargs)

    def project(self, axis, keepdims=False):
        return self._per_map("project", axis, keepdims=keepdims)

    def reorder_dimensions(self, order):
        return self._per_map("reorder_dimensions", order)

    def squeeze(self):
        return self._per_map("squeeze")

    def rebin(self, *args, **kwargs):
        return self._per_map("rebin", *args, **kwargs)

    def downsample(self, *args, **kwargs):
        return self._per_map("downsample", *args, **kwargs)

    def _per_map(self, method, *args, **kwargs):
        vals = [getattr(m, method)(*args, **kwargs) for m in self.maps]
        return self._like(vals) if all(isinstance(v, Map) for v in vals) else self.collate_with_names(vals)

    def set_poisson_errors(self):
        for m in self.maps:
            m.set_poisson_errors()

    def fluctuate(self, method, random_state=None, jumpahead=None):
        """every map fluctuated, all from ONE random state in the order of the maps (map.py:2776-2790)"""
        rs = random_state if isinstance(random_state, np.random.RandomState) \
            else np.random.RandomState(random_state)
        return self._like([m.fluctuate(method, rs, jumpahead=jumpahead) for m in self.maps])

    # -- comparison ---------------------------------------------------------------
    def __eq__(self, other):
        if not isinstance(other, MapSet) or len(other) != len(self):
            return False
        return all(a == b for a, b in zip(self.maps, other.maps))

    def __ne__(self, other):
        return not self == other

    __hash__ = object.__hash__

    def allclose(self, other):
        return isinstance(other, MapSet) and len(other) == len(self) and \
            all(a.allclose(b) for a, b in zip(self.maps, other.maps))

    def compare(self, ref):
        assert isinstance(ref, MapSet) and len(ref) == len(self)
        return OrderedDict((m.name, m.compare(ref[m.name] if self.collate_by_name else r))
                           for m, r in zip(self.maps, ref.maps))

    # -- metrics ------------------------------------------------------------------
    def metric_per_map(self, expected_values, metric):
        out = OrderedDict()
        for i, m in enumerate(self.maps):
            if isinstance(expected_values, MapSet):
                exp = expected_values[m.name] if self.collate_by_name else expected_values[i]
            else:
                exp = expected_values
            # 'binned_<metric>': the per-bin values instead of their sum (map.py:3121-3130)
            out[m.name] = m.metric(exp, metric[7:], binned=True) if metric.startswith("binned_") else m.metric(exp, metric)
        return out

    def metric_total(self, expected_values, metric, metric_kwargs=None):
        return float(np.sum(list(self.metric_per_map(expected_values, metric).values())))

    def chi2_per_map(self, expected_values):
        return self.metric_per_map(expected_values, "chi2")

    def chi2_total(self, expected_values):
        return self.metric_total(expected_values, "chi2")

    def llh_per_map(self, expected_values):
        return self.metric_per_map(expected_values, "llh")

    def llh_total(self, expected_values):
        return self.metric_total(expected_values, "llh")

    # -- files ----------------------------------------------------------------------
    @property
    def serializable_state(self):
        return OrderedDict([("maps", [m.serializable_state for m in self.maps]), ("name", self.name),
                            ("tex", self.tex), ("collate_by_name", self.collate_by_name)])

    def to_json(self, filename, **kwargs):
        from pisa_amd.utils import jsons

        jsons.to_json(self.serializable_state, filename, **kwargs)

    @classmethod
    def from_json(cls, resource):
        from pisa_amd.utils import jsons

        state = resource if isinstance(resource, Mapping) else jsons.from_json(resource)
        return cls(**state)

    def __repr__(self):
        return "MapSet(name=%r, maps=%s)" % (self.name, self.names)
