"""Binning value types: `OneDimBinning`, `MultiDimBinning`, `VarBinning`.

Host-side counterparts of pisa/core/binning.py (OneDimBinning :142, MultiDimBinning :1484, VarBinning
:3043).  What defines NUMBERS on the hot path is reproduced operation for operation: edge generation
(`np.logspace` / `np.linspace`, binning.py:416-428), `weighted_centers` (geometric mean for log bins,
:901-911), the regularity tests (:1047-1115), C-order flattening / meshgrid (:2669-2711), oversampling and
bin volumes (used by the KDE stage).  The rest is the reference's value-type behaviour: equality and hashes
on values normalised to base units and `HASH_SIGFIGS` figures, indexing / slicing by bin, down- and
oversampling, unit conversion, compatibility, iteration over bins, JSON states.  Plotting is out of scope.
"""
import hashlib
import itertools
from collections import OrderedDict, namedtuple
from collections.abc import Iterable, Mapping, Sequence

import numpy as np

from pisa_amd import FTYPE, HASH_SIGFIGS
from pisa_amd.core.units import Quantity, Unit, ureg

__all__ = ["OneDimBinning", "MultiDimBinning", "VarBinning", "is_binning"]

_ALLCLOSE = dict(rtol=1e-12, atol=np.finfo(FTYPE).eps, equal_nan=True)


def _mag(x):
    return x.magnitude if isinstance(x, Quantity) else x


def _round_sig(values, sigfigs):
    """`values` rounded to `sigfigs` significant figures (utils/comparisons.py normQuant)"""
    return np.array([float("%.*e" % (sigfigs - 1, v)) for v in np.ravel(values)], dtype=np.float64)


def _digest(*parts):
    h = hashlib.md5()
    for p in parts:
        h.update(p if isinstance(p, bytes) else repr(p).encode())
    return int.from_bytes(h.digest()[:8], "little", signed=True)


def _as_unit(units):
    if units is None or isinstance(units, Unit):
        return units
    if isinstance(units, Quantity):
        return units.units
    return ureg.parse_units(units)


def is_binning(x):
    return isinstance(x, (OneDimBinning, MultiDimBinning, VarBinning))


class OneDimBinning:
    """Bins of one dimension: name, edges with units, lin / log character, optional bin names."""

    def __init__(self, name, tex=None, bin_edges=None, units=None, domain=None, num_bins=None,
                 is_lin=None, is_log=None, bin_names=None):
        if not isinstance(name, str):
            raise TypeError('`name` must be a string; got "%s".' % type(name))
        if bin_names is not None:
            if isinstance(bin_names, str):
                bin_names = (bin_names,)
            if not (isinstance(bin_names, Iterable) and all(isinstance(n, str) and n for n in bin_names)):
                raise ValueError("`bin_names` must either be None or an iterable of nonzero-length strings.")
            bin_names = tuple(bin_names)
        if bin_edges is not None and domain is not None:
            raise ValueError("Both `domain` and `bin_edges` are specified.")
        if is_lin is not None and is_log is not None and bool(is_lin) == bool(is_log):
            raise ValueError("`is_log=%s` contradicts `is_lin=%s`" % (is_log, is_lin))
        if is_log is None:
            is_log = False if is_lin is None else not is_lin
        self._is_log = bool(is_log)
        self._name = name
        self._tex = tex
        units = _as_unit(units)
        edges = None
        if bin_edges is not None:
            if isinstance(bin_edges, Quantity):
                if units is None:
                    units = bin_edges.units
                elif bin_edges.units.dims != units.dims:
                    raise ValueError("`bin_edges` units %s are incompatible with units %s." % (bin_edges.units, units))
                bin_edges = bin_edges.to(units).magnitude
            elif isinstance(bin_edges, Iterable) and not isinstance(bin_edges, np.ndarray):
                bin_edges = list(bin_edges)
                if bin_edges and isinstance(bin_edges[0], Quantity):
                    units = units or bin_edges[0].units
                    bin_edges = [e.m_as(units) for e in bin_edges]
            edges = np.array(bin_edges, dtype=FTYPE)
        if domain is not None:
            if isinstance(domain, Quantity):
                if units is None:
                    units = domain.units
                elif domain.units.dims != units.dims:
                    raise ValueError("`domain` units %s are incompatible with units %s." % (domain.units, units))
                domain = domain.to(units).magnitude
            elif isinstance(domain, Sequence) and len(domain) == 2 and isinstance(domain[0], Quantity):
                assert isinstance(domain[1], Quantity) and domain[0].units.dims == domain[1].units.dims
                units = units or domain[0].units
                domain = [domain[0].to(units).magnitude, domain[1].to(units).magnitude]
            domain = (float(domain[0]), float(domain[1]))
        if units is None:
            units = ureg.dimensionless
        if edges is None:
            if num_bins is None or domain is None:
                raise ValueError("If not specifying bin edges explicitly, `domain` and `num_bins`"
                                 " must be specified (and optionally set `is_log=True`).")
            if self._is_log:
                edges = np.logspace(np.log10(domain[0]), np.log10(domain[1]), num_bins + 1,
                                    dtype=FTYPE)
            else:
                edges = np.linspace(domain[0], domain[1], num_bins + 1, dtype=FTYPE)
        if not self.is_binning_ok(edges):
            raise ValueError("bin edges must be strictly increasing and define >= 1 bin")
        if num_bins is not None:
            assert num_bins == len(edges) - 1, "%s, %s" % (num_bins, edges)
        if bin_names is not None and len(bin_names) != len(edges) - 1:
            raise ValueError("There are %d bins, so there must be %d `bin_names` (or None) provided; got %d: %s."
                             % (len(edges) - 1, len(edges) - 1, len(bin_names), bin_names))
        edges.setflags(write=False)
        self._edges = edges
        self._units = units
        self._bin_names = bin_names
        self._normalize_values = True
        self._is_irregular = None
        self._hash = None
        self._edges_hash = None

    # ---- basic attributes
    name = property(lambda self: self._name)
    units = property(lambda self: self._units)
    bin_names = property(lambda self: self._bin_names)
    is_log = property(lambda self: self._is_log)
    is_lin = property(lambda self: not self._is_log)
    num_bins = property(lambda self: len(self._edges) - 1)
    size = num_bins
    shape = property(lambda self: (self.num_bins,))
    edge_magnitudes = property(lambda self: self._edges)
    bin_edges = property(lambda self: Quantity(self._edges, self._units))
    domain = property(lambda self: Quantity(np.array([self._edges[0], self._edges[-1]]), self._units))
    range = property(lambda self: Quantity(self._edges[-1] - self._edges[0], self._units))

    @property
    def tex(self):
        return self._tex

    @tex.setter
    def tex(self, val):
        assert val is None or isinstance(val, str)
        self._tex = val

    @property
    def label(self):
        """TeX axis label, units included unless dimensionless (binning.py:775-788)"""
        name_tex = r"{\rm %s}" % self._name.replace("_", r"\_") if self._tex is None else self._tex
        if self._units.dimensionless and self._units.scale == 1.0:
            return name_tex
        return name_tex + r" \; \left( \mathrm{%s} \right)" % str(self._units).replace("**", "^").replace("*", r"\cdot")

    def __len__(self):
        return self.num_bins

    @property
    def basename(self):
        return self._name.replace("true_", "").replace("reco_", "")

    @property
    def normalize_values(self):
        return self._normalize_values

    @normalize_values.setter
    def normalize_values(self, b):
        assert isinstance(b, bool)
        if b != self._normalize_values:
            self._normalize_values = b
            self.rehash()

    def _new(self, **changes):
        """a binning of its own with this one's attributes, `changes` applied"""
        kw = dict(name=self._name, tex=self._tex, bin_edges=self._edges, units=self._units, is_log=self._is_log,
                  bin_names=self._bin_names)
        kw.update(changes)
        new = OneDimBinning(**kw)
        new._normalize_values = self._normalize_values
        return new

    # ---- regularity
    @staticmethod
    def is_binning_ok(bin_edges):
        """two or more edges, strictly increasing (binning.py:1116-1137)"""
        e = np.asarray(_mag(bin_edges))
        return bool(len(e) >= 2 and not np.any(np.diff(e) <= 0))

    @staticmethod
    def is_bin_spacing_log_uniform(bin_edges):
        e = np.asarray(_mag(bin_edges), dtype=FTYPE)
        if len(e) < 3:
            raise ValueError("%d bin edge(s) passed; require at least 3" % len(e))
        if np.any(e[:-1] == 0) or not np.all(np.isfinite(e)):
            return False
        ratio = e[1:] / e[:-1]
        return bool(np.allclose(ratio, ratio[0], **_ALLCLOSE))

    @staticmethod
    def is_bin_spacing_lin_uniform(bin_edges):
        e = np.asarray(_mag(bin_edges), dtype=FTYPE)
        if len(e) == 1:
            raise ValueError("Single bin edge passed; require at least 2")
        if not np.all(np.isfinite(e)):
            return False
        if len(e) == 2:
            return True
        d = np.diff(e)
        return bool(np.allclose(d, d[0], **_ALLCLOSE))

    @property
    def is_irregular(self):
        """not uniform in the space the binning lives in (binning.py:879-890)"""
        if self._is_irregular is None:
            if self.num_bins == 1:
                self._is_irregular = False
            elif self._is_log:
                self._is_irregular = not self.is_bin_spacing_log_uniform(self._edges)
            else:
                self._is_irregular = not self.is_bin_spacing_lin_uniform(self._edges)
        return self._is_irregular

    # ---- derived arrays
    @property
    def midpoints(self):
        return Quantity((self._edges[:-1] + self._edges[1:]) / 2.0, self._units)

    @property
    def weighted_centers(self):
        if self._is_log:
            return Quantity(np.sqrt(self._edges[:-1] * self._edges[1:]), self._units)
        return self.midpoints

    @property
    def bin_widths(self):
        return Quantity(np.abs(np.diff(self._edges)), self._units)

    @property
    def weighted_bin_widths(self):
        if self._is_log:
            return Quantity(np.log(self._edges[1:] / self._edges[:-1]), ureg.dimensionless)
        return self.bin_widths

    @property
    def inbounds_criteria(self):
        """a boolean expression in the dimension's name: inside the binning's limits (binning.py:993-1009)"""
        return "(%s >= %.15e) & (%s <= %.15e)" % (self._name, self._edges.min(), self._name, self._edges.max())

    # ---- resampling, units
    def oversample(self, factor):
        """`factor` sub-bins per bin, uniform in the binning's own space (binning.py:1307-1365)."""
        if int(factor) != float(factor):
            raise ValueError("Floating point `factor` is non-integral.")
        factor = int(factor)
        if factor < 1:
            raise ValueError("`factor` must be >= 1; got %d" % factor)
        if factor == 1:
            return self
        e = self._edges
        parts = []
        for lo, hi in zip(e[:-1], e[1:]):
            if self._is_log:
                sub = np.logspace(np.log10(lo), np.log10(hi), factor + 1, dtype=FTYPE)
            else:
                sub = np.linspace(lo, hi, factor + 1, dtype=FTYPE)
            sub[0], sub[-1] = lo, hi
            parts.append(sub[:-1])
        parts.append(np.array([e[-1]]))
        return self._new(bin_edges=np.concatenate(parts), bin_names=None)

    def downsample(self, factor):
        """every `factor`-th edge; `factor` divides the number of bins (binning.py:1270-1319).  Bin names are
        not carried over."""
        if int(factor) != float(factor):
            raise ValueError("Floating point `factor` is non-integral.")
        factor = int(factor)
        if factor == 1:
            return self
        if factor < 1 or factor > self.num_bins:
            raise ValueError("`factor` %d is out of range; must be >= 1 and <= number of bins (%d)."
                             % (factor, self.num_bins))
        if self.num_bins % factor != 0:
            raise ValueError("`factor` %d does not evenly divide number of bins (%d)." % (factor, self.num_bins))
        return self._new(bin_edges=self._edges[::factor], bin_names=None)

    def to(self, units):
        """the same bins expressed in `units` (equal to this binning: equality is on normalised values)"""
        units = ureg.dimensionless if units is None or units == "" else _as_unit(units)
        if units.dims != self._units.dims:
            from pisa_amd.core.units import DimensionalityError

            raise DimensionalityError("cannot convert '%s' binning from %s to %s" % (self._name, self._units, units))
        if units is self._units or (units == self._units and str(units) == str(self._units)):
            return self
        return self._new(bin_edges=self.bin_edges.to(units).magnitude, units=units)

    def ito(self, units):
        new = self.to(units)
        if new is not self:
            self._edges, self._units = new._edges, new._units
            self._is_irregular = None
            self.rehash()

    @property
    def basename_binning(self):
        """the same bins under the dimension's basename, without tex (binning.py:1194-1199)"""
        return self._new(name=self.basename, tex=None)

    @property
    def finite_binning(self):
        """infinite outer edges replaced by the largest finite doubles (binning.py:1201-1209)"""
        fi = np.finfo(FTYPE)
        return self._new(bin_edges=np.clip(self._edges, fi.min, fi.max))

    # ---- comparison
    def _norm_edges(self):
        e = self._edges * self._units.scale
        return _round_sig(e, HASH_SIGFIGS) if self._normalize_values else e

    @property
    def edges_hash(self):
        if self._edges_hash is None:
            self._edges_hash = _digest(np.ascontiguousarray(self._norm_edges()).tobytes(), self._units.dims)
        return self._edges_hash

    @property
    def serializable_state(self):
        return OrderedDict([("name", self._name), ("bin_edges", self._edges), ("units", str(self._units)),
                            ("is_log", self.is_log), ("is_lin", self.is_lin), ("bin_names", self._bin_names),
                            ("tex", self._tex)])

    @property
    def hashable_state(self):
        return OrderedDict([("name", self._name), ("edges_hash", self.edges_hash), ("is_log", self.is_log),
                            ("is_lin", self.is_lin), ("bin_names", self._bin_names)])

    @property
    def normalized_state(self):
        return OrderedDict([("name", self._name), ("bin_edges", self._norm_edges()), ("is_log", self.is_log),
                            ("is_lin", self.is_lin), ("bin_names", self._bin_names)])

    @property
    def hash(self):
        if self._hash is None:
            self._hash = _digest(*self.hashable_state.items())
        return self._hash

    def rehash(self):
        self._hash = self._edges_hash = None
        return self.hash

    def __hash__(self):
        return self.hash

    def __eq__(self, other):
        return isinstance(other, OneDimBinning) and self.hash == other.hash

    def __ne__(self, other):
        return not self == other

    def is_compat(self, other):
        """this binning's edges are a subset of `other`'s (one can downsample `other` to it), same name and
        dimensionality (binning.py:1140-1187)"""
        if not isinstance(other, OneDimBinning) or self._name != other._name or self._units.dims != other._units.dims:
            return False
        if self._normalize_values:
            mine, theirs = set(self._norm_edges().tolist()), set(_round_sig(other._edges * other._units.scale,
                                                                            HASH_SIGFIGS).tolist())
        else:
            mine, theirs = set((self._edges * self._units.scale).tolist()), \
                set((other._edges * other._units.scale).tolist())
        return mine.issubset(theirs)

    def assert_compat(self, other):
        if not self.is_compat(other):
            raise AssertionError("incompatible %s binning" % self._name)

    # ---- bins
    def index(self, x):
        """position of the bin `x` names: an int in range, or a bin name (binning.py:602-637)"""
        if isinstance(x, str):
            if self._bin_names is not None and x in self._bin_names:
                return self._bin_names.index(x)
        elif isinstance(x, (int, np.integer)) and not isinstance(x, bool):
            if 0 <= x < len(self):
                return int(x)
        else:
            raise TypeError("`x` must be either int or string; got %s instead." % type(x))
        raise ValueError('Bin corresponding to "%s" could not be located. Specify an int in %s%s.'
                         % (x, [0, len(self) - 1],
                            "" if self._bin_names is None else " or a valid bin name in %s" % (self._bin_names,)))

    def __contains__(self, x):
        try:
            self.index(x)
        except (ValueError, TypeError):
            return False
        return True

    def iterbins(self):
        return (self[i] for i in range(len(self)))

    def iteredgetuples(self):
        e = self._edges
        return ((float(a), float(b)) for a, b in zip(e[:-1], e[1:]))

    def __iter__(self):
        return self.iterbins()

    def __getitem__(self, index):
        """the binning of the bin(s) `index` picks: an int, a bin name, a slice, `...`, or a sequence of
        adjacent positions / names (binning.py:1387-1470).  The bins must be contiguous and at least one."""
        if index is Ellipsis:
            return self
        n = len(self)
        if isinstance(index, str):
            index = self.index(index)
        if isinstance(index, (int, np.integer)):
            if index < -n or index >= n:
                raise ValueError("Bin index %d is out of range (%d bins)" % (index, n))
            idxs = [int(index) % n]
        elif isinstance(index, slice):
            if index.step not in (None, 1):
                raise ValueError("Only contiguous bins can be selected; got a step of %s" % (index.step,))
            idxs = list(range(n)[index])
        elif isinstance(index, Iterable):
            idxs = [self.index(i) if isinstance(i, str) else int(i) % n for i in index]
            if any(b - a != 1 for a, b in zip(idxs[:-1], idxs[1:])):
                raise ValueError("Bin indices must be monotonically increasing and adjacent: %s" % (idxs,))
        else:
            raise TypeError("Unhandled index type %s" % type(index))
        if not idxs:
            raise ValueError('`index` "%s" results in no bins being specified.' % (index,))
        names = None if self._bin_names is None else self._bin_names[idxs[0]: idxs[-1] + 1]
        return self._new(bin_edges=self._edges[idxs[0]: idxs[-1] + 2], bin_names=names)

    # ---- algebra
    def __mul__(self, other):
        if isinstance(other, OneDimBinning):
            return MultiDimBinning([self, other])
        if isinstance(other, MultiDimBinning):
            return MultiDimBinning([self] + list(other.dimensions))
        return OneDimBinning(name=self._name, tex=self._tex, bin_edges=self.bin_edges * other, is_log=self._is_log)

    def __add__(self, other):
        if isinstance(other, OneDimBinning):
            return MultiDimBinning([self, other])
        if isinstance(other, MultiDimBinning):
            return MultiDimBinning([self] + list(other.dimensions))
        return OneDimBinning(name=self._name, tex=self._tex, bin_edges=self.bin_edges + other, is_log=self._is_log)

    # ---- text, files
    def __repr__(self):
        """an expression `eval` turns back into an equal binning (with numpy's `array` in scope)"""
        parts = ["name=%r" % self._name, "tex=%r" % self._tex,
                 "bin_edges=array(%r)" % (self._edges.tolist(),), "units=%r" % str(self._units),
                 "is_log=%r" % self.is_log, "bin_names=%r" % (self._bin_names,)]
        return "OneDimBinning(%s)" % ", ".join(parts)

    def __str__(self):
        plural = "" if self.num_bins == 1 else "s"
        if self.is_irregular:
            kind = "irregularly-sized bin%s with edges at [%s]" % (plural, ", ".join("%g" % e for e in self._edges))
        elif self.num_bins == 1:
            kind = "bin spanning [%g, %g]" % (self._edges[0], self._edges[-1])
        else:
            kind = "%s bins spanning [%g, %g]" % ("logarithmically-uniform" if self._is_log else "equally-sized",
                                                  self._edges[0], self._edges[-1])
        units = "" if str(self._units) == "dimensionless" else " %s" % self._units
        names = "" if self._bin_names is None else " (bin names: %s)" % (self._bin_names,)
        return "'%s': %d %s%s%s" % (self._name, self.num_bins, kind, units, names)

    def to_json(self, filename, **kwargs):
        from pisa_amd.utils import jsons

        kwargs.pop("warn", None)
        jsons.to_json(self.serializable_state, filename, **kwargs)

    @classmethod
    def from_json(cls, resource):
        from pisa_amd.utils import jsons

        return cls(**_state_kwargs(jsons.from_json(resource)))


def _state_kwargs(state):
    """a `serializable_state` as constructor arguments (`is_lin` is implied by `is_log`)"""
    kw = dict(state)
    kw.pop("is_lin", None)
    return kw


class MultiDimBinning:
    """An ordered set of `OneDimBinning`s with distinct names; bins are flattened in C order."""

    def __init__(self, dimensions, name=None, mask=None):
        if isinstance(dimensions, OneDimBinning):
            dimensions = [dimensions]
        if isinstance(dimensions, MultiDimBinning):
            mask = dimensions.mask if mask is None else mask
            dimensions = dimensions.dimensions
        if isinstance(dimensions, Mapping):
            dimensions = [dimensions]
        dims = []
        for d in dimensions:
            if isinstance(d, Mapping):
                d = OneDimBinning(**_state_kwargs(d))
            if not isinstance(d, OneDimBinning):
                raise TypeError("Argument/object #%d unhandled type: %s" % (len(dims), type(d)))
            dims.append(d)
        names = [d.name for d in dims]
        if len(set(names)) != len(names):
            raise ValueError("dimension names must be unique: %s" % names)
        self._dimensions = tuple(dims)
        self.name = name
        self._shape = tuple(d.num_bins for d in dims)
        self._names = tuple(d.name for d in dims)
        self._name_set = frozenset(self._names)
        self._size = int(np.prod(self._shape)) if dims else 1
        if mask is not None:
            mask = np.asarray(mask, dtype=bool)
            if mask.shape != self._shape:
                raise ValueError("mask of shape %s does not fit the binning's %s" % (mask.shape, self._shape))
        self._mask = mask
        self._hash = None
        self._coord = None

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_coord"] = None              # a namedtuple type made on demand: not picklable, rebuilt when asked for
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)

    dimensions = property(lambda self: self._dimensions)
    dims = dimensions
    mask = property(lambda self: self._mask)
    names = property(lambda self: list(self._names))
    basenames = property(lambda self: [d.basename for d in self._dimensions])
    num_dims = property(lambda self: len(self._dimensions))
    shape = property(lambda self: self._shape)
    num_bins = property(lambda self: list(self._shape))
    size = property(lambda self: self._size)
    tot_num_bins = size
    bin_edges = property(lambda self: [d.bin_edges for d in self._dimensions])
    domains = property(lambda self: [d.domain for d in self._dimensions])
    units = property(lambda self: [d.units for d in self._dimensions])

    def iterdims(self):
        return iter(self._dimensions)

    @property
    def is_irregular(self):
        return bool(np.any([d.is_irregular for d in self]))

    @property
    def is_lin(self):
        return bool(np.all([d.is_lin for d in self]))

    @property
    def is_log(self):
        return bool(np.all([d.is_log for d in self]))

    @property
    def normalize_values(self):
        flags = {d.normalize_values for d in self._dimensions}
        assert len(flags) == 1
        return flags.pop()

    @normalize_values.setter
    def normalize_values(self, b):
        for d in self._dimensions:
            d.normalize_values = b
        self._hash = None

    @property
    def basename_binning(self):
        return MultiDimBinning([d.basename_binning for d in self._dimensions])

    @property
    def finite_binning(self):
        return MultiDimBinning([d.finite_binning for d in self._dimensions])

    @property
    def inbounds_criteria(self):
        return "(%s)" % " & ".join(d.inbounds_criteria for d in self._dimensions)

    def __iter__(self):
        return iter(self._dimensions)

    def __len__(self):
        return self.num_dims

    def __contains__(self, x):
        try:
            self.index(x)
        except (ValueError, TypeError):
            return False
        return True

    def index(self, dim, use_basenames=False):
        """position of a dimension given as a name, a `OneDimBinning` or a number (binning.py:2003-2056)"""
        if isinstance(dim, OneDimBinning):
            dim = dim.name
        if isinstance(dim, str):
            names = self.basenames if use_basenames else self.names
            d = dim.replace("true_", "").replace("reco_", "") if use_basenames else dim
            if d not in names:
                raise ValueError("Dimension %s'%s' not present; valid %snames are %s"
                                 % ("base" if use_basenames else "", d, "base" if use_basenames else "", names))
            return names.index(d)
        if isinstance(dim, (int, np.integer)) and not isinstance(dim, bool):
            if dim < 0 or dim >= len(self):
                raise ValueError("Dimension %d does not exist. Valid dimensions indices are in the range %s."
                                 % (dim, [0, len(self) - 1]))
            return int(dim)
        raise TypeError('Unhandled type for `dim`: "%s"' % type(dim))

    def remove(self, dims):
        """the binning without the dimensions `dims` (names / numbers)"""
        if isinstance(dims, (str, int, OneDimBinning)):
            dims = [dims]
        drop = {self.index(d) for d in dims}
        return MultiDimBinning([d for i, d in enumerate(self._dimensions) if i not in drop])

    def squeeze(self):
        """without the dimensions that have one bin (binning.py:2615-2627)"""
        return MultiDimBinning([d for d in self._dimensions if len(d) > 1])

    # ---- bins
    @property
    def coord(self):
        """namedtuple type of a bin's coordinates, one field per dimension"""
        if self._coord is None:
            self._coord = namedtuple("coord", self._names)
        return self._coord

    def index2coord(self, index):
        """flat (C order) bin number -> coordinate tuple (binning.py:2298-2323)"""
        return self.coord(*(int(i) for i in np.unravel_index(index, self._shape)))

    def itercoords(self):
        return (self.coord(*c) for c in itertools.product(*(range(n) for n in self._shape)))

    def iterbins(self):
        """one single-bin `MultiDimBinning` per bin, in C order"""
        return (MultiDimBinning(bins) for bins in itertools.product(*(d.iterbins() for d in self._dimensions)))

    def iteredgetuples(self):
        return itertools.product(*(d.iteredgetuples() for d in self._dimensions))

    def indexer(self, **kwargs):
        """an index tuple for arrays of this binning's shape: the named dimensions indexed as given, all
        others in full (binning.py:2085-2155)"""
        idx = [slice(None)] * self.num_dims
        for name, sel in kwargs.items():
            idx[self.index(name)] = sel
        return tuple(idx)

    def slice(self, **kwargs):
        return self[self.indexer(**kwargs)]

    def broadcast(self, a, from_dim, to_dims):
        """the one-dimensional `a` (along `from_dim`) shaped so that it broadcasts over `to_dims`"""
        a = np.asarray(a)
        assert a.ndim == 1
        if isinstance(to_dims, (str, int, OneDimBinning)):
            to_dims = [to_dims]
        present = sorted({self.index(from_dim)} | {self.index(d) for d in to_dims})
        shape = [len(a) if i == self.index(from_dim) else 1 for i in present]
        return a.reshape(shape)

    def __getitem__(self, index):
        """a dimension by NAME; otherwise bins: one int or slice per dimension (binning.py:2946-3032)"""
        if index is Ellipsis:
            return self
        if isinstance(index, str):
            for d in self._dimensions:
                if d.name == index:
                    return d
            raise ValueError("index '%s' not in %s" % (index, self.names))
        if isinstance(index, Iterable) and not isinstance(index, Sequence):
            index = list(index)
        if not isinstance(index, Sequence):
            index = [index]
        sel = []
        for idx in index:
            if isinstance(idx, (int, np.integer)) and not isinstance(idx, bool):
                sel.append(slice(idx, idx + 1) if idx >= 0 else slice(idx, idx + 1 if idx != -1 else None))
            elif isinstance(idx, slice):
                sel.append(idx)
            else:
                raise ValueError("Binning idx is %s, int or slice is needed" % (idx,))
        if len(sel) != self.num_dims:
            raise ValueError("Binning is %dD, but %dD indexing was passed" % (self.num_dims, len(sel)))
        mask = None if self._mask is None else self._mask[tuple(sel)]
        return MultiDimBinning([d[s] for d, s in zip(self._dimensions, sel)], mask=mask)

    def __getattr__(self, attr):
        # dimension access by attribute, e.g. binning.reco_energy
        if attr.startswith("_"):
            raise AttributeError(attr)
        for d in self.__dict__.get("_dimensions", ()):
            if d.name == attr:
                return d
        raise AttributeError(attr)

    # ---- algebra
    def __add__(self, other):
        other = MultiDimBinning(other)
        return MultiDimBinning(list(self._dimensions) + list(other._dimensions))

    def __mul__(self, other):
        if isinstance(other, (OneDimBinning, MultiDimBinning)):
            return self + other
        return MultiDimBinning([d * other for d in self._dimensions])

    def reorder_dimensions(self, order, use_deepcopy=False, use_basenames=False):
        """the dimensions in the order given by names, numbers, `OneDimBinning`s or another binning; entries
        naming no dimension of this binning are ignored, but every dimension must be named
        (binning.py:2325-2388)"""
        if isinstance(order, MultiDimBinning):
            order = order.names
        picked = []
        for o in order:
            try:
                i = self.index(o, use_basenames=use_basenames)
            except ValueError:
                continue
            if i not in picked:
                picked.append(i)
        if len(picked) != self.num_dims:
            raise ValueError("Invalid `order`: Only a subset of the dimensions present were specified. `order`=%s,"
                             " but dimensions=%s" % (list(order), self.names))
        mask = None if self._mask is None else np.transpose(self._mask, picked)
        return MultiDimBinning([self._dimensions[i] for i in picked], name=self.name, mask=mask)

    def _per_dim(self, args, kwargs, default):
        """one value per dimension from positional values (one for all, or one each) or values by name"""
        if args and kwargs:
            raise ValueError("Either specify positional or keyword arguments, not both")
        if kwargs:
            for k in kwargs:
                self.index(k)
            return [kwargs.get(n, default) for n in self._names]
        if len(args) == 1:
            return list(args) * self.num_dims
        if len(args) != self.num_dims:
            raise ValueError("%d value(s) given for %d dimension(s)" % (len(args), self.num_dims))
        return list(args)

    def oversample(self, *args, **kwargs):
        """`oversample(3)`, `oversample(2, 5)`, `oversample(coszen=10, energy=2)` (binning.py:2415-2512)"""
        factors = self._per_dim(args, kwargs, 1)
        return MultiDimBinning([d.oversample(f) for d, f in zip(self, factors)], name=self.name)

    def downsample(self, *args, **kwargs):
        factors = self._per_dim(args, kwargs, 1)
        return MultiDimBinning([d.downsample(f) for d, f in zip(self, factors)], name=self.name)

    def to(self, *args, **kwargs):
        """`to('MeV', '')`, `to(energy='MeV')`: the dimensions converted to the units given (None or '' for a
        dimensionless one keeps it)"""
        keep = object()
        units = self._per_dim(args, kwargs, keep)
        return MultiDimBinning([d if u is keep else d.to(u) for d, u in zip(self, units)], name=self.name,
                               mask=self._mask)

    def ito(self, *args, **kwargs):
        new = self.to(*args, **kwargs)
        self._dimensions = new._dimensions
        self._hash = None

    def is_compat(self, other):
        """same dimension names in the same order, each compatible (binning.py:2390-2413)"""
        if not isinstance(other, MultiDimBinning) or self._names != other._names:
            return False
        return all(a.is_compat(b) for a, b in zip(self._dimensions, other._dimensions))

    def assert_compat(self, other):
        """`other` (a binning, or something that carries one) must be compatible with this binning"""
        if not isinstance(other, MultiDimBinning):
            other = getattr(other, "binning", other)
        if not self.is_compat(other):
            raise AssertionError("incompatible binning: %s vs. %s" % (self, other))

    def assert_array_fits(self, array):
        if array.shape != self._shape:
            raise ValueError("Array shape %s does not match binning shape %s" % (array.shape, self._shape))

    # ---- derived arrays
    def meshgrid(self, entity="weighted_centers", attach_units=False):
        """`np.meshgrid(..., indexing='ij')` of the per-dimension entity (binning.py:2669-2711)."""
        arrs = [_mag(getattr(d, entity)) for d in self]
        grid = np.meshgrid(*arrs, indexing="ij")
        if attach_units:
            return [Quantity(g, getattr(d, entity).units) for g, d in zip(grid, self)]
        return grid

    @property
    def weighted_centers(self):
        return [d.weighted_centers for d in self]

    @property
    def midpoints(self):
        return [d.midpoints for d in self]

    def _volumes(self, entity, attach_units):
        widths = [getattr(d, entity) for d in self._dimensions]
        vol = _mag(widths[0])
        for w in widths[1:]:
            vol = np.multiply.outer(vol, _mag(w))
        if attach_units:
            u = widths[0].units
            for w in widths[1:]:
                u = u * w.units
            return Quantity(vol, u)
        return vol

    def bin_volumes(self, attach_units=False):
        """outer product of the bin widths (binning.py:2713-2731)"""
        return self._volumes("bin_widths", attach_units)

    def weighted_bin_volumes(self, attach_units=False):
        return self._volumes("weighted_bin_widths", attach_units)

    # ---- maps
    def _map(self, name, hist, map_kw, kwargs):
        from pisa_amd.core.map import Map

        kw = dict(map_kw or {})
        kw.update(kwargs)
        return Map(name=name, hist=hist, binning=self, **kw)

    def empty(self, name, map_kw=None, **kwargs):
        return self._map(name, np.empty(self._shape, dtype=FTYPE), map_kw, kwargs)

    def zeros(self, name, map_kw=None, **kwargs):
        return self._map(name, np.zeros(self._shape, dtype=FTYPE), map_kw, kwargs)

    def ones(self, name, map_kw=None, **kwargs):
        return self._map(name, np.ones(self._shape, dtype=FTYPE), map_kw, kwargs)

    def full(self, fill_value, name, map_kw=None, **kwargs):
        return self._map(name, np.full(self._shape, fill_value, dtype=FTYPE), map_kw, kwargs)

    # ---- comparison, text, files
    @property
    def mask_hash(self):
        return None if self._mask is None else _digest(np.ascontiguousarray(self._mask).tobytes())

    @property
    def serializable_state(self):
        return OrderedDict([("dimensions", [d.serializable_state for d in self._dimensions]), ("name", self.name),
                            ("mask", self._mask)])

    @property
    def hashable_state(self):
        return OrderedDict([("dimensions", [d.hashable_state for d in self._dimensions]), ("name", self.name),
                            ("mask_hash", self.mask_hash)])

    @property
    def edges_hash(self):
        return _digest(*(d.edges_hash for d in self._dimensions))

    @property
    def hash(self):
        if self._hash is None:
            self._hash = _digest(*(d.hash for d in self._dimensions), self.mask_hash)
        return self._hash

    def __hash__(self):
        return self.hash

    def __eq__(self, other):
        return isinstance(other, MultiDimBinning) and self.hash == other.hash

    def __ne__(self, other):
        return not self == other

    def __repr__(self):
        args = ["dimensions=[\n    %s\n]" % ",\n    ".join(repr(d) for d in self)]
        if self.name is not None:
            args.append("name=%r" % self.name)
        if self._mask is not None:
            args.append("mask=array(%r)" % (self._mask.tolist(),))
        return "MultiDimBinning(%s)" % ", ".join(args)

    def __str__(self):
        head = "" if self.name is None else '"%s":\n' % self.name
        return head + "\n".join("    " + str(d) for d in self._dimensions)

    def to_json(self, filename, **kwargs):
        from pisa_amd.utils import jsons

        kwargs.pop("warn", None)
        jsons.to_json(self.serializable_state, filename, **kwargs)

    @classmethod
    def from_json(cls, resource):
        from pisa_amd.utils import jsons

        return cls(**jsons.from_json(resource))


class VarBinning:
    """One `MultiDimBinning` per event selection (binning.py:3043-3178): a pipeline whose output binning is a
    `VarBinning` returns one `MapSet` per selection.  `selections` is a list of cut expressions over
    container keys (mutually exclusive: checked by the pipeline), or a `OneDimBinning` whose bins are the
    selections; the binnings must not bin in the variable the selection cuts on."""

    def __init__(self, binnings, selections):
        if not isinstance(selections, (OneDimBinning, list)):
            raise ValueError("Selection type %s not supported!" % type(selections))
        assert isinstance(binnings, list) and len(binnings) == len(selections)
        assert len(binnings) > 1            # one selection: an ordinary cut and a MultiDimBinning
        for b in binnings:
            assert isinstance(b, MultiDimBinning)
            shared = self._selection_vars_in(b, selections)
            if shared and isinstance(selections, OneDimBinning):
                raise ValueError("Selection variable %s (the OneDimBinning dimension) may not simultaneously be"
                                 " part of any MultiDimBinning!" % shared[0])
        self._binnings = binnings
        self._selections = selections
        self._hash = None

    @staticmethod
    def _selection_vars_in(binning, selections):
        """the dimensions of `binning` the selections refer to"""
        if isinstance(selections, OneDimBinning):
            return [selections.name] if selections.name in binning.names else []
        import re

        words = {w for cut in selections for w in re.findall(r"\b\w+\b", cut)}
        return [d for d in binning.names if d in words]

    binnings = property(lambda self: self._binnings)
    selections = property(lambda self: self._selections)
    nselections = property(lambda self: len(self._selections))
    names = property(lambda self: [b.names for b in self._binnings])

    @property
    def selection_strings(self):
        """the selections as cut expressions (a binned selection: `(x >= lo) & (x < hi)` per bin, the last bin
        closed above)"""
        if not isinstance(self._selections, OneDimBinning):
            return list(self._selections)
        s, e = self._selections, self._selections.edge_magnitudes
        last = len(e) - 2
        return ["(%s >= %.15e) & (%s %s %.15e)" % (s.name, e[i], s.name, "<=" if i == last else "<", e[i + 1])
                for i in range(len(e) - 1)]

    def __len__(self):
        return len(self._binnings)

    def __iter__(self):
        return iter(self._binnings)

    def __getitem__(self, i):
        return self._binnings[i]

    @property
    def serializable_state(self):
        sel = self._selections
        return OrderedDict([("binnings", [b.serializable_state for b in self._binnings]),
                            ("selections", sel.serializable_state if isinstance(sel, OneDimBinning) else sel)])

    @property
    def hash(self):
        if self._hash is None:
            sel = self._selections
            self._hash = _digest(*(b.hash for b in self._binnings),
                                 sel.hash if isinstance(sel, OneDimBinning) else tuple(sel))
        return self._hash

    def __hash__(self):
        return self.hash

    def __eq__(self, other):
        return isinstance(other, VarBinning) and self.hash == other.hash

    def __ne__(self, other):
        return not self == other

    def __repr__(self):
        return "VarBinning(binnings=%r, selections=%r)" % (self._binnings, self._selections)
