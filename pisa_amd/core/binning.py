"""Binning value types: `OneDimBinning`, `MultiDimBinning`.

Host-side counterparts of pisa/core/binning.py (OneDimBinning :142,
MultiDimBinning :1484) restricted to what defines the NUMBERS on the hot
path: edge generation (`np.logspace` / `np.linspace`, binning.py:416-428),
`weighted_centers` (geometric mean for log bins, :901-911), regularity tests
(:1047-1115), C-order flattening / meshgrid (:2669-2711), oversampling and
bin volumes (used by the KDE stage).  Plotting, JSON round-trips, rebinning
and `VarBinning` are out of scope.
"""
import hashlib
from collections.abc import Iterable, Sequence

import numpy as np

from pisa_amd import FTYPE
from pisa_amd.core.units import Quantity, Unit, ureg

__all__ = ["OneDimBinning", "MultiDimBinning"]

_ALLCLOSE = dict(rtol=1e-12, atol=np.finfo(FTYPE).eps, equal_nan=True)


def _mag(x):
    return x.magnitude if isinstance(x, Quantity) else x


class OneDimBinning:
    def __init__(self, name, tex=None, bin_edges=None, units=None, domain=None, num_bins=None,
                 is_lin=None, is_log=None, bin_names=None):
        if not isinstance(name, str):
            raise TypeError("`name` must be a string")
        if is_lin and is_log:
            raise ValueError("`is_lin` and `is_log` are mutually exclusive")
        self._name = name
        self._tex = tex
        if units is not None and not isinstance(units, Unit):
            units = units.units if isinstance(units, Quantity) else ureg.parse_units(units)
        if bin_edges is not None:
            if isinstance(bin_edges, Quantity):
                units = units or bin_edges.units
                bin_edges = bin_edges.to(units).magnitude
            edges = np.array(bin_edges, dtype=FTYPE)
        else:
            edges = None
        if domain is not None:
            if isinstance(domain, Quantity):
                units = units or domain.units
                domain = domain.to(units).magnitude
            elif isinstance(domain, Sequence) and len(domain) == 2 and isinstance(domain[0], Quantity):
                units = units or domain[0].units
                domain = [domain[0].to(units).magnitude, domain[1].to(units).magnitude]
            domain = (float(domain[0]), float(domain[1]))
        if units is None:
            units = ureg.dimensionless
        if is_log is None and is_lin is None:
            is_log = False
        elif is_log is None:
            is_log = not is_lin
        self._is_log = bool(is_log)
        if edges is None:
            if num_bins is None or domain is None:
                raise ValueError("If not specifying bin edges explicitly, `domain` and `num_bins`"
                                 " must be specified (and optionally set `is_log=True`).")
            if self._is_log:
                edges = np.logspace(np.log10(domain[0]), np.log10(domain[1]), num_bins + 1,
                                    dtype=FTYPE)
            else:
                edges = np.linspace(domain[0], domain[1], num_bins + 1, dtype=FTYPE)
        elif domain is not None:
            assert domain[0] == edges[0] and domain[1] == edges[-1]
        if len(edges) < 2 or np.any(np.diff(edges) <= 0):
            raise ValueError("bin edges must be strictly increasing and define >= 1 bin")
        if num_bins is not None:
            assert num_bins == len(edges) - 1, "%s, %s" % (num_bins, edges)
        self._edges = edges
        self._units = units
        self._bin_names = list(bin_names) if bin_names is not None else None
        self._is_irregular = None
        self._hash = None

    # ---- basic attributes
    name = property(lambda self: self._name)
    tex = property(lambda self: self._tex)
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

    def __len__(self):
        return self.num_bins

    @property
    def basename(self):
        return self._name.replace("true_", "").replace("reco_", "")

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

    def oversample(self, factor):
        """`factor` sub-bins per bin, uniform in the binning's own space (binning.py:1307-1365)."""
        factor = int(factor)
        assert factor >= 1
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
        return OneDimBinning(self._name, tex=self._tex, bin_edges=np.concatenate(parts),
                             units=self._units, is_log=self._is_log)

    @property
    def hash(self):
        if self._hash is None:
            h = hashlib.md5()
            h.update(self._name.encode())
            h.update(b"log" if self._is_log else b"lin")
            h.update(np.ascontiguousarray(self._edges * self._units.scale).tobytes())
            self._hash = int.from_bytes(h.digest()[:8], "little", signed=True)
        return self._hash

    def __hash__(self):
        return self.hash

    def __eq__(self, other):
        return isinstance(other, OneDimBinning) and self.hash == other.hash

    def __ne__(self, other):
        return not self == other

    def __getitem__(self, idx):
        """sub-binning by bin index / slice"""
        idxs = np.arange(self.num_bins)[idx]
        idxs = np.atleast_1d(idxs)
        assert np.all(np.diff(idxs) == 1) or len(idxs) == 1
        edges = self._edges[idxs[0]: idxs[-1] + 2]
        return OneDimBinning(self._name, tex=self._tex, bin_edges=edges, units=self._units,
                             is_log=self._is_log)

    def __repr__(self):
        return "OneDimBinning('%s', %d %s bins spanning [%g, %g] %s)" % (
            self._name, self.num_bins, "log" if self._is_log else "lin", self._edges[0],
            self._edges[-1], self._units)


class MultiDimBinning:
    def __init__(self, dimensions, name=None, mask=None):
        if isinstance(dimensions, OneDimBinning):
            dimensions = [dimensions]
        if isinstance(dimensions, MultiDimBinning):
            dimensions = dimensions.dimensions
        dims = []
        for d in dimensions:
            if isinstance(d, dict):
                d = OneDimBinning(**d)
            assert isinstance(d, OneDimBinning)
            dims.append(d)
        names = [d.name for d in dims]
        if len(set(names)) != len(names):
            raise ValueError("dimension names must be unique: %s" % names)
        self._dimensions = tuple(dims)
        self.name = name
        self.mask = mask
        self._hash = None
        self._shape = tuple(d.num_bins for d in dims)
        self._names = tuple(d.name for d in dims)
        self._name_set = frozenset(self._names)
        self._size = int(np.prod(self._shape)) if dims else 1

    dimensions = property(lambda self: self._dimensions)
    dims = dimensions
    names = property(lambda self: list(self._names))
    num_dims = property(lambda self: len(self._dimensions))
    shape = property(lambda self: self._shape)
    num_bins = shape
    size = property(lambda self: self._size)
    tot_num_bins = size
    bin_edges = property(lambda self: [d.bin_edges for d in self._dimensions])
    domains = property(lambda self: [d.domain for d in self._dimensions])
    units = property(lambda self: [d.units for d in self._dimensions])

    @property
    def is_irregular(self):
        return bool(np.any([d.is_irregular for d in self]))

    @property
    def is_lin(self):
        return bool(np.all([d.is_lin for d in self]))

    @property
    def is_log(self):
        return bool(np.all([d.is_log for d in self]))

    def __iter__(self):
        return iter(self._dimensions)

    def __len__(self):
        return self.num_dims

    def index(self, dim):
        if isinstance(dim, OneDimBinning):
            dim = dim.name
        if isinstance(dim, str):
            if dim not in self.names:
                raise ValueError("dimension '%s' not in binning %s" % (dim, self.names))
            return self.names.index(dim)
        return int(dim)

    def __getitem__(self, key):
        if isinstance(key, str):
            return self._dimensions[self.index(key)]
        if isinstance(key, (int, np.integer)):
            return self._dimensions[key]
        raise TypeError("index a MultiDimBinning by dimension name or number")

    def __getattr__(self, attr):
        # dimension access by attribute, e.g. binning.reco_energy
        if attr.startswith("_"):
            raise AttributeError(attr)
        for d in self.__dict__.get("_dimensions", ()):
            if d.name == attr:
                return d
        raise AttributeError(attr)

    def __add__(self, other):
        other = MultiDimBinning(other)
        return MultiDimBinning(list(self._dimensions) + list(other._dimensions))

    def reorder_dimensions(self, order):
        return MultiDimBinning([self[n] for n in order], name=self.name)

    def oversample(self, *factors):
        if len(factors) == 1:
            factors = list(factors) * self.num_dims
        return MultiDimBinning([d.oversample(f) for d, f in zip(self, factors)], name=self.name)

    def meshgrid(self, entity="weighted_centers", attach_units=False):
        """`np.meshgrid(..., indexing='ij')` of the per-dimension entity (binning.py:2669-2711)."""
        arrs = [_mag(getattr(d, entity)) for d in self]
        grid = np.meshgrid(*arrs, indexing="ij")
        if attach_units:
            return [Quantity(g, d.units) for g, d in zip(grid, self)]
        return grid

    @property
    def weighted_centers(self):
        return [d.weighted_centers for d in self]

    @property
    def midpoints(self):
        return [d.midpoints for d in self]

    def bin_volumes(self, attach_units=True):
        """outer product of the bin widths (binning.py:2713-2731)"""
        vol = _mag(self._dimensions[0].bin_widths)
        for d in self._dimensions[1:]:
            vol = np.multiply.outer(vol, _mag(d.bin_widths))
        return vol

    def weighted_bin_volumes(self, attach_units=True):
        vol = _mag(self._dimensions[0].weighted_bin_widths)
        for d in self._dimensions[1:]:
            vol = np.multiply.outer(vol, _mag(d.weighted_bin_widths))
        return vol

    @property
    def hash(self):
        if self._hash is None:
            h = hashlib.md5()
            for d in self._dimensions:
                h.update(d.hash.to_bytes(8, "little", signed=True))
            self._hash = int.from_bytes(h.digest()[:8], "little", signed=True)
        return self._hash

    def __hash__(self):
        return self.hash

    def __eq__(self, other):
        return isinstance(other, MultiDimBinning) and self.hash == other.hash

    def __ne__(self, other):
        return not self == other

    def __repr__(self):
        return "MultiDimBinning(\n    %s\n)" % ",\n    ".join(repr(d) for d in self)
