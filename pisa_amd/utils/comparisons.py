"""Comparing and normalising numbers, quantities and containers of them: the interface of
pisa/utils/comparisons.py (`ALLCLOSE_KW` :81-93 -- the reference's own definition of "equal" --, `isscalar`,
`isbarenumeric`, `isunitless`, `recursiveEquality` :185-356, `recursiveAllclose`, `normQuant` :431-710,
`interpret_quantity` :713-761) on this package's `Quantity`.  Host-side helpers."""
from collections import OrderedDict
from collections.abc import Iterable, Mapping, Sequence

import numpy as np

from pisa_amd import FTYPE, HASH_SIGFIGS
from pisa_amd.core.units import Quantity, ureg

__all__ = ["FTYPE_PREC", "FTYPE_SIGFIGS", "EQUALITY_SIGFIGS", "EQUALITY_PREC", "ALLCLOSE_KW", "isvalidname", "isscalar",
           "isbarenumeric", "isunitless", "recursiveEquality", "recursiveAllclose", "normQuant", "interpret_quantity"]

FTYPE_PREC = np.finfo(FTYPE).eps
FTYPE_SIGFIGS = int(np.abs(np.ceil(np.log10(FTYPE_PREC))))
EQUALITY_SIGFIGS = min(HASH_SIGFIGS, FTYPE_SIGFIGS)
EQUALITY_PREC = 10 ** -EQUALITY_SIGFIGS
ALLCLOSE_KW = dict(rtol=EQUALITY_PREC, atol=FTYPE_PREC, equal_nan=True)


def isvalidname(x):
    return isinstance(x, str) and x.isidentifier()


def isscalar(x):
    """a single number, with or without units"""
    if isinstance(x, Quantity):
        x = x.magnitude
    if isinstance(x, np.ndarray):
        return x.ndim == 0
    return not isinstance(x, (str, Mapping, Sequence, Iterable))


def isbarenumeric(x):
    """number(s) without units: a Python / numpy number or a numeric array"""
    if isinstance(x, Quantity) or isinstance(x, (str, bool)):
        return False
    if isinstance(x, (int, float, complex, np.number)):
        return True
    return isinstance(x, np.ndarray) and np.issubdtype(x.dtype, np.number)


def isunitless(x):
    """no units attached (numbers, arrays, and the containers the reference accepts)"""
    return not isinstance(x, Quantity)


def _close(x, y, allclose_kw):
    try:
        return bool(np.allclose(x, y, **allclose_kw)) if allclose_kw else bool(np.all(x == y))
    except TypeError:
        return bool(np.all(x == y))


def recursiveEquality(x, y, allclose_kw=ALLCLOSE_KW):  # noqa: N802 (the reference's name)
    """equal through any nesting of mappings, sequences, arrays and quantities: numbers to `ALLCLOSE_KW` (None:
    exactly), quantities after conversion to common units, NaN equal to NaN"""
    if hasattr(x, "hash") and not isinstance(x, (Mapping, np.ndarray)) and type(x) is type(y) and hasattr(x, "__eq__") \
            and not isinstance(x, Quantity):
        return bool(x == y)
    if isinstance(x, Quantity) or isinstance(y, Quantity):
        if not (isinstance(x, Quantity) and isinstance(y, Quantity)):
            return False
        if x.units.dims != y.units.dims:
            return False
        return _close(np.asarray(x.m_as(y.units), dtype=float), np.asarray(y.magnitude, dtype=float), allclose_kw)
    if isinstance(x, Mapping) or isinstance(y, Mapping):
        if not (isinstance(x, Mapping) and isinstance(y, Mapping)) or set(x.keys()) != set(y.keys()):
            return False
        if isinstance(x, OrderedDict) and isinstance(y, OrderedDict) and list(x) != list(y):
            return False
        return all(recursiveEquality(x[k], y[k], allclose_kw) for k in x)
    if isinstance(x, str) or isinstance(y, str) or x is None or y is None:
        return x == y
    if isinstance(x, np.ndarray) or isinstance(y, np.ndarray):
        xa, ya = np.asarray(x), np.asarray(y)
        if xa.shape != ya.shape:
            return False
        if xa.dtype == object or ya.dtype == object:
            return all(recursiveEquality(a, b, allclose_kw) for a, b in zip(xa.ravel(), ya.ravel()))
        return _close(xa, ya, allclose_kw)
    if isinstance(x, (Sequence, set, frozenset)) or isinstance(y, (Sequence, set, frozenset)):
        if not (isinstance(x, Iterable) and isinstance(y, Iterable)):
            return False
        xs, ys = list(x), list(y)
        return len(xs) == len(ys) and all(recursiveEquality(a, b, allclose_kw) for a, b in zip(xs, ys))
    try:
        if isbarenumeric(x) and isbarenumeric(y):
            return _close(x, y, allclose_kw)
        return bool(x == y)
    except (TypeError, ValueError):
        return False


def recursiveAllclose(x, y, *args, **kwargs):  # noqa: N802
    kw = dict(ALLCLOSE_KW)
    if args:
        kw.update(zip(("rtol", "atol", "equal_nan"), args))
    kw.update(kwargs)
    return recursiveEquality(x, y, allclose_kw=kw)


def _round(values, sigfigs):
    v = np.asarray(values, dtype=np.float64)
    out = np.array([float("%.*e" % (sigfigs - 1, a)) if np.isfinite(a) else a for a in v.ravel()]).reshape(v.shape)
    return float(out) if out.ndim == 0 else out


def normQuant(obj, sigfigs=None, full_norm=True):  # noqa: N802
    """`obj` in a form in which things that SHOULD be equal ARE: quantities in base units, numbers rounded to
    `sigfigs` significant figures, through mappings (plain dicts by sorted key) and sequences; objects that carry a
    `normalized_state` give that"""
    if not full_norm or isinstance(obj, str) or obj is None:
        return obj
    if sigfigs is not None:
        if not (int(sigfigs) == float(sigfigs) and sigfigs > 0):
            raise ValueError("`sigfigs` must be an integer > 0.")
        sigfigs = int(sigfigs)
    if hasattr(obj, "normalized_state"):
        return obj.normalized_state
    if isinstance(obj, Mapping):
        keys = obj.keys() if isinstance(obj, OrderedDict) else sorted(obj.keys())
        return OrderedDict((k, normQuant(obj[k], sigfigs, full_norm)) for k in keys)
    if isinstance(obj, Quantity):
        base = obj.to_base_units()
        m = base.magnitude if sigfigs is None else _round(base.magnitude, sigfigs)
        return Quantity(m, base.units)
    if isinstance(obj, Iterable) and not isinstance(obj, np.ndarray):
        return [normQuant(x, sigfigs, full_norm) for x in obj]
    if sigfigs is not None and isbarenumeric(obj) and not isinstance(obj, (bool, int, np.integer)):
        return _round(obj, sigfigs)
    return obj


def interpret_quantity(value, expect_sequence):
    """a number, a sequence of numbers, a quantity, or a sequence of quantities of one dimension made ONE quantity
    (dimensionless when no units are given); its magnitude a scalar or, `expect_sequence`, an array"""
    if isinstance(value, Quantity):
        q = value
    elif isinstance(value, Iterable) and not isinstance(value, (str, np.ndarray)):
        items = list(value)
        if items and all(isinstance(v, Quantity) for v in items):
            u = items[0].units
            q = Quantity(np.array([v.m_as(u) for v in items], dtype=FTYPE), u)
        elif any(isinstance(v, Quantity) for v in items):
            raise ValueError("Either all or no elements may have units")
        else:
            q = Quantity(np.array(items, dtype=FTYPE), ureg.dimensionless)
    else:
        q = Quantity(value, ureg.dimensionless)
    is_seq = isinstance(q.magnitude, np.ndarray) and q.magnitude.ndim > 0
    if expect_sequence and not is_seq:
        raise ValueError("Sequence expected, got %s" % (value,))
    if not expect_sequence and is_seq:
        raise ValueError("Scalar expected, got a sequence: %s" % (value,))
    return q
