"""Parameters: `Prior`, `Param`, `ParamSet`, `ParamSelector`.

Host-side counterparts of pisa/core/param.py and pisa/core/prior.py with the
semantics the evaluation loop relies on:

* `ParamSet.values_hash` drives the per-stage compute memo (stage.py:536-557).
  The reference md5-hashes 12-significant-figure-normalised pint quantities
  (param.py:1560-1565, ~ms per stage); here it is a tuple of the current
  magnitudes in the parameter's own units rounded the same way -- cheap and
  with the same equality classes.
* `_rescaled_value` maps values to [0,1] over `range` for minimisers
  (param.py:358-400); `randomize_free` draws uniform [0,1] per free param
  (param.py:1433-1449).
* `priors_penalty(metric)` = sum of prior llh (or chi2 = -2 llh) values
  (param.py:1372-1396, prior.py:204-253).
"""
from collections import OrderedDict
from collections.abc import Iterable, Mapping, Sequence

from copy import deepcopy

import numpy as np

from pisa_amd import HASH_SIGFIGS  # noqa: F401 (re-exported)
from pisa_amd.core.units import Quantity, ureg

__all__ = ["Prior", "Param", "DerivedParam", "LinearFunction", "ParamSet", "ParamSelector"]


FTYPE_PREC = np.finfo(np.float64).eps

LLH_METRICS = ("llh", "poisson_llh", "conv_llh", "barlow_llh", "mcllh_mean", "mcllh_eff",
               "generalized_poisson_llh")
CHI2_METRICS = ("chi2", "mod_chi2", "correct_chi2", "weighted_chi2", "signed_sqrt_mod_chi2")


def _as_quantity(v):
    if isinstance(v, Quantity) or v is None or isinstance(v, (bool, str)):
        return v
    return Quantity(v)


def _same(a, b):
    """recursive equality of Param state entries (utils/comparisons.py recursiveEquality: quantities by
    value and dimension, arrays element by element, sequences and mappings entry by entry)"""
    if a is b:
        return True
    if a is None or b is None:
        return False
    if isinstance(a, Quantity) or isinstance(b, Quantity):
        if not (isinstance(a, Quantity) and isinstance(b, Quantity)) or a.units.dims != b.units.dims:
            return False
        return bool(np.all(np.asarray(a.m_as(b.units)) == np.asarray(b.magnitude)))
    if isinstance(a, Prior) or isinstance(b, Prior):
        return isinstance(a, Prior) and isinstance(b, Prior) and _same(a.state, b.state)
    if isinstance(a, Mapping) and isinstance(b, Mapping):
        return a.keys() == b.keys() and all(_same(a[k], b[k]) for k in a)
    if isinstance(a, (list, tuple)) and isinstance(b, (list, tuple)):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        return np.shape(a) == np.shape(b) and bool(np.all(np.asarray(a) == np.asarray(b)))
    try:
        return bool(a == b)
    except (TypeError, ValueError):
        return False


class Prior:
    """uniform / gaussian / jeffreys priors (prior.py:189-262). `llh(x)` returns the
    log-prior (up to a constant); chi2 = -2 llh (prior.py:395-400)."""

    def __init__(self, kind, **kwargs):
        self.kind = None if kind is None else str(kind).lower()
        if self.kind in (None, "none", "uniform"):
            self.kind = "uniform"
            self.llh_offset = kwargs.get("llh_offset", 0.0)
            self.units = None
        elif self.kind == "gaussian":
            self.mean = _as_quantity(kwargs["mean"])
            self.stddev = _as_quantity(kwargs["stddev"]).to(self.mean.units)
            self.units = self.mean.units
        elif self.kind == "jeffreys":
            self.A = _as_quantity(kwargs["A"])
            self.B = _as_quantity(kwargs["B"]).to(self.A.units)
            self.units = self.A.units
        elif self.kind == "spline":
            # prior.py:285-318: scipy.interpolate.splev(x, (knots, coeffs, deg), ext=2)
            self.knots = _as_quantity(kwargs["knots"])
            if kwargs.get("units") is not None:
                u = ureg.parse_units(kwargs["units"])
                self.knots = (Quantity(self.knots.magnitude, u) if self.knots.units == ureg.dimensionless
                              else self.knots.to(u))
            self.coeffs = np.asarray(kwargs["coeffs"], dtype=np.float64)
            self.deg = int(kwargs["deg"])
            self.units = self.knots.units
        elif self.kind == "linterp":
            # prior.py:262-283: linear interpolation, error outside the tabulated range
            self.param_vals = _as_quantity(kwargs["param_vals"])
            self.llh_vals = np.asarray(kwargs["llh_vals"], dtype=np.float64)
            self.units = self.param_vals.units
        else:
            raise ValueError("prior kind '%s' unknown (uniform, gaussian, jeffreys, spline, linterp)" % kind)

    def __deepcopy__(self, memo):
        # a prior's attributes are set once and replaced, never changed in place: an object of its own
        # sharing them is as independent as the generic copy and costs a dict copy
        # (`HypoFitResult` snapshots the whole parameter set after every fit, analysis.py:356-372)
        new = object.__new__(Prior)
        new.__dict__.update(self.__dict__)
        memo[id(self)] = new
        return new

    _kind_attrs = {"uniform": ("llh_offset",), "gaussian": ("mean", "stddev"), "jeffreys": ("A", "B"),
                   "spline": ("knots", "coeffs", "deg"), "linterp": ("param_vals", "llh_vals")}

    @property
    def state(self):
        """`kind` and that kind's constructor arguments (prior.py:189-194): `Prior(**state)` is an equal prior"""
        return OrderedDict([("kind", self.kind)] + [(a, getattr(self, a)) for a in self._kind_attrs[self.kind]])

    serializable_state = state

    def __eq__(self, other):
        return isinstance(other, Prior) and _same(self.state, other.state)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = object.__hash__

    def _strip(self, x):
        x = _as_quantity(x)
        return x.m_as(self.units) if self.units is not None else x.magnitude

    @property
    def valid_range(self):
        """(low, high) quantities between which the prior is defined (prior.py:209-210, 232-233, 257-258, 280-281, 313-314)"""
        u = self.units if self.units is not None else ureg.dimensionless
        if self.kind in ("uniform", "gaussian"):
            lo, hi = -np.inf, np.inf
        elif self.kind == "jeffreys":
            lo, hi = self.A.magnitude, self.B.magnitude
        elif self.kind == "spline":
            lo, hi = np.min(self.knots.magnitude), np.max(self.knots.magnitude)
        else:
            lo, hi = np.min(self.param_vals.magnitude), np.max(self.param_vals.magnitude)
        return Quantity(lo, u), Quantity(hi, u)

    @property
    def max_at(self):
        """where the prior is largest: NaN for a uniform prior, the mean / A, the tabulated maxima, or scipy's
        `fminbound` of chi2 between the outer knots of a spline (prior.py:207, 230, 255, 277, 306-310)"""
        if self.kind == "uniform":
            return np.nan
        if self.kind == "gaussian":
            return self.mean
        if self.kind == "jeffreys":
            return self.A
        if self.kind == "linterp":
            return Quantity(np.asarray(self.param_vals.magnitude)[self.llh_vals == np.max(self.llh_vals)], self.units)
        from scipy.optimize import fminbound

        k = self.knots.magnitude
        return Quantity(fminbound(func=lambda v: self.chi2(Quantity(v, self.units)), x1=np.min(k), x2=np.max(k)), self.units)

    def llh(self, x):
        if self.kind == "uniform":
            return 0.0 * _as_quantity(x).magnitude + self.llh_offset
        v = self._strip(x)
        if self.kind == "gaussian":
            m, s = self.mean.magnitude, self.stddev.magnitude
            return -(v - m) ** 2 / (2 * s ** 2)
        if self.kind == "spline":
            from scipy.interpolate import splev

            return splev(v, tck=(self.knots.magnitude, self.coeffs, self.deg), ext=2)
        if self.kind == "linterp":
            xs, ys = np.asarray(self.param_vals.magnitude, dtype=np.float64), self.llh_vals
            if np.any(np.asarray(v) < xs.min()) or np.any(np.asarray(v) > xs.max()):
                raise ValueError("A value in x_new is outside the interpolation range.")
            order = np.argsort(xs)
            return np.interp(v, xs[order], ys[order])
        a, b = self.A.magnitude, self.B.magnitude
        return -np.log(v) + np.log(np.log(b) - np.log(a))

    def chi2(self, x):
        return -2 * self.llh(x)

    def __repr__(self):
        if self.kind == "gaussian":
            return "Prior(gaussian, mean=%s, stddev=%s)" % (self.mean, self.stddev)
        if self.kind == "spline":
            return "Prior(spline, deg=%d, %d knots)" % (self.deg, len(self.knots.magnitude))
        return "Prior(%s)" % self.kind


class Param:
    """One named quantity with prior, range and fixed flag (param.py:60-330).

    `Param.clock` counts value changes of ALL params of the process and `_ver` those of one
    param: evaluation plans that skip the per-stage hashing (core/fastplan.py) compare these
    instead of re-deriving `values_hash` (a value set to what it already was counts as a change)."""

    clock = 0
    fix_clock = 0      # counts fixed <-> free changes of ALL params (cached free-parameter views)

    is_fixed = property(lambda self: self._is_fixed)

    @is_fixed.setter
    def is_fixed(self, flag):
        self._is_fixed = bool(flag)
        Param.fix_clock += 1

    def __deepcopy__(self, memo):
        """an independent Param: scalar quantities are immutable and shared, array magnitudes are copied,
        the prior and the range list are objects of their own (2 ms -> 0.1 ms for a 30-parameter set)"""
        new = object.__new__(Param)
        d = dict(self.__dict__)
        for k in ("_value", "_nominal_value"):
            v = d.get(k)
            if isinstance(v, Quantity) and isinstance(v.magnitude, np.ndarray):
                d[k] = Quantity(v.magnitude.copy(), v.units)
            elif not isinstance(v, Quantity) and v is not None and not isinstance(v, (str, bool, int, float)):
                d[k] = deepcopy(v, memo)
        if d.get("_range") is not None:
            d["_range"] = list(d["_range"])
        if d.get("prior") is not None:
            d["prior"] = deepcopy(d["prior"], memo)
        new.__dict__.update(d)
        memo[id(self)] = new
        return new

    def m_in(self, units):
        """magnitude of the value in `units`, converted once per value (a fit reads the same few
        parameters at every point; the conversion through the units registry costs more than the
        arithmetic it feeds)"""
        c = self.__dict__.get("_m_cache")
        if c is not None and c[0] == self._ver and c[1] == units:
            return c[2]
        m = self._value.m_as(units)
        self._m_cache = (self._ver, units, m)
        return m

    def __init__(self, name, value, prior=None, range=None, is_fixed=True, unique_id=None,
                 is_discrete=False, nominal_value=None, tex=None, help="", scales_as_log=False):
        self.name = name
        self.unique_id = unique_id if unique_id is not None else name
        self._tex = tex
        self.help = help
        self.is_fixed = bool(is_fixed)
        self.is_discrete = bool(is_discrete)
        self.scales_as_log = bool(scales_as_log)
        self._ver = 0
        self._value = _as_quantity(value)
        self._units = self._value.units if isinstance(self._value, Quantity) else None
        self.prior = prior if (prior is None or isinstance(prior, Prior)) else Prior(**prior)
        self._range = None
        self.range = range
        self._nominal_value = self._value if nominal_value is None else _as_quantity(nominal_value)

    # -- value -----------------------------------------------------------
    @property
    def value(self):
        return self._value

    @value.setter
    def value(self, val):
        val = _as_quantity(val)
        if isinstance(self._value, Quantity):
            if not isinstance(val, Quantity):
                raise TypeError("value must be a quantity")
            if val.units.dims != self._units.dims:
                raise ValueError('Value "%s" units incompatible with units "%s".' % (val, self._units))
            val = val.to(self._units)
        self.validate_value(val)
        self._value = val
        self._touch()

    def _touch(self):
        self._ver += 1
        Param.clock += 1

    def validate_value(self, val):
        if self._range is not None and isinstance(val, Quantity):
            # the range's magnitudes in this parameter's units, converted once per range object
            c = self.__dict__.get("_range_m")
            if c is None or c[0] is not self._range or c[1] is not self._range[0] or c[2] is not self._range[1]:
                lo, hi = self._range[0].m_as(self._units), self._range[1].m_as(self._units)
                c = self._range_m = (self._range, self._range[0], self._range[1], min(lo, hi), max(lo, hi))
            v = val.magnitude if val.units is self._units else val.m_as(self._units)
            if v < c[3] or v > c[4]:
                raise ValueError("Param %s has a value %s which is not in the range of %s"
                                 % (self.name, val, self._range))

    m = property(lambda self: self._value.magnitude)
    magnitude = m
    units = property(lambda self: self._units)
    u = units
    dimensionality = property(lambda self: self._value.dimensionality)

    def m_as(self, u):
        return self._value.m_as(u)

    @property
    def range(self):
        return self._range

    @range.setter
    def range(self, rng):
        if rng is None:
            self._range = None
            return
        if isinstance(rng, Quantity):
            rng = [rng[0], rng[1]]
        rng = [_as_quantity(v) for v in rng]
        assert len(rng) == 2
        self._range = [v.to(self._units) if isinstance(self._value, Quantity) else v for v in rng]

    @property
    def nominal_value(self):
        return self._nominal_value

    @nominal_value.setter
    def nominal_value(self, v):
        self._nominal_value = _as_quantity(v)

    @property
    def tex(self):
        return r"{\rm %s}" % self.name.replace("_", r"\;") if self._tex is None else self._tex

    def reset(self):
        self._value = self._nominal_value
        self._touch()

    def set_nominal_to_current_value(self):
        self._nominal_value = self._value

    def randomize(self, random_state=None):
        rs = random_state if isinstance(random_state, np.random.RandomState) \
            else np.random.RandomState(random_state)
        self._rescaled_value = rs.rand()

    # -- [0,1] rescaling for minimisers (param.py:358-400) -----------------
    @property
    def _rescaled_value(self):
        if self.is_discrete:
            return self.value
        if self._range is None:
            raise ValueError("Cannot rescale without a range specified for parameter %s" % self.name)
        r0, r1 = self._range[0].m_as(self._units), self._range[1].m_as(self._units)
        v = self._value.m_as(self._units)
        if self.scales_as_log:
            if r0 < 0:
                r0, r1, v = -r0, -r1, -v
            return (np.log(v) - np.log(r0)) / (np.log(r1) - np.log(r0))
        return (v - r0) / (r1 - r0)

    @_rescaled_value.setter
    def _rescaled_value(self, rval):
        if self._range is None:
            raise ValueError("Cannot rescale without a range specified for parameter %s" % self.name)
        if rval < 0 or rval > 1 + FTYPE_PREC:
            raise ValueError("%s: `rval`=%.15e, but cannot be outside [0, 1]" % (self.name, rval))
        rval = 1.0 if rval > 1.0 else rval            # (np.min([1.0, rval]) cost 3 us of this 6 us setter: a fit calls it per free parameter and point)
        r0, r1 = self._range[0].m_as(self._units), self._range[1].m_as(self._units)
        if self.scales_as_log:
            v = np.exp(rval * (np.log(np.abs(r1)) - np.log(np.abs(r0)))) * r0
        else:
            v = r0 + (r1 - r0) * rval
        v = min(max(v, min(r0, r1)), max(r0, r1))
        if isinstance(self._value, Quantity) and self._value.units == self._units and self._value.magnitude == v:
            # the very value it has: nothing changed, no stage needs to recompute (a minimiser sets ALL
            # free parameters at every point, although a finite-difference step moves one of them)
            return
        self._value = Quantity(v, self._units)
        self._touch()

    def prior_penalty_cached(self, metric):
        """`prior_penalty`, evaluated once per (value, prior, metric)"""
        c = self.__dict__.get("_pen_cache")
        if c is not None and c[0] == self._ver and c[1] is self.prior and c[2] == metric:
            return c[3]
        v = self.prior_penalty(metric)
        self._pen_cache = (self._ver, self.prior, metric, v)
        return v

    def prior_penalty(self, metric):
        metric = metric.strip().lower()
        if self.prior is None:
            return 0
        if metric in LLH_METRICS:
            return self.prior.llh(self._value)
        if metric in CHI2_METRICS:
            return self.prior.chi2(self._value)
        raise ValueError('Metric "%s" is invalid' % metric)

    def _hashable(self):
        """the value as it enters `values_hash`; computed once per change of the value (`_ver`)"""
        c = self.__dict__.get("_hashable_cache")
        if c is not None and c[0] == self._ver:
            return c[1]
        h = self._hashable_now()
        self._hashable_cache = (self._ver, h)
        return h

    def _hashable_now(self):
        v = self._value
        if isinstance(v, Quantity):
            m = v.magnitude
            if np.isscalar(m) and isinstance(m, (float, np.floating)) and m != 0 and np.isfinite(m):
                # normQuant: round to HASH_SIGFIGS significant figures (utils/comparisons.py)
                m = float("%.*e" % (HASH_SIGFIGS - 1, m))
            elif not np.isscalar(m):
                m = tuple(np.ravel(m).tolist())
            return (m, v.units.scale, v.units.dims)
        return v

    _state_attrs = ("name", "unique_id", "value", "prior", "range", "is_fixed", "is_discrete",
                    "nominal_value", "tex", "help", "scales_as_log")

    @property
    def state(self):
        """the attributes that define the Param, in a fixed order (param.py:402-414)"""
        return OrderedDict((a, getattr(self, a)) for a in self._state_attrs)

    @property
    def serializable_state(self):
        s = self.state
        s["tex"] = self._tex            # the constructor's argument, not the derived default
        return s

    def to_json(self, filename, **kwargs):
        """the state as a JSON file `Param.from_json` reads back (param.py:567-578)"""
        from pisa_amd.utils import jsons

        jsons.to_json(self.serializable_state, filename, **kwargs)

    @classmethod
    def from_json(cls, filename):
        from pisa_amd.utils import jsons

        return cls(**jsons.from_json(filename))

    def __eq__(self, other):
        """same state (param.py:223-226): name, value, nominal value, range, prior, flags"""
        if not isinstance(other, Param):
            return False
        if other is self:
            return True
        return all(_same(getattr(self, a), getattr(other, a)) for a in self._state_attrs)

    def __ne__(self, other):
        return not self.__eq__(other)

    def __lt__(self, other):
        return self.name < other.name

    __hash__ = object.__hash__      # identity: plans and caches key on the Param OBJECT

    def __repr__(self):
        return "Param(%s=%s, fixed=%s, range=%s, prior=%s)" % (self.name, self._value,
                                                               self.is_fixed, self._range, self.prior)


class LinearFunction:
    """offset + sum_i coeff_i * value of the parameter named name_i: the functions `add_covariance` builds (the
    reference composes them from `utils.callable.Var` / `Funct` objects, param.py:1070-1081)"""

    def __init__(self, names, coeffs, offset=0.0):
        self.names, self.coeffs, self.offset = tuple(names), tuple(float(c) for c in coeffs), float(offset)

    def __call__(self, **params):
        value = 0.0
        for n, c in zip(self.names, self.coeffs):
            value += c * float(params[n].value.m)
        return value + self.offset

    @property
    def state(self):
        return OrderedDict([("kind", "linear"), ("names", list(self.names)), ("coeffs", list(self.coeffs)),
                            ("offset", self.offset)])


class DerivedParam(Param):
    """A parameter whose value is a function of other parameters (param.py:579-766): always fixed, no prior penalty
    of its own (its arguments carry the priors), never validated.  `callable(**{name: Param})` gives the
    (dimensionless) value.  Its change counter `_ver` is the sum of its arguments': every cache keyed on a
    parameter's version follows the arguments."""

    def __init__(self, name, value, unique_id=None, is_discrete=False, scales_as_log=False, nominal_value=None,
                 tex=None, range=None, depends_names="", function_file="", help=""):  # noqa: A002
        d = self.__dict__
        d["_dependson"], d["_configured"], d["_callable"] = OrderedDict(), False, None
        d["_depends_names"] = tuple(depends_names) if not isinstance(depends_names, str) else tuple(depends_names.split())
        self.name = name
        self.unique_id = unique_id if unique_id is not None else name
        self._tex, self.help = tex, help
        d["_is_fixed"] = True
        self.is_discrete, self.scales_as_log = bool(is_discrete), bool(scales_as_log)
        self._units = ureg.dimensionless
        self.prior = None
        self._range = None if range is None else list(range)
        start = _as_quantity(value)
        self._nominal_value = start if nominal_value is None else _as_quantity(nominal_value)
        d["_start_value"] = start
        if function_file:
            raise NotImplementedError("DerivedParam functions from a file (`function_file`) are not part of this build;"
                                      " set `callable` to a Python callable")

    # the arguments and the function
    @property
    def callable(self):
        if self._callable is None:
            raise ValueError("No set callable for DerivedParam %s" % self.name)
        return self._callable

    @callable.setter
    def callable(self, what):
        self.__dict__["_callable"] = what
        Param.clock += 1

    @property
    def depends_names(self):
        return self._depends_names

    @property
    def dependson(self):
        if not self._configured:
            raise ValueError("Cannot access unconfigured Derived parameter!")
        return self._dependson

    @dependson.setter
    def dependson(self, params):
        working = [Param(**p) if isinstance(p, Mapping) else p for p in params]
        if not all(isinstance(p, Param) for p in working):
            raise TypeError("a DerivedParam depends on Params")
        self.__dict__["_dependson"] = OrderedDict((p.name, p) for p in working)
        self.__dict__["_depends_names"] = tuple(p.name for p in working)
        self.__dict__["_configured"] = True
        Param.clock += 1

    # value, version
    @property
    def _value(self):
        if not self._configured or self._callable is None:
            return self._start_value
        return Quantity(self._callable(**self._dependson), ureg.dimensionless)

    @_value.setter
    def _value(self, v):
        raise AttributeError("the value of DerivedParam '%s' follows from %s" % (self.name, list(self._depends_names)))

    @property
    def _ver(self):
        return sum(p._ver for p in self._dependson.values())

    @_ver.setter
    def _ver(self, v):
        pass

    value = property(lambda self: self._value)

    @value.setter
    def value(self, val):
        raise AttributeError("the value of DerivedParam '%s' follows from %s" % (self.name, list(self._depends_names)))

    is_fixed = property(lambda self: True)

    @is_fixed.setter
    def is_fixed(self, flag):
        if not flag:
            raise ValueError("DerivedParam '%s' cannot be freed: free the parameters it depends on" % self.name)

    def validate_value(self, val):
        return

    def reset(self):
        pass

    def prior_penalty(self, metric):
        return 0.0          # the arguments carry the priors: no double counting (param.py:729-733)

    def __deepcopy__(self, memo):
        new = object.__new__(DerivedParam)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = deepcopy(v, memo)
        return new

    @property
    def state(self):
        return OrderedDict([("callable", getattr(self._callable, "state", repr(self._callable))),
                            ("depends", list(self._depends_names)), ("range", self._range)])

    serializable_state = state

    def __eq__(self, other):
        return isinstance(other, DerivedParam) and self.name == other.name and _same(self.state, other.state) \
            and _same(self.value, other.value)

    __hash__ = object.__hash__


class ParamSet(Sequence):
    """Ordered set of `Param`s with name access (param.py:776-1600).

    `ParamSet.struct_clock` counts structural changes (a Param object added, replaced) of ALL sets
    of the process: merged views (`Pipeline.params`) and evaluation plans are rebuilt when it moved."""

    struct_clock = 0

    def __init__(self, *args):
        params = []
        for a in args:
            if a is None:
                continue
            if isinstance(a, Param):
                params.append(a)
            elif isinstance(a, (ParamSet, Sequence, Iterable)) and not isinstance(a, Mapping):
                params.extend(list(a))
            elif isinstance(a, Mapping):
                params.append(Param(**a))
        names = [p.name for p in params]
        if len(set(names)) != len(names):
            raise ValueError("duplicate parameter names: %s" % sorted(n for n in names if names.count(n) > 1))
        object.__setattr__(self, "_params", params)
        object.__setattr__(self, "normalize_values", True)
        # name -> position; every mutator below keeps it current (a Param's name never changes)
        object.__setattr__(self, "_index", {p.name: i for i, p in enumerate(params)})
        # a transient set (a merged VIEW built for one call, `DistributionMaker.params`) is owned by
        # nobody: filling it is not a structural change of the process' parameter sets
        object.__setattr__(self, "_transient", False)

    # sequence protocol
    def __len__(self):
        return len(self._params)

    def __iter__(self):
        return iter(self._params)

    def __getitem__(self, i):
        if isinstance(i, str):
            return self._params[self.index(i)]
        return self._params[i]

    def __contains__(self, item):
        """a name: a parameter of that name is present; a Param: an EQUAL one is (the reference's set has no
        `__contains__`, membership falls to iteration and `Param.__eq__`, param.py:223)"""
        if isinstance(item, Param):
            i = self._pos(item.name)
            return i is not None and self._params[i] == item
        return self._pos(item) is not None

    def issubset(self, other):
        return all(p in other for p in self._params)

    def issuperset(self, other):
        return all(p in self for p in other)

    def isdisjoint(self, other):
        return not any(p in other for p in self._params)

    def __le__(self, other):
        return self.issubset(other)

    def __lt__(self, other):
        return len(other) > len(self) and self.issubset(other)

    def __ge__(self, other):
        return self.issuperset(other)

    def __gt__(self, other):
        return len(self) > len(other) and self.issuperset(other)

    def __eq__(self, other):
        if not isinstance(other, ParamSet):
            return False
        return len(self) == len(other) and all(a == b for a, b in zip(self._params, other._params))

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = object.__hash__

    def __delitem__(self, i):
        """`del ps[3]`, `del ps['name']` (param.py:1268-1270)"""
        del self._params[self.index(i)]
        self._reindex()

    def __setitem__(self, i, val):
        assert isinstance(val, Param)
        self._params[self.index(i)] = val
        self._reindex()

    def insert(self, index, value):
        assert isinstance(value, Param)
        if value.name in self._index:
            raise ValueError("parameter '%s' already present" % value.name)
        self._params.insert(index, value)
        self._reindex()

    def remove(self, value):
        del self[self.index(value)]

    def pop(self, i=-1):
        p = self._params.pop(self.index(i) if not isinstance(i, (int, np.integer)) else i)
        self._reindex()
        return p

    def _reindex(self):
        self._index.clear()
        self._index.update((p.name, i) for i, p in enumerate(self._params))
        self._bump()

    def _pos(self, name):
        return self._index.get(name)

    def _bump(self):
        if not self._transient:
            ParamSet.struct_clock += 1

    def __getattr__(self, attr):
        if attr.startswith("_"):
            raise AttributeError(attr)
        i = self._pos(attr)
        if i is None:
            raise AttributeError("no parameter named '%s'" % attr)
        return self._params[i]

    def __setattr__(self, attr, val):
        if attr in self.__dict__ or attr in type(self).__dict__:
            object.__setattr__(self, attr, val)
            return
        i = self._pos(attr)
        if i is None:
            object.__setattr__(self, attr, val)
        elif isinstance(val, Param):            # `ps.theta23 = other_param` replaces the object (param.py:1312-1314)
            assert val.name == attr
            if self._params[i] is not val:
                self._params[i] = val
                self._bump()
        else:
            self._params[i].value = val

    names = property(lambda self: tuple(p.name for p in self._params))
    def _each(attr):            # noqa: N805 -- one attribute of every param, read as a tuple, set from a sequence
        def get(self):
            return tuple(getattr(p, attr) for p in self._params)

        def put(self, vals):
            assert len(vals) == len(self._params)
            for p, v in zip(self._params, vals):
                setattr(p, attr, v)
        return property(get, put)

    values = _each("value")
    nominal_values = _each("nominal_value")
    priors = _each("prior")
    ranges = _each("range")
    del _each
    free = property(lambda self: ParamSet([p for p in self._params if not p.is_fixed]))
    fixed = property(lambda self: ParamSet([p for p in self._params if p.is_fixed]))
    are_fixed = property(lambda self: tuple(p.is_fixed for p in self._params))
    continuous = property(lambda self: ParamSet([p for p in self._params if not p.is_discrete]))
    discrete = property(lambda self: ParamSet([p for p in self._params if p.is_discrete]))
    are_discrete = property(lambda self: tuple(p.is_discrete for p in self._params))
    tex = property(lambda self: r",\,".join(p.tex for p in self._params))
    name_val_dict = property(lambda self: OrderedDict((p.name, p.value) for p in self._params))
    is_nominal = property(lambda self: all(_same(p.value, p.nominal_value) for p in self._params))
    state = property(lambda self: tuple(p.state for p in self._params))

    serializable_state = property(lambda self: [p.serializable_state for p in self._params])

    def to_json(self, filename, **kwargs):
        """a JSON file of the params' states, read back by `ParamSet.from_json` (param.py:1590-1601)"""
        from pisa_amd.utils import jsons

        jsons.to_json(self.serializable_state, filename, **kwargs)

    @classmethod
    def from_json(cls, filename):
        from pisa_amd.utils import jsons

        return cls([Param(**st) for st in jsons.from_json(filename)])

    def set_values(self, new_params):
        """values of the params of the same names in `new_params` (param.py:1116-1129)"""
        for p in new_params:
            self[p.name].value = p.value

    def update_existing(self, obj):
        self.update(obj, existing_must_match=False, extend=False)

    def priors_penalties(self, metric):
        return [p.prior_penalty(metric) for p in self._params]
    has_derived = property(lambda self: any(isinstance(p, DerivedParam) for p in self._params))

    def add_covariance(self, covmat):
        """Correlated priors (param.py:949-1097): `covmat` = {name: {name: covariance}} over some of the set's
        parameters.  With x the correlated parameters, mu their prior means (a uniform prior: the middle of the
        range) and T the eigenvectors of the covariance, the fit runs in v = (x - mu) T: one new free parameter
        `<name>_rotated` per x, Gaussian prior of width sqrt(eigenvalue) around 0, ranges from the corners of the
        x ranges; every x becomes a `DerivedParam` x_i = sum_j v_j (T^-1)_ji + mu_i."""
        names = list(covmat.keys())
        dim = len(names)
        if dim == 0:
            return
        cov = np.zeros((dim, dim))
        for i, key in enumerate(names):
            if key not in self.names:
                raise KeyError("Key %s not in Params" % key)
            if not isinstance(covmat[key], Mapping):
                raise TypeError("Each entry in covmat should be another dict, found %s" % type(covmat[key]))
            for j, sub in enumerate(covmat[key].keys()):
                if sub not in self.names:
                    raise KeyError("Key %s not in Params" % sub)
                cov[i][j] = covmat[key][sub]
        if np.linalg.det(cov) < 0:
            raise ValueError("Covariance matrix *must* be positive definite!")
        params = [self[n] for n in names]
        means = []
        for prm in params:
            if prm.prior is not None and prm.prior.kind == "gaussian":
                means.append(float(prm.prior.mean.m_as(prm.units)))
            elif prm.prior is not None and prm.prior.kind == "uniform":
                means.append(0.5 * float((prm.range[1] + prm.range[0]).m_as(prm.units)))
            else:
                raise NotImplementedError("prior mean of '%s' (%s)" % (prm.name, prm.prior))
        evals, inv_t = np.linalg.eig(cov)
        sigmas = np.sqrt(evals)
        if any(abs(sg) < 1e-20 for sg in sigmas):
            raise ValueError("Found zero-width param %s - your parameters might be linearly dependent!" % (sigmas,))
        transformation = np.linalg.inv(inv_t)
        ranges = [[float(r.m_as(prm.units)) for r in prm.range] for prm in params]
        rotated = []
        for i, prm in enumerate(params):
            v_max = v_min = 0.0
            for j in range(dim):
                t = inv_t[j][i]
                hi, lo = t * (ranges[j][1] - means[j]), t * (ranges[j][0] - means[j])
                v_max += hi if t > 0 else lo
                v_min += hi if t < 0 else lo
            new = Param(name=prm.name + "_rotated", value=0.0 * ureg.dimensionless,
                        prior=Prior(kind="gaussian", mean=0.0, stddev=float(sigmas[i])), range=(v_min, v_max),
                        is_fixed=False, is_discrete=False, scales_as_log=prm.scales_as_log,
                        nominal_value=0.0 * ureg.dimensionless, tex=prm.tex + "'")
            rotated.append(new)
            self.update(new)
        for i, prm in enumerate(params):
            derived = DerivedParam(name=prm.name, value=prm.value, range=prm.range)
            derived.dependson = rotated
            derived.callable = LinearFunction([r.name for r in rotated], [transformation[j][i] for j in range(dim)],
                                              means[i])
            self.replace(derived)

    def index(self, name):
        if isinstance(name, Param):
            name = name.name
        if isinstance(name, (int, np.integer)):
            return int(name)
        i = self._pos(name)
        if i is None:
            raise ValueError("'%s' is not a parameter of this set" % name)
        return i

    def fix(self, x):
        for n in ([x] if isinstance(x, (str, Param)) else x):
            self[self.index(n)].is_fixed = True

    def unfix(self, x):
        for n in ([x] if isinstance(x, (str, Param)) else x):
            self[self.index(n)].is_fixed = False

    def extend(self, obj):
        new = [obj] if isinstance(obj, Param) else list(obj)
        for p in new:
            if p.name in self.names:
                raise ValueError("parameter '%s' already present" % p.name)
        for p in new:
            self._index[p.name] = len(self._params)
            self._params.append(p)
        self._bump()

    def replace(self, new):
        self._params[self.index(new.name)] = new
        self._bump()

    def update(self, obj, existing_must_match=False, extend=True):
        """param.py:1221-1260"""
        new = [obj] if isinstance(obj, Param) else list(obj)
        for p in new:
            i = self._pos(p.name)
            if i is not None:
                if existing_must_match and p._hashable() != self._params[i]._hashable():
                    raise ValueError("Param '%s' specified in multiple stages with different values"
                                     % p.name)
                if self._params[i] is not p:
                    self._params[i] = p
                    self._bump()
            elif extend:
                self._index[p.name] = len(self._params)
                self._params.append(p)
                self._bump()

    def reset_all(self):
        for p in self._params:
            p.reset()

    def reset_free(self):
        for p in self.free:
            p.reset()

    def set_nominal_by_current_values(self):
        for p in self._params:
            p.set_nominal_to_current_value()

    def randomize_free(self, random_state=None):
        rs = random_state if isinstance(random_state, np.random.RandomState) \
            else np.random.RandomState(random_state)
        free = self.free
        free._rescaled_values = rs.rand(len(free))

    @property
    def _rescaled_values(self):
        return tuple(p._rescaled_value for p in self._params)

    @_rescaled_values.setter
    def _rescaled_values(self, vals):
        assert len(vals) == len(self)
        for p, v in zip(self._params, vals):
            p._rescaled_value = v

    def priors_penalty(self, metric):
        # the same sum over the same terms in the same order (param.py:1372-1396); a term is re-evaluated
        # only when its parameter moved
        # (a fit asks at every point and moves one or two parameters between points: the terms live in an array,
        # the entries of moved parameters are replaced, np.sum adds the same float64 terms in the same order)
        params = self._params
        c = self.__dict__.get("_pen_terms")
        if c is None or c[0] != metric or c[1] != ParamSet.struct_clock or len(c[2]) != len(params):
            c = (metric, ParamSet.struct_clock, [None] * len(params), np.zeros(len(params)), [None] * len(params))
            object.__setattr__(self, "_pen_terms", c)
        vers, arr, priors = c[2], c[3], c[4]
        try:
            for i, p in enumerate(params):
                if vers[i] != p._ver or priors[i] is not p.prior:
                    arr[i] = p.prior_penalty(metric)
                    vers[i], priors[i] = p._ver, p.prior
        except (TypeError, ValueError):     # a term that is not a scalar: the plain sum
            object.__setattr__(self, "_pen_terms", None)
            return np.sum([p.prior_penalty_cached(metric) for p in self._params])
        return np.sum(arr)

    @property
    def values_hash(self):
        return hash(tuple(p._hashable() for p in self._params))

    @property
    def hash(self):
        return hash(tuple((p.name, p._hashable(), p.is_fixed) for p in self._params))

    def __repr__(self):
        return "ParamSet(\n  %s\n)" % "\n  ".join(repr(p) for p in self._params)


class ParamSelector:
    """Regular params + alternative param sets keyed by selector
    (param.py:1604-1900): `param.nh.theta23` / `param.ih.theta23`."""

    def __init__(self, regular_params=None, selector_param_sets=None, selections=None):
        self._regular = ParamSet(regular_params) if regular_params is not None else ParamSet()
        self._selector_sets = OrderedDict()
        if selector_param_sets:
            for sel, ps in selector_param_sets.items():
                self._selector_sets[sel.strip().lower()] = ParamSet(ps)
        self._selections = []
        self._current = ParamSet(list(self._regular))
        self.select_params(selections, error_on_missing=False)

    @property
    def params(self):
        return self._current

    @property
    def param_selections(self):
        return list(self._selections)

    def select_params(self, selections=None, error_on_missing=False):
        if selections is None:
            return self._current
        if isinstance(selections, str):
            selections = [s.strip() for s in selections.split(",")]
        found = False
        for sel in selections:
            if sel is None:
                continue
            key = sel.strip().lower()
            if key not in self._selector_sets:
                continue
            found = True
            # one selection per "dimension": drop selections sharing param names
            new_names = set(self._selector_sets[key].names)
            self._selections = [s for s in self._selections
                                if not (set(self._selector_sets[s].names) & new_names)]
            self._selections.append(key)
            self._current.update(self._selector_sets[key], extend=True)
        if error_on_missing and not found and len(self._selector_sets) > 0:
            raise KeyError("none of the selections %s present" % (selections,))
        return self._current

    def update(self, p, selector=None, existing_must_match=False, extend=True):
        """Update params; params shared by name across stages become one object
        (pipeline.py:342-346; param.py:1708-1730: without a selector the regular and the current sets take
        the params, with one that selector's set does and the current selection is applied again)."""
        new = [p] if isinstance(p, Param) else list(p)
        if selector is not None:
            self._selector_sets.setdefault(selector.strip().lower(), ParamSet()).update(
                new, existing_must_match=existing_must_match, extend=extend)
            if selector.strip().lower() in self._selections:
                self.select_params([selector], error_on_missing=False)
            return
        for q in new:
            if q.name in self._regular.names:
                self._regular.update(q, existing_must_match=existing_must_match)
            for sel, ps in self._selector_sets.items():
                if q.name in ps.names and sel in self._selections:
                    ps.update(q)
            if q.name in self._current.names:
                self._current.update(q, existing_must_match=False)
            elif extend:
                self._regular.extend(q)
                self._current.extend(q)

    def __iter__(self):
        return iter(self._current)

    def __eq__(self, other):
        """same selections, regular params and per-selector sets (param.py:1691-1700)"""
        if not isinstance(other, ParamSelector):
            return False
        return (sorted(self._selections) == sorted(other._selections) and self._regular == other._regular
                and self._selector_sets.keys() == other._selector_sets.keys()
                and all(self._selector_sets[k] == other._selector_sets[k] for k in self._selector_sets))

    __hash__ = object.__hash__

    def get(self, name, selector=None):
        if selector is None:
            return self._regular[name]
        return self._selector_sets[selector.strip().lower()][name]
