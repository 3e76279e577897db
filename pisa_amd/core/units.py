"""A small physical-units layer with the slice of the `pint` API that PISA
configs and services use (pisa/__init__.py:89 creates `ureg`; services call
`param.value.m_as('rad')`, cfgs write `42. * units.degree`,
`[0.001, 0.007] * units.eV**2`, `2.5 * units.common_year`).

Only what the hot path's host code needs: multiplicative units over the base
dimensions (length, time, mass, energy, angle), parsing of unit expressions,
`Quantity` arithmetic with scalars/arrays, `.m`, `.magnitude`, `.units`,
`.u`, `.m_as()`, `.to()`, `.dimensionality`.  Nothing of this runs in the
per-event loop; per eval it costs a few dict look-ups (the reference spends
milliseconds in pint here).
"""
import ast
import math
import operator
import re

import numpy as np

__all__ = ["ureg", "Quantity", "Unit", "DimensionalityError", "Q_"]

_BASE = ("length", "time", "mass", "energy", "angle")


class DimensionalityError(ValueError):
    pass


class Unit:
    """scale * prod(base_i ** dims_i); `name` keeps the user's spelling."""

    __slots__ = ("scale", "dims", "name")
    __array_ufunc__ = None  # numpy defers to __rmul__ / __rtruediv__

    def __init__(self, scale=1.0, dims=None, name="dimensionless"):
        self.scale = float(scale)
        self.dims = tuple(dims) if dims is not None else (0,) * len(_BASE)
        self.name = name

    @property
    def dimensionless(self):
        return all(d == 0 for d in self.dims)

    @property
    def dimensionality(self):
        return {b: d for b, d in zip(_BASE, self.dims) if d != 0}

    def _combine(self, other, sign):
        dims = tuple(a + sign * b for a, b in zip(self.dims, other.dims))
        scale = self.scale * other.scale ** sign
        if self.dimensionless and self.name == "dimensionless":
            name = other.name if sign > 0 else "1 / %s" % ("(%s)" % other.name if " " in other.name else other.name)
        elif other.name == "dimensionless":
            name = self.name
        else:
            rhs = "(%s)" % other.name if (sign < 0 and " " in other.name) else other.name
            name = "%s %s %s" % (self.name, "*" if sign > 0 else "/", rhs)
        return Unit(scale, dims, name)

    def __mul__(self, other):
        if isinstance(other, Unit):
            return self._combine(other, +1)
        if isinstance(other, Quantity):
            return Quantity(other.magnitude, self._combine(other.units, +1))
        return Quantity(other, self)

    __rmul__ = __mul__

    def __truediv__(self, other):
        if isinstance(other, Unit):
            return self._combine(other, -1)
        return Quantity(1.0 / np.asarray(other), self)

    def __rtruediv__(self, other):
        return Quantity(other, Unit() / self)

    def __pow__(self, p):
        name = self.name if self.name == "dimensionless" else \
            "%s ** %s" % ("(%s)" % self.name if " " in self.name else self.name, p)
        return Unit(self.scale ** p, tuple(d * p for d in self.dims), name)

    def __eq__(self, other):
        if isinstance(other, str):
            other = ureg.parse_units(other)
        if not isinstance(other, Unit):
            return NotImplemented
        return self.dims == other.dims and math.isclose(self.scale, other.scale, rel_tol=1e-15)

    def __hash__(self):
        return hash((self.dims, round(math.log(self.scale), 12) if self.scale > 0 else 0))

    def __repr__(self):
        return "<Unit('%s')>" % self.name

    def __str__(self):
        return self.name


class Quantity:
    __slots__ = ("_m", "_u")
    __array_priority__ = 100
    __array_ufunc__ = None

    def __init__(self, magnitude, units=None):
        if isinstance(magnitude, Quantity):
            units = magnitude.units if units is None else units
            magnitude = magnitude.magnitude
        if isinstance(units, str):
            units = ureg.parse_units(units)
        if isinstance(magnitude, (list, tuple)):
            magnitude = np.array(magnitude, dtype=np.float64)
        elif isinstance(magnitude, str):            # Quantity("0.1 dimensionless"), as pint parses it
            q = ureg.parse_expression(magnitude)
            if isinstance(q, Unit):
                q = Quantity(1, q)
            if isinstance(q, Quantity):
                magnitude, units = (q.magnitude, q.units) if units is None else (q.m_as(units), units)
            else:
                magnitude = q
        self._m = magnitude
        self._u = units if units is not None else Unit()

    magnitude = property(lambda self: self._m)
    m = magnitude
    units = property(lambda self: self._u)
    u = units
    dimensionality = property(lambda self: self._u.dimensionality)
    dimensionless = property(lambda self: self._u.dimensionless)

    def to(self, units):
        if isinstance(units, str):
            units = ureg.parse_units(units)
        if isinstance(units, Quantity):
            units = units.units
        if units.dims != self._u.dims and units.dims[:-1] != self._u.dims[:-1]:
            raise DimensionalityError("Cannot convert from '%s' to '%s'" % (self._u, units))
        factor = self._u.scale / units.scale
        if factor == 1.0:
            return Quantity(self._m, units)
        return Quantity(np.asarray(self._m) * factor if not np.isscalar(self._m) else self._m * factor,
                        units)

    def m_as(self, units):
        # same arithmetic as `to(units).magnitude`, without building the intermediate Quantity
        if isinstance(units, str):
            units = ureg.parse_units(units)
        elif isinstance(units, Quantity):
            units = units.units
        mine = self._u
        if units is mine:
            return self._m
        if units.dims != mine.dims and units.dims[:-1] != mine.dims[:-1]:
            # pint's radian is dimensionless: `theta.m_as('dimensionless')` is the angle in radians (the last base
            # dimension here is the angle, kept apart only so that names survive)
            raise DimensionalityError("Cannot convert from '%s' to '%s'" % (mine, units))
        factor = mine.scale / units.scale
        if factor == 1.0:
            return self._m
        return np.asarray(self._m) * factor if not np.isscalar(self._m) else self._m * factor

    def ito(self, units):
        q = self.to(units)
        self._m, self._u = q._m, q._u

    def to_tuple(self):
        """(magnitude, ((unit name, power), ...)): pint's `Quantity.to_tuple`, the form the reference's JSON
        files hold (utils/jsons.py:300-302)"""
        return self.magnitude, _name_powers(self.units.name)

    @classmethod
    def from_tuple(cls, tup):
        m, parts = tup
        u = ureg.dimensionless
        for name, power in parts:
            b = ureg.parse_units(name)
            u = u * (b if power == 1 else b ** (int(power) if float(power).is_integer() else power))
        return cls(m, u)

    def to_base_units(self):
        return Quantity(np.asarray(self._m) * self._u.scale if not np.isscalar(self._m)
                        else self._m * self._u.scale,
                        Unit(1.0, self._u.dims, _base_name(self._u.dims)))

    def _coerce(self, other):
        if isinstance(other, Quantity):
            return other.to(self._u).magnitude
        if isinstance(other, Unit):
            return Quantity(1.0, other).to(self._u).magnitude
        if not self._u.dimensionless:
            if np.all(np.asarray(other) == 0):
                return other
            raise DimensionalityError("Cannot combine '%s' with a bare number" % self._u)
        return np.asarray(other) / self._u.scale if not np.isscalar(other) else other / self._u.scale

    def __add__(self, other):
        return Quantity(self._m + self._coerce(other), self._u)

    __radd__ = __add__

    def __sub__(self, other):
        return Quantity(self._m - self._coerce(other), self._u)

    def __rsub__(self, other):
        return Quantity(self._coerce(other) - self._m, self._u)

    def __neg__(self):
        return Quantity(-self._m, self._u)

    def __abs__(self):
        return Quantity(abs(self._m), self._u)

    def __mul__(self, other):
        if isinstance(other, Quantity):
            return Quantity(self._m * other._m, self._u * other._u)
        if isinstance(other, Unit):
            return Quantity(self._m, self._u * other)
        return Quantity(self._m * other, self._u)

    __rmul__ = __mul__

    def __truediv__(self, other):
        if isinstance(other, Quantity):
            return Quantity(self._m / other._m, self._u / other._u)
        if isinstance(other, Unit):
            return Quantity(self._m, self._u / other)
        return Quantity(self._m / other, self._u)

    def __rtruediv__(self, other):
        return Quantity(other / self._m, Unit() / self._u)

    def __pow__(self, p):
        return Quantity(self._m ** p, self._u ** p)

    def _cmp(self, other, op):
        return op(self._m, self._coerce(other))

    def __eq__(self, other):
        try:
            return self._cmp(other, operator.eq)
        except DimensionalityError:
            return False

    def __ne__(self, other):
        r = self.__eq__(other)
        return ~r if isinstance(r, np.ndarray) else not r

    __hash__ = None

    def __lt__(self, other):
        return self._cmp(other, operator.lt)

    def __le__(self, other):
        return self._cmp(other, operator.le)

    def __gt__(self, other):
        return self._cmp(other, operator.gt)

    def __ge__(self, other):
        return self._cmp(other, operator.ge)

    def __float__(self):
        if not self._u.dimensionless:
            raise DimensionalityError("only dimensionless quantities convert to float")
        return float(self._m * self._u.scale)

    def __bool__(self):
        # pint: the truth value of a quantity is that of its magnitude (arrays: numpy's rule)
        return bool(self._m)

    def __len__(self):
        return len(self._m)

    def __iter__(self):
        for v in self._m:
            yield Quantity(v, self._u)

    def __getitem__(self, idx):
        return Quantity(self._m[idx], self._u)

    def __repr__(self):
        return "<Quantity(%r, '%s')>" % (self._m, self._u)

    def __str__(self):
        return "%s %s" % (self._m, self._u)

    def __format__(self, spec):
        return "%s %s" % (format(self._m, spec), self._u)


Q_ = Quantity


_BASE_UNIT_NAMES = {"length": "meter", "time": "second", "mass": "gram", "energy": "electron_volt", "angle": "radian"}


def _base_name(dims):
    """'meter / second ** 2' for the dimensions of a unit with scale 1"""
    parts = ["%s ** %g" % (_BASE_UNIT_NAMES.get(b, b), d) if d != 1 else _BASE_UNIT_NAMES.get(b, b)
             for b, d in zip(_BASE, dims) if d != 0]
    return " * ".join(parts) if parts else "dimensionless"


def _name_powers(name):
    """'meter / second ** 2' -> (('meter', 1.0), ('second', -2.0)); a name that is not a product of powers
    of registered units (a numeric factor in it) stays one entry, readable by `parse_units`"""
    if name == "dimensionless":
        return ()
    acc = {}

    def walk(node, sign):
        if isinstance(node, ast.Name):
            acc[node.id] = acc.get(node.id, 0.0) + sign
        elif isinstance(node, ast.BinOp) and isinstance(node.op, (ast.Mult, ast.Div)):
            walk(node.left, sign)
            walk(node.right, sign if isinstance(node.op, ast.Mult) else -sign)
        elif isinstance(node, ast.BinOp) and isinstance(node.op, ast.Pow) and isinstance(node.right, ast.Constant):
            walk(node.left, sign * float(node.right.value))
        elif isinstance(node, ast.BinOp) and isinstance(node.op, ast.Pow) and isinstance(node.right, ast.UnaryOp) \
                and isinstance(node.right.op, ast.USub) and isinstance(node.right.operand, ast.Constant):
            walk(node.left, -sign * float(node.right.operand.value))
        elif isinstance(node, ast.Constant) and node.value == 1:
            pass
        else:
            raise ValueError(name)

    try:
        walk(ast.parse(name.replace("^", "**"), mode="eval").body, 1.0)
    except (ValueError, SyntaxError):
        return ((name, 1.0),)
    return tuple((n, p) for n, p in acc.items() if p != 0)


def _dims(**kw):
    return tuple(kw.get(b, 0) for b in _BASE)


class UnitRegistry:
    """Attribute / item access yields `Unit`s: ureg.km, ureg['eV'], ureg.eV**2."""

    def __init__(self):
        L, T, M, E, A = (_dims(length=1), _dims(time=1), _dims(mass=1), _dims(energy=1),
                         _dims(angle=1))
        u = {}

        def add(names, scale, dims):
            for n in names:
                u[n] = Unit(scale, dims, names[0])

        add(["dimensionless", "none", "unitless"], 1.0, _dims())
        add(["percent"], 0.01, _dims())
        # angle (radian is the base, so m_as('rad') of a degree value multiplies by pi/180)
        add(["radian", "rad", "radians"], 1.0, A)
        add(["degree", "deg", "degrees", "arcdeg"], math.pi / 180.0, A)
        # length
        add(["meter", "m", "metre", "meters"], 1.0, L)
        add(["kilometer", "km", "kilometre", "kilometers"], 1e3, L)
        add(["centimeter", "cm", "centimetre"], 1e-2, L)
        add(["millimeter", "mm"], 1e-3, L)
        add(["micrometer", "um", "micron"], 1e-6, L)
        add(["nanometer", "nm"], 1e-9, L)
        add(["foot", "ft", "feet"], 0.3048, L)
        add(["inch"], 0.0254, L)
        add(["mile", "mi"], 1609.344, L)
        # time
        add(["second", "s", "sec", "seconds"], 1.0, T)
        add(["millisecond", "ms"], 1e-3, T)
        add(["microsecond", "us"], 1e-6, T)
        add(["nanosecond", "ns"], 1e-9, T)
        add(["minute", "min"], 60.0, T)
        add(["hour", "hr", "h"], 3600.0, T)
        add(["day", "d", "days"], 86400.0, T)
        add(["common_year", "common_years"], 365 * 86400.0, T)
        add(["year", "yr", "julian_year", "years", "a"], 365.25 * 86400.0, T)
        add(["hertz", "Hz"], 1.0, _dims(time=-1))
        # mass
        add(["gram", "g"], 1.0, M)
        add(["kilogram", "kg"], 1e3, M)
        # energy (eV is the base)
        add(["electron_volt", "eV", "electronvolt"], 1.0, E)
        for pre, f in (("meV", 1e-3), ("keV", 1e3), ("MeV", 1e6), ("GeV", 1e9), ("TeV", 1e12),
                       ("PeV", 1e15)):
            add([pre], f, E)
        add(["joule", "J"], 1.0 / 1.602176634e-19, E)
        self._units = u

    def __getattr__(self, name):
        try:
            return self.__dict__["_units"][name]
        except KeyError:
            raise AttributeError("undefined unit '%s'" % name)

    def __getitem__(self, name):
        return self.parse_units(name)

    def __call__(self, expr):
        return self.parse_expression(expr)

    def Quantity(self, value, units=None):  # noqa: N802 (pint spelling)
        return Quantity(value, units)

    def Unit(self, expr):  # noqa: N802
        return self.parse_units(expr)

    def parse_units(self, expr):
        if isinstance(expr, Unit):
            return expr
        expr = str(expr).strip()
        if expr in ("", "dimensionless"):
            return self._units["dimensionless"]
        if expr in self._units:
            return self._units[expr]
        cache = self.__dict__.setdefault("_parsed", {})
        hit = cache.get(expr)
        if hit is not None:
            return hit
        q = self.parse_expression(expr)
        if isinstance(q, Quantity):
            q = Unit(q.units.scale * float(q.magnitude), q.units.dims, expr)
        elif not isinstance(q, Unit):
            q = Unit(float(q), None, expr)
        cache[expr] = q   # compound expressions ("eV**2", "g/cm**3") are parsed once
        return q

    def parse_expression(self, expr):
        """Evaluate e.g. '2.5 * common_year', 'eV**2', 'g/cm**3', '1e-3 eV ** 2'."""
        s = str(expr).strip().replace("^", "**")
        # implicit multiplication: "33.48 deg" -> "33.48 * deg"
        s = re.sub(r"(?<=[0-9.)])\s+(?=[A-Za-z_(])", " * ", s)
        s = re.sub(r"(?<=[A-Za-z_)])\s+(?=[A-Za-z_(])", " * ", s)
        tree = ast.parse(s, mode="eval")
        return self._eval(tree.body)

    def _eval(self, node):
        if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)):
            return node.value
        if isinstance(node, ast.Name):
            if node.id in self._units:
                return self._units[node.id]
            raise AttributeError("undefined unit '%s'" % node.id)
        if isinstance(node, ast.BinOp):
            a, b = self._eval(node.left), self._eval(node.right)
            ops = {ast.Mult: operator.mul, ast.Div: operator.truediv, ast.Pow: operator.pow,
                   ast.Add: operator.add, ast.Sub: operator.sub}
            return ops[type(node.op)](a, b)
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, ast.USub):
            return -self._eval(node.operand)
        raise ValueError("cannot parse unit expression")


ureg = UnitRegistry()
