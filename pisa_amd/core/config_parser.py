"""Pipeline-config reader: accepts the reference's cfg grammar
(pisa/utils/config_parser.py:1-216 module docstring) so
that e.g. `settings/pipeline/osc_example.cfg` is usable as is:

* `#include <resource> as <section>` / `#include <resource>` pre-processing
  (config_parser.py:1156-1279), resources found via `PISA_RESOURCES` or the
  packaged `pisa_amd/resources`;
* `${section:key}` interpolation (ExtendedInterpolation), case-sensitive keys;
* `[binning]`: `<name>.order`, `<name>.<dim> = {dict}` evaluated with `np`,
  `units`, `inf` in scope (config_parser.py:646-697);
* `[pipeline]`: order, name, param_selections, output_binning, output_key;
* `[stage.service]`: constructor kwargs, `calc_mode` / `apply_mode`,
  `*_names` lists, `true/false/none` literals, `units.` quantities;
* `param.[<selector>.]<name> = <quantity>` with `.fixed .range .prior .tex
  .scales_as_log`; `a +/- b` gives a Gaussian prior (config_parser.py:453-565).

Returns the same structure as `parse_pipeline_config` (config_parser.py:700-958):
an OrderedDict with key 'pipeline' and (stage, service) tuples.
"""
import configparser
import math
import re
from collections import OrderedDict
from collections.abc import Mapping

import numpy as np

from pisa_amd import FTYPE
from pisa_amd.core.binning import MultiDimBinning, OneDimBinning, VarBinning
from pisa_amd.core.param import Param, ParamSelector, Prior
from pisa_amd.core.units import Quantity, ureg
from pisa_amd.utils.resources import find_resource

__all__ = ["parse_pipeline_config", "parse_quantity", "parse_string_literal", "PISAConfigParser"]

PARAM_RE = re.compile(r"^param\.(?P<subfields>(([^.\s]+)(\.|$))+)", re.IGNORECASE)
PARAM_ATTRS = ["range", "prior", "fixed", "tex", "scales_as_log"]
INCLUDE_AS_RE = re.compile(r"^\s*#include\s+(?P<file>\S+)\s+as\s+(?P<as>\S+)\s*$")
INCLUDE_RE = re.compile(r"^\s*#include\s+(?P<file>\S+)\s*$")

units = ureg  # name available to eval'd cfg expressions
inf = np.inf


def split(string, sep=","):
    return [x.strip() for x in str(string).split(sep) if x.strip() != ""]


def _read_with_includes(path, as_section=None, seen=None):
    seen = seen or set()
    path = find_resource(path)
    if path in seen:
        raise ValueError("recursive #include of %s" % path)
    seen = seen | {path}
    out = []
    if as_section is not None:
        out.append("[%s]" % as_section)
    with open(path) as fh:
        for line in fh:
            m = INCLUDE_AS_RE.match(line)
            if m:
                out.extend(_read_with_includes(m.group("file"), m.group("as"), seen))
                continue
            m = INCLUDE_RE.match(line)
            if m:
                out.extend(_read_with_includes(m.group("file"), None, seen))
                continue
            out.append(line.rstrip("\n"))
    return out


class PISAConfigParser(configparser.ConfigParser):
    def __init__(self):
        super().__init__(interpolation=configparser.ExtendedInterpolation(), empty_lines_in_values=False)
        self.optionxform = str  # case-sensitive keys

    def read(self, filenames, encoding=None):
        if isinstance(filenames, str):
            filenames = [filenames]
        for fn in filenames:
            lines = _read_with_includes(fn)
            # included sections must precede the including file's own sections
            self.read_string("\n".join(lines))
        return filenames


def parse_string_literal(string):
    s = string.strip().lower()
    if s == "true":
        return True
    if s == "false":
        return False
    if s == "none":
        return None
    return string


class _UQuantity:
    """value +/- std_dev with units (stand-in for pint+uncertainties)"""

    def __init__(self, n, s, u):
        self.nominal_value, self.std_dev, self.units = n, s, u

    n = property(lambda self: self.nominal_value)
    s = property(lambda self: self.std_dev)


def parse_quantity(string):
    """'1.2 +/- 0.7 * units.meter', '33.48 units.deg', '7.5e-5 units.eV**2', '1e4'
    (config_parser.py:303-353)."""
    value = string.replace(" ", "")
    if "units." in value:
        value, unit = value.split("units.", 1)
    else:
        unit = None
    value = value.rstrip("*")
    if "+/-" in value:
        n, s = value.split("+/-")
        n, s = float(n), float(s)
    else:
        n, s = float(value), float("nan")
    u = ureg.parse_units(unit) if unit else ureg.dimensionless
    return _UQuantity(n, s, u)


def _parse_multidimbinning(config, binning, order):
    dims = []
    for bin_name in order:
        raw = config.get("binning", binning + "." + bin_name)
        kwargs = eval(raw, {"np": np, "numpy": np, "units": ureg, "inf": np.inf})  # pylint: disable=eval-used
        dims.append(OneDimBinning(name=bin_name, **kwargs))
    return MultiDimBinning(dims, name=binning)


_BINNING_NS = {"np": np, "numpy": np, "units": ureg, "inf": np.inf}


def _parse_varbinning(config, binning, order, bin_split):
    """`<name>.split` = a dict (the `OneDimBinning` whose bins select the events) or comma-separated cut
    expressions; a dimension's entry is one dict for all selections or a list with one per selection; an
    optional `<name>.mask` likewise (config_parser.py:584-644)"""
    try:
        parsed = eval(bin_split, dict(_BINNING_NS))  # pylint: disable=eval-used
    except Exception:  # pylint: disable=broad-except
        parsed = None
    if isinstance(parsed, Mapping):
        selections = OneDimBinning(**parsed)
    else:
        selections = split(bin_split)
    n = len(selections)
    dims = [[] for _ in range(n)]
    for bin_name in order:
        kwargs = eval(config.get("binning", binning + "." + bin_name), dict(_BINNING_NS))  # pylint: disable=eval-used
        if isinstance(kwargs, list):
            assert len(kwargs) == n
        else:
            kwargs = [kwargs] * n
        for i, kw in enumerate(kwargs):
            dims[i].append(OneDimBinning(name=bin_name, **kw))
    mask = config["binning"].get(binning + ".mask", None)
    if mask is not None:
        mask = eval(mask, dict(_BINNING_NS))  # pylint: disable=eval-used
        if not all(np.ndim(m) == len(order) for m in mask):     # ONE mask (as many levels as dimensions):
            mask = [mask] * n                                     # for every selection
        assert len(mask) == n
    else:
        mask = [None] * n
    return VarBinning(binnings=[MultiDimBinning(dims[i], name="%s_%d" % (binning, i), mask=mask[i]) for i in range(n)],
                      selections=selections)


def _param_subfields(subfields):
    """param.<selector>.<name>.<attr> decomposition (config_parser.py:394-451)"""
    selector = pname = attr = None
    fields = list(subfields)
    if fields and fields[-1] in PARAM_ATTRS:
        attr = fields.pop()
    elif len(fields) >= 2 and fields[-2] == "prior":  # e.g. .prior.data
        attr = ".".join(fields[-2:])
        fields = fields[:-2]
    if len(fields) == 1:
        pname = fields[0]
    elif len(fields) == 2:
        selector, pname = fields
    else:
        raise ValueError("cannot interpret param spec '%s'" % ".".join(subfields))
    return selector, pname, attr


def parse_param(config, section, selector, fullname, pname, value):
    kwargs = dict(name=pname, is_fixed=True, prior=None, range=None)
    uq = None
    try:
        uq = parse_quantity(value)
        kwargs["value"] = Quantity(uq.nominal_value, uq.units)
    except ValueError:
        kwargs["value"] = parse_string_literal(value)
    if config.has_option(section, fullname + ".fixed"):
        kwargs["is_fixed"] = config.getboolean(section, fullname + ".fixed")
    if config.has_option(section, fullname + ".scales_as_log"):
        kwargs["scales_as_log"] = config.getboolean(section, fullname + ".scales_as_log")
    if config.has_option(section, fullname + ".tex"):
        kwargs["tex"] = config.get(section, fullname + ".tex")
    if config.has_option(section, fullname + ".range"):
        range_ = config.get(section, fullname + ".range")
        scope = {"np": np, "numpy": np, "units": ureg, "inf": np.inf, "FTYPE": FTYPE}
        if "nominal" in range_:
            scope["nominal"] = Quantity(uq.n, uq.units)
        if "sigma" in range_:
            scope["sigma"] = Quantity(uq.s, uq.units)
        range_ = range_.replace("[", "np.array([").replace("]", "], dtype=FTYPE)")
        rng = eval(range_, scope)  # pylint: disable=eval-used
        if not isinstance(rng, Quantity):
            rng = Quantity(rng, uq.units if uq is not None else ureg.dimensionless)
        kwargs["range"] = rng.to(uq.units)
    if config.has_option(section, fullname + ".prior"):
        prior = str(config.get(section, fullname + ".prior")).strip().lower()
        if prior == "uniform":
            kwargs["prior"] = Prior(kind="uniform")
        elif prior == "jeffreys":
            kwargs["prior"] = Prior(kind="jeffreys", A=kwargs["range"][0], B=kwargs["range"][1])
        elif prior == "none":
            kwargs["prior"] = None
        elif prior == "spline":
            # config_parser.py:541-553: knots / coeffs / deg from a JSON resource, entry
            # "<name>[_<selector>]", knots converted to the parameter's units
            import json

            from pisa_amd.utils.resources import find_resource

            priorname = pname if selector is None else pname + "_" + selector
            with open(find_resource(config.get(section, fullname + ".prior.data"))) as fh:
                data = json.load(fh)[priorname]
            knots = Quantity(np.asarray(data["knots"], dtype=np.float64), data["units"]).to(kwargs["value"].units)
            kwargs["prior"] = Prior(kind="spline", knots=knots, coeffs=np.asarray(data["coeffs"], dtype=np.float64),
                                    deg=int(data["deg"]))
        else:
            raise Exception("Prior type unknown")
    elif uq is not None and not math.isnan(uq.std_dev):
        kwargs["prior"] = Prior(kind="gaussian", mean=Quantity(uq.n, uq.units),
                                stddev=Quantity(uq.s, uq.units))
    return Param(**kwargs)


def parse_pipeline_config(config):
    if isinstance(config, str):
        cfg = PISAConfigParser()
        cfg.read(config)
        config = cfg
    elif not isinstance(config, PISAConfigParser):
        raise TypeError("`config` must either be a string or PISAConfigParser. Got %s instead."
                        % type(config))
    if not config.has_section("binning"):
        raise configparser.NoSectionError("binning")
    binning_dict = {}
    for name in config["binning"]:
        if name.endswith(".order"):
            order = split(config.get("binning", name))
            binning = name[: -len(".order")]
            bin_split = config["binning"].get(binning + ".split", None)
            if bin_split is not None:
                binning_dict[binning] = _parse_varbinning(config, binning, order, bin_split)
            else:
                binning_dict[binning] = _parse_multidimbinning(config, binning, order)

    stage_dicts = OrderedDict()
    sec = "pipeline"
    pd = stage_dicts[sec] = {}
    order = [split(x, ".") for x in split(config.get(sec, "order"))]
    pd["name"] = config.get(sec, "name") if config.has_option(sec, "name") else "none"
    if config.has_option(sec, "output_binning"):
        pd["output_binning"] = binning_dict[config.get(sec, "output_binning")]
        key = split(config.get(sec, "output_key"))
        if len(key) == 1:
            pd["output_key"] = key[0]
        elif len(key) == 2:
            pd["output_key"] = tuple(key)
        else:
            raise ValueError("Output key should be exactly one key, or a tuple (key, error_key), "
                             "but is %s" % key)
    else:
        pd["output_binning"] = pd["output_format"] = pd["output_key"] = None
    param_selections = split(config.get(sec, "param_selections")) \
        if config.has_option(sec, "param_selections") else []
    pd["detector_name"] = config.get(sec, "detector_name") if config.has_option(sec, "detector_name") else None

    for stage, service in order:
        section = "%s.%s" % (stage, service)
        if not config.has_section(section):
            raise IOError('missing section in cfg for stage "%s" service "%s"' % (stage, service))
        kwargs = OrderedDict()
        selector = ParamSelector(selections=param_selections)
        kwargs["params"] = selector
        n_params = 0
        for fullname in config.options(section):
            value = config.get(section, fullname)
            m = PARAM_RE.match(fullname)
            if m is not None:
                n_params += 1
                sel, pname, attr = _param_subfields(m.group("subfields").split("."))
                if attr is not None:
                    continue
                # a param defined in an earlier stage is shared (config_parser.py:852-879)
                param = None
                for kw in stage_dicts.values():
                    if "params" not in kw:
                        continue
                    try:
                        param = kw["params"].get(name=pname, selector=sel)
                    except (KeyError, ValueError):
                        param = None
                        continue
                    for a in PARAM_ATTRS:
                        if config.has_option(section, "%s.%s" % (fullname, a)):
                            raise ValueError("Parameter spec. '%s' of '%s' found in section '%s', "
                                             "but parameter exists in previous stage!"
                                             % (a, fullname, section))
                    break
                if param is None:
                    param = parse_param(config, section, sel, fullname, pname, value)
                selector.update(param, selector=sel, extend=True)
            elif value in binning_dict:
                kwargs[fullname] = binning_dict[value]
            elif "binning" in fullname:
                kwargs[fullname] = binning_dict[value]
            elif fullname in ("calc_mode", "apply_mode", "output_format"):
                v = parse_string_literal(value)
                kwargs[fullname] = binning_dict.get(v, v) if isinstance(v, str) else v
            elif fullname.endswith("_names"):
                kwargs[fullname] = split(value)
            else:
                if re.search(r"[^a-z_]units\.[a-z]+", value, flags=re.IGNORECASE):
                    try:
                        q = parse_quantity(value)
                        kwargs[fullname] = Quantity(q.nominal_value, q.units)
                    except ValueError:
                        kwargs[fullname] = parse_string_literal(value)
                else:
                    kwargs[fullname] = parse_string_literal(value)
        if n_params == 0:
            kwargs.pop("params")
        else:
            selector.select_params(param_selections, error_on_missing=False)
        stage_dicts[(stage, service)] = kwargs
    return stage_dicts
