"""`EventsPi`: events of a PISA HDF5 file (or an equivalent nested mapping) sorted into flavour / interaction
groups with chosen, renamed variables (counterpart of pisa/core/events_pi.py:104-637, the part the
`data.simple_data_loader` service uses).  Host-side file handling; the arrays become container columns."""
import re
from collections import OrderedDict
from collections.abc import Iterable, Mapping, Sequence

import numpy as np

from pisa_amd import FTYPE

__all__ = ["EventsPi", "NU_FLAVORS", "NU_INTERACTIONS", "OUTPUT_NUFLAVINT_KEYS", "split_nu_events_by_flavor_and_interaction",
           "fix_oppo_flux"]

NU_FLAVORS = OrderedDict(nue=12, nuebar=-12, numu=14, numubar=-14, nutau=16, nutaubar=-16)
NU_INTERACTIONS = OrderedDict(cc=1, nc=2)
OUTPUT_NUFLAVINT_KEYS = tuple("%s_%s" % (f, i) for f in NU_FLAVORS for i in NU_INTERACTIONS)
# {"<flavour>[_bar]": {"cc" | "nc": ...}} of older files -> "<flavour>[bar]_<int>" (events_pi.py:61-71)
_LEGACY_FLAVOURS = {k: k.replace("_", "") for k in ("nue", "nuebar", "nue_bar", "numu", "numubar", "numu_bar",
                                                     "nutau", "nutaubar", "nutau_bar")}
# the files' "oppo" fluxes are the fluxes of the OTHER sign (events_pi.py:74-85, 725-744)
_OPPO_NU = {"nominal_nue_flux": "neutrino_nue_flux", "nominal_numu_flux": "neutrino_numu_flux",
            "nominal_nuebar_flux": "neutrino_oppo_nue_flux", "nominal_numubar_flux": "neutrino_oppo_numu_flux"}
_OPPO_NUBAR = {"nominal_nue_flux": "neutrino_oppo_nue_flux", "nominal_numu_flux": "neutrino_oppo_numu_flux",
               "nominal_nuebar_flux": "neutrino_nue_flux", "nominal_numubar_flux": "neutrino_numu_flux"}


def _append(key, val, into):
    """arrays of several files joined, through any depth of groups"""
    if isinstance(val, Mapping):
        sub = into.setdefault(key, OrderedDict())
        for k, v in val.items():
            _append(k, v, sub)
        return
    assert isinstance(val, np.ndarray), "'%s' is not an array, is a %s" % (key, type(val))
    into[key] = np.append(into[key], val) if key in into else val


def split_nu_events_by_flavor_and_interaction(input_data):
    """groups named "<flavour>_<cc|nc>": taken as they are, flattened from the legacy two-level form, or cut out of
    mixed groups by `pdg_code` and `interaction` (events_pi.py:640-722)"""
    assert isinstance(input_data, Mapping) and input_data, "`input_data` has no members"
    out = OrderedDict()

    def put(key, data):
        if key in out:
            for var, arr in data.items():
                out[key][var] = np.concatenate([out[key][var], arr])
        else:
            out[key] = OrderedDict(data)

    for key, data in input_data.items():
        if key in OUTPUT_NUFLAVINT_KEYS:
            put(key, data)
        elif key in _LEGACY_FLAVOURS:
            for sub, sub_data in data.items():
                assert sub in ("cc", "nc"), str(sub)
                put("%s_%s" % (_LEGACY_FLAVOURS[key], sub), sub_data)
        else:
            assert "pdg_code" in data, "No 'pdg_code' variable found for %s data" % key
            assert np.all(np.isin(data["pdg_code"], list(NU_FLAVORS.values()))), \
                "%s data does not appear to be a neutrino data" % key
            assert "interaction" in data, "No 'interaction' variable found for %s data" % key
            for flav, pdg in NU_FLAVORS.items():
                for inter, code in NU_INTERACTIONS.items():
                    mask = (data["pdg_code"] == pdg) & (data["interaction"] == code)
                    if np.any(mask):
                        put("%s_%s" % (flav, inter), OrderedDict((v, a[mask]) for v, a in data.items()))
    return out


def fix_oppo_flux(input_data):
    for key, val in input_data.items():
        if "neutrino_oppo_nue_flux" not in val:
            continue
        for new, old in (_OPPO_NUBAR if "bar" in key else _OPPO_NU).items():
            val[new] = val.pop(old)


class EventsPi(OrderedDict):
    """{group name: {variable: array}} plus `metadata`"""

    def __init__(self, *args, name=None, neutrinos=True, fraction_events_to_keep=None, events_subsample_index=0, **kwargs):
        super().__init__(*args, **kwargs)
        self.name = name
        self.neutrinos = neutrinos
        self.fraction_events_to_keep = fraction_events_to_keep
        self.events_subsample_index = events_subsample_index
        self.metadata = OrderedDict([("detector", ""), ("geom", ""), ("runs", []), ("proc_ver", ""), ("cuts", [])])
        if fraction_events_to_keep is not None:
            frac = float(fraction_events_to_keep)
            assert 0.0 <= frac <= 1.0, "`fraction_events_to_keep` must be in range [0.,1.], or None to disable"
            assert isinstance(events_subsample_index, int) and events_subsample_index >= 0
            assert events_subsample_index < int(np.floor(1.0 / frac)), \
                "`events_subsample_index` = %d is too large given `fraction_events_to_keep` = %g" % (events_subsample_index, frac)
            self.fraction_events_to_keep = frac

    def load_events_file(self, events_file, variable_mapping=None, required_metadata=None, seed=123456):
        """fill from HDF5 file(s) or mapping(s); `variable_mapping` = {name here: name in the file, or several names
        whose arrays become the columns of one 2-d array} (events_pi.py:175-491)"""
        from pisa_amd.utils.hdf import from_hdf

        if not isinstance(events_file, (str, Mapping, Sequence)):
            raise TypeError("`events_file` must be either string or mapping; got (%s)" % type(events_file))
        if variable_mapping is not None:
            if not isinstance(variable_mapping, Mapping):
                raise TypeError("'variable_mapping' must be a mapping (e.g., dict)")
            for dst, src in variable_mapping.items():
                if not isinstance(dst, str):
                    raise TypeError("`variable_mapping` 'dst' (key) must be a string")
                if not isinstance(src, str) and not (isinstance(src, Iterable) and all(isinstance(v, str) for v in src)):
                    raise TypeError("`variable_mapping` 'src' (value) must be a string or an iterable of strings")
        files = [events_file] if isinstance(events_file, (str, Mapping)) else list(events_file)
        input_data = OrderedDict()
        for infile in files:
            if isinstance(infile, str):
                choose = None
                if variable_mapping is not None:
                    choose = [v for src in variable_mapping.values() for v in ([src] if isinstance(src, str) else src)]
                    choose += [m[v] for v in list(choose) for m in (_OPPO_NU, _OPPO_NUBAR) if v in m]
                loaded, attrs = from_hdf(infile, choose=choose, return_attrs=True)
                assert len(loaded) > 0, "No input data found"
            else:
                loaded, attrs = infile, getattr(infile, "metadata", None) or getattr(infile, "attrs", None) or {}
            for k, v in loaded.items():
                _append(k, v, input_data)
            for k in required_metadata or ():
                assert k in attrs, "Expected metadata '%s' not found" % k
                val = attrs[k].item() if isinstance(attrs[k], np.ndarray) and attrs[k].ndim == 0 else attrs[k]
                if k in self.metadata and k == "livetime":
                    self.metadata[k] += val
                elif k in self.metadata and k not in ("detector", "geom", "runs", "proc_ver", "cuts"):
                    assert self.metadata[k] == val
                else:
                    self.metadata[k] = val
        if self.neutrinos:
            input_data = split_nu_events_by_flavor_and_interaction(input_data)
            fix_oppo_flux(input_data)
        for group, arrays in input_data.items():
            if group in self:
                raise ValueError("Key '%s' has already been added to this data structure" % group)
            if not isinstance(arrays, Mapping):
                raise Exception("'%s' input data is not a mapping, unknown format (%s)" % (group, type(arrays)))
            self[group] = OrderedDict()
            mapping = variable_mapping.items() if variable_mapping is not None else [(k, k) for k in arrays]
            chosen = None
            rand = np.random.RandomState(seed)       # the same sub-sample every time
            for dst, src in mapping:
                cols = []
                for var in ([src] if isinstance(src, str) else src):
                    if var not in arrays:
                        raise KeyError("Variable '%s' cannot be found for '%s' events" % (var, group))
                    cols.append(np.asarray(arrays[var]).astype(FTYPE))
                data = np.squeeze(np.stack(cols, axis=1))
                if self.fraction_events_to_keep is not None:
                    if chosen is None:
                        # statistically independent sub-samples: draw one, remove it, draw the next ... (events_pi.py:468-487)
                        n0 = data.size
                        want = int(self.fraction_events_to_keep * float(n0))
                        current = np.arange(n0)
                        i = 0
                        while True:
                            assert current.size >= want, "Not enough events available"
                            chosen = np.sort(rand.choice(current, replace=False, size=want))
                            if i == self.events_subsample_index:
                                break
                            current = np.sort(np.setxor1d(current, chosen))
                            i += 1
                    data = data[chosen]
                self[group][dst] = np.ascontiguousarray(data)

    def apply_cut(self, keep_criteria):
        """a new EventsPi with the events of every group that satisfy the numpy expression `keep_criteria`; all
        groups or none (events_pi.py:493-569)"""
        assert isinstance(keep_criteria, str)
        if keep_criteria in self.metadata["cuts"]:
            return self
        cut = EventsPi(name=self.name, neutrinos=self.neutrinos)
        cut.metadata = OrderedDict((k, list(v) if isinstance(v, list) else v) for k, v in self.metadata.items())
        for group, arrays in self.items():
            expr = keep_criteria
            for var in sorted(arrays, key=len, reverse=True):
                expr = re.sub(r"\b%s\b" % re.escape(var), 'arrays["%s"]' % var, expr)
            mask = eval(expr, {"np": np, "numpy": np, "arrays": arrays})  # pylint: disable=eval-used
            cut[group] = OrderedDict((v, a[mask]) for v, a in arrays.items())
        cut.metadata["cuts"].append(keep_criteria)
        return cut

    def keep_inbounds(self, binning):
        """events inside the limits of `binning`'s dimensions that they have"""
        from pisa_amd.core.binning import MultiDimBinning, OneDimBinning

        if isinstance(binning, OneDimBinning):
            binning = [binning]
        binning = MultiDimBinning(binning)
        out = self
        for dim in binning:
            if all(dim.name in arrays for arrays in self.values()):
                out = out.apply_cut(dim.inbounds_criteria)
        return out
