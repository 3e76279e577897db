"""Effective areas from parameterisation functions (counterpart of pisa/stages/aeff/param.py:26-193): per container
name a function of the true energy and one of the true coszen -- a Python callable, a string that evaluates to one
(`lambda E: ...`, `np.poly1d([...])`) or a table to interpolate linearly (0 outside) -- and
`weights *= aeff_scale * livetime_s * f_E(true_energy) * f_cz(true_coszen)`.  The functions are the user's Python
and depend on no parameter: they are evaluated once per event set on the host (the reference does it at every
run) and kept as columns; the per-run product runs on the device in the reference's order of multiplications
(`pisa_hip_bin_scale`)."""
from collections.abc import Mapping

import numpy as np  # noqa: F401  (the parameterisation strings refer to `np`)
from scipy.interpolate import interp1d

from pisa_amd import FTYPE
from pisa_amd import kernels as K
from pisa_amd.core.stage import Stage
from pisa_amd.utils.fileio import from_file

__all__ = ["load_aeff_param", "param"]


def load_aeff_param(source):
    """dict container name -> callable, from a file name or a mapping (param.py:26-114)"""
    if not isinstance(source, (str, Mapping)):
        raise TypeError("`source` must be string or mapping")
    aeff_dict = from_file(source) if isinstance(source, str) else dict(source)
    out = {}
    for k, func in aeff_dict.items():
        if isinstance(func, str):
            param_func = eval(func)  # pylint: disable=eval-used
        elif callable(func):
            param_func = func
        elif isinstance(func, Mapping):
            is_energy, is_coszen = "energy" in func, "coszen" in func
            if "aeff" not in func:
                raise ValueError("No effective area values are provided for %s" % k)
            if not (is_energy or is_coszen):
                raise ValueError("No energy or coszen values are provided for %s" % k)
            param_func = interp1d(func["energy" if is_energy else "coszen"], func["aeff"], kind="linear",
                                  bounds_error=False, fill_value=0)
        else:
            raise TypeError("Expected parameteriation to be either a string that can be interpreted by eval or as a"
                            ' mapping of values from which to construct a spline. Got "%s".' % type(func))
        out[k] = param_func
    return out


class param(Stage):  # pylint: disable=invalid-name
    def __init__(self, **std_kwargs):
        super().__init__(expected_params=("aeff_energy_paramfile", "aeff_coszen_paramfile", "livetime", "aeff_scale"),
                         expected_container_keys=("true_energy", "true_coszen", "weights"), **std_kwargs)
        self.energy_param = load_aeff_param(self.params.aeff_energy_paramfile.value)
        self.coszen_param = load_aeff_param(self.params.aeff_coszen_paramfile.value)
        self._factors = {}

    def _static_factors(self, container):
        """(f_E(true_energy), f_cz(true_coszen)) as device columns, made once per version of the coordinates"""
        key = (container.name, container.version("true_energy"), container.version("true_coszen"), container.representation)
        hit = self._factors.get(container.name)
        if hit is None or hit[0] != key:
            cols = []
            for funcs, var in ((self.energy_param, "true_energy"), (self.coszen_param, "true_coszen")):
                if container.name in funcs:
                    cols.append(K.to_device(np.ascontiguousarray(funcs[container.name](np.asarray(container[var])), dtype=FTYPE)))
                else:
                    cols.append(None)
            hit = self._factors[container.name] = (key, cols)
        return hit[1]

    def apply_function(self):
        scale = self.params.aeff_scale.m_as("dimensionless") * self.params.livetime.m_as("sec")
        for container in self.data:
            f_e, f_cz = self._static_factors(container)
            weights = container.device("weights")
            if f_e is None and f_cz is None:
                container["weights"] = K.bin_scale(weights, None, scale)
                continue
            # param.py:170-180: scale = s * ones; scale *= f_E; scale *= f_cz; weights *= scale
            total = K.bin_scale(f_e if f_e is not None else f_cz, None, scale)
            if f_e is not None and f_cz is not None:
                total = K.bin_scale(total, f_cz)
            container["weights"] = K.bin_scale(weights, total)


def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    from pisa_amd.core.param import Param, ParamSet
    from pisa_amd.core.units import ureg

    return param(params=ParamSet([Param(name="aeff_energy_paramfile", value="aeff/vlvnt_aeff_energy_param.json", **param_kwargs),
                                  Param(name="aeff_coszen_paramfile", value="aeff/vlvnt_aeff_coszen_param.json", **param_kwargs),
                                  Param(name="livetime", value=10 * ureg.s, **param_kwargs),
                                  Param(name="aeff_scale", value=1.0, **param_kwargs)]))
