"""Atmospheric-muon background systematics (counterpart of pisa/stages/background/atm_muons.py:19-184): the
primary-cosmic-ray uncertainty tabulated against cos(zenith) is interpolated at the events once
(`rw_array`; `cr_rw_array` = that minus its mean), and per run
`weights *= clip((1 + delta_gamma_mu * cr_rw_array) * atm_muon_scale, 0, inf)`.  The interpolation is scipy's
`interp1d(kind='linear')` = `numpy.interp` on the device (`pisa_hip_interp_linear`; a coszen outside [0, 1] raises,
as interp1d's bounds_error does), the update `pisa_hip_poly_scale`.  Other spline kinds are not built."""
import numpy as np

from pisa_amd import FTYPE
from pisa_amd import kernels as K
from pisa_amd.core.stage import Stage
from pisa_amd.utils.resources import find_resource

__all__ = ["atm_muons"]


class atm_muons(Stage):  # pylint: disable=invalid-name
    def __init__(self, input_names, **std_kwargs):  # pylint: disable=unused-argument
        super().__init__(expected_params=("atm_muon_scale", "delta_gamma_mu_file", "delta_gamma_mu_spline_kind",
                                          "delta_gamma_mu_variable", "delta_gamma_mu"),
                         expected_container_keys=("true_coszen", "weights"), **std_kwargs)

    def setup_function(self):
        xvals, yvals = self._make_prim_unc_spline()
        xk, yk = K.to_device(xvals), K.to_device(yvals)
        rw_variable = self.params["delta_gamma_mu_variable"].value
        for container in self.data:
            rw = K.interp_linear(xk, yk, container.device(rw_variable))
            container["rw_array"] = rw
            host = container["rw_array"]
            norm = host.sum() / host.size
            container["cr_rw_array"] = rw - norm

    def apply_function(self):
        atm_muon_scale = self.params["atm_muon_scale"].value.m_as("dimensionless")
        cr_rw_scale = self.params["delta_gamma_mu"].value.m_as("dimensionless")
        for container in self.data:
            weights = container.device("weights").clone()
            K.poly_scale([container.device("cr_rw_array")], None, [cr_rw_scale], weights, scale=atm_muon_scale)
            container["weights"] = weights

    def _make_prim_unc_spline(self):
        """the knots of the interpolant of atm_muons.py:103-166: zeros in the table take their right neighbour's value,
        the ends are continued flat to coszen 0 and 1"""
        variable = self.params["delta_gamma_mu_variable"].value
        bare_variable = variable.split("true_")[-1]
        if not bare_variable == "coszen":
            raise ValueError("Muon primary cosmic ray systematic is currently only implemented as a function of"
                             " cos(zenith). %s was set in the configuration file." % variable)
        fname = self.params["delta_gamma_mu_file"].value
        if bare_variable not in fname:
            raise ValueError("Variable set in configuration file is %s but the file you have selected, %s, does not make"
                             " reference to this in its name." % (variable, fname))
        kind = self.params["delta_gamma_mu_spline_kind"].value
        if kind != "linear":
            raise NotImplementedError("delta_gamma_mu_spline_kind = '%s': only 'linear' is built on the device" % kind)
        with open(find_resource(fname)) as fh:
            uncdata = np.genfromtxt(fh).T
        while 0.0 in uncdata[1]:
            for zero_index in np.where(uncdata[1] == 0)[0]:
                uncdata[1][zero_index] = uncdata[1][zero_index + 1]
        xvals = np.append(np.insert(uncdata[0], 0, 0.0), 1.0)
        yvals = np.append(np.insert(uncdata[1], 0, uncdata[1][0]), uncdata[1][-1])
        return np.ascontiguousarray(xvals, dtype=FTYPE), np.ascontiguousarray(yvals, dtype=FTYPE)


def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    from pisa_amd.core.param import Param, ParamSet

    values = [("atm_muon_scale", 1.0), ("delta_gamma_mu_file", "background/muongun_primary_cr_uncertainties_coszenith.txt"),
              ("delta_gamma_mu_spline_kind", "linear"), ("delta_gamma_mu_variable", "true_coszen"), ("delta_gamma_mu", 1.0)]
    return atm_muons(input_names="muon", params=ParamSet([Param(name=n, value=v, **param_kwargs) for n, v in values]))
