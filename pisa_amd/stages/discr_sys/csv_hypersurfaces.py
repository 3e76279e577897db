"""Hypersurfaces of the data releases, stored as CSV tables (counterpart of
pisa/stages/discr_sys/csv_hypersurfaces.py:37-243): one table per (linked) container with, for every value of the
interpolation parameter and every bin, an intercept, one gradient per systematic and their errors.  Per parameter
change the two table slices around the interpolation parameter are interpolated linearly, the plane
`intercept + sum_p gradient_p * (value_p - nominal_p)` is evaluated per bin (non-finite -> 1) -- a few hundred numbers,
on the host like the reference -- and per run the maps are scaled on the device: `weights = clip(weights * hs_scales,
0, inf)`, `errors = weights * hs_scales_uncertainty` (or `errors *= hs_scales`), `bin_unc2` like the weights
(`pisa_hip_bin_scale`)."""
import ast
import os
from collections.abc import Mapping

import numpy as np

from pisa_amd import FTYPE
from pisa_amd import kernels as K
from pisa_amd.core.binning import MultiDimBinning
from pisa_amd.core.stage import Stage
from pisa_amd.utils.format import split
from pisa_amd.utils.log import logging
from pisa_amd.utils.resources import find_resource

__all__ = ["csv_hypersurfaces"]


class csv_hypersurfaces(Stage):  # pylint: disable=invalid-name
    def __init__(self, fit_results_file, nominal_systematics, inter_param, links=None, propagate_uncertainty=True,
                 **std_kwargs):
        self.hs = {}
        self.fit_results_file = split(fit_results_file)
        if isinstance(nominal_systematics, str):
            self.nominal_systematics = eval(nominal_systematics)  # pylint: disable=eval-used
        elif isinstance(nominal_systematics, dict):
            self.nominal_systematics = nominal_systematics
        else:
            raise ValueError("Unsupported type %s for nominal_systematics." % type(nominal_systematics))
        self.inter_param = inter_param
        self.propagate_uncertainty = propagate_uncertainty
        keys = ["weights"]
        if std_kwargs.get("error_method"):
            keys.append("errors")
        super().__init__(expected_params=list(self.nominal_systematics.keys()) + [self.inter_param],
                         expected_container_keys=keys,
                         supported_reps={"calc_mode": MultiDimBinning, "apply_mode": [MultiDimBinning, "events"]},
                         **std_kwargs)
        if links is None:
            self.links = {}
        elif not isinstance(links, Mapping):
            self.links = ast.literal_eval(links)
        else:
            self.links = links

    def _link(self):
        for key, val in self.links.items():
            self.data.link_containers(key, val)

    def setup_function(self):
        import pandas as pd

        for f in self.fit_results_file:
            k = os.path.splitext(os.path.basename(f))[0]
            if k.startswith("hs_"):
                k = k[3:]
            if k in self.hs:
                raise ValueError("%s already exists in HS dict." % k)
            self.hs[k] = pd.read_csv(find_resource(f))
        self._link()
        for container in self.data:
            assert container.name in self.hs, "No match for %s found in the hypersurfaces." % container.name
            container["hs_scales"] = np.empty(container.size, dtype=FTYPE)
            if self.propagate_uncertainty:
                hs = self.hs[container.name]
                value = self.params[self.inter_param].m
                start = int(np.argmin(np.abs(np.asarray(hs[self.inter_param]) - value)))
                _, counts = np.unique(hs[self.inter_param], return_counts=True)
                unc = np.asarray(hs["intercept_sigma"][start:start + counts[0]], dtype=FTYPE)
                container["hs_scales_uncertainty"] = unc.reshape(container.size)
        self.data.unlink_containers()

    def get_corr_factors(self, hs, param_values):
        diffs = {k: v - self.nominal_systematics[k] for k, v in param_values.items()}
        return hs["intercept"] + sum([hs[p] * d for p, d in diffs.items()])

    def compute_function(self):
        self._link()
        param_values = {name: self.params[name].m for name in self.nominal_systematics}
        value = self.params[self.inter_param].m
        for container in self.data:
            hs = self.hs[container.name]
            column = hs[self.inter_param]
            if value < column.min() or column.max() < value:
                raise ValueError("%s of %f is outside of interpolation range." % (self.inter_param, value))
            nodes = column.unique()
            lower_val = nodes[nodes <= value].max()
            upper_val = nodes[nodes > value].min()
            hs_lower = hs.loc[column == lower_val].reset_index()
            hs_upper = hs.loc[column == upper_val].reset_index()
            hs_interpolated = hs_lower.copy()
            for p in ["intercept"] + list(param_values.keys()):
                binlen = hs_upper[self.inter_param][0] - hs_lower[self.inter_param][0]
                grad = (np.array(hs_upper[p]) - np.array(hs_lower[p])) / binlen
                hs_interpolated[p] = grad * (value - hs_lower[self.inter_param][0]) + hs_lower[p]
            scales = np.array(self.get_corr_factors(hs_interpolated, param_values), dtype=FTYPE).reshape(container.size)
            empty = ~np.isfinite(scales)
            if empty.any():
                logging.warning("%i empty bins found in hypersurface for %s", int(empty.sum()), container.name)
            scales[empty] = 1.0
            container["hs_scales"] = scales
            container.mark_valid("hs_scales")
        self.data.unlink_containers()

    def apply_function(self):
        for container in self.data:
            scales = container.device("hs_scales")
            if self.error_method == "sumw2":
                if self.data.representation == "events":
                    logging.warning("running stage in events mode. Hypersurface error propagation will be IGNORED.")
                elif self.propagate_uncertainty:
                    container["errors"] = K.bin_scale(container.device("weights"), container.device("hs_scales_uncertainty"))
                else:
                    container["errors"] = K.bin_scale(container.device("errors"), scales)
                if "bin_unc2" in container.keys:
                    container["bin_unc2"] = K.bin_scale(container.device("bin_unc2"), scales, floor=0.0)
            container["weights"] = K.bin_scale(container.device("weights"), scales, floor=0.0)


def service_test_binning():
    """the binning of the example table `events/hs_test.csv` (csv_hypersurfaces.py:260-271)"""
    from pisa_amd.core.binning import OneDimBinning
    from pisa_amd.core.units import ureg

    dd_en = OneDimBinning("reco_energy", num_bins=10, is_log=True, tex=r"E_{\\rm reco}",
                          bin_edges=[6.31, 8.46, 11.34, 15.20, 20.38, 27.31, 36.61, 49.08, 65.79, 88.20, 158.49] * ureg.GeV)
    dd_cz = OneDimBinning("reco_coszen", num_bins=10, is_lin=True, domain=[-1, 0.1], tex=r"\\cos{\\theta}_{\\rm reco}")
    dd_pid = OneDimBinning("pid", bin_edges=[0.55, 0.75, 1.0], tex=r"{\\rm PID}")
    return MultiDimBinning([dd_en, dd_cz, dd_pid], name="oscNext_verification")


def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    from pisa_amd.core.param import Param, ParamSet
    from pisa_amd.core.units import ureg

    values = [("dom_eff", 1.0), ("hole_ice_p0", 0.1), ("hole_ice_p1", -0.05), ("bulk_ice_scatter", 1.05),
              ("bulk_ice_abs", 1.0), ("dm31", 3e-3 * ureg.eV ** 2)]
    nominal = {"dom_eff": 1.00, "hole_ice_p0": 0.10, "hole_ice_p1": -0.05, "bulk_ice_abs": 1.00, "bulk_ice_scatter": 1.00}
    binning = service_test_binning()
    return csv_hypersurfaces(fit_results_file="events/hs_test.csv", nominal_systematics=nominal, inter_param="dm31",
                             links={"test": ["test1_cc", "test2_nc"]},
                             params=ParamSet([Param(name=n, value=v, **param_kwargs) for n, v in values]),
                             calc_mode=binning, apply_mode=binning)
