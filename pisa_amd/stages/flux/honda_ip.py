"""Nominal atmospheric flux from a Honda table (counterpart of
pisa/stages/flux/honda_ip.py:22-107).

`compute_function` evaluates the integral-preserving interpolation of
pisa/utils/flux_weights.py:267-349 for the four primaries of every container on the
device (`pisa_hip_flux_2d`): one launch per container instead of the reference's
Python loop over events with four spline fits per event.
"""
import numpy as np

from pisa_amd import FTYPE
from pisa_amd.core.stage import Stage
from pisa_amd.utils.flux_weights import calculate_2d_flux_weights, load_2d_table

__all__ = ["honda_ip"]

_ALL = ["nue_cc", "numu_cc", "nutau_cc", "nue_nc", "numu_nc", "nutau_nc",
        "nuebar_cc", "numubar_cc", "nutaubar_cc", "nuebar_nc", "numubar_nc", "nutaubar_nc"]


class honda_ip(Stage):  # pylint: disable=invalid-name
    def __init__(self, **std_kwargs):
        super().__init__(expected_params=("flux_table",),
                         expected_container_keys=("true_energy", "true_coszen"), **std_kwargs)
        self.flux_table = None

    def _link(self):
        # on a grid all containers share the nodes: compute once (honda_ip.py:63-67, 80-84)
        if self.data.is_map:
            names = [n for n in _ALL if n in self.data.names]
            if len(names) > 1:
                self.data.link_containers("nu", names)

    def setup_function(self):
        self.flux_table = load_2d_table(self.params.flux_table.value)
        self._link()
        for container in self.data:
            container["nu_flux_nominal"] = np.empty((container.size, 2), dtype=FTYPE)
            container["nubar_flux_nominal"] = np.empty((container.size, 2), dtype=FTYPE)
        self.data.unlink_containers()

    def compute_function(self):
        self._link()
        for container in self.data:
            nu, nubar = calculate_2d_flux_weights(container.device("true_energy"),
                                                  container.device("true_coszen"), self.flux_table)
            container["nu_flux_nominal"] = nu
            container["nubar_flux_nominal"] = nubar
            container.mark_valid("nu_flux_nominal")
            container.mark_valid("nubar_flux_nominal")
        self.data.unlink_containers()

def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    from pisa_amd.core.param import Param, ParamSet

    return honda_ip(params=ParamSet([Param(name="flux_table", value="flux/honda-2015-spl-solmin-aa.d", **param_kwargs)]))
