"""Power-law astrophysical flux (counterpart of pisa/stages/flux/astrophysical.py:16-149):
`astro_flux_nominal = 0.787e-18 (E / 100 TeV)^-2.5` at setup, `astro_flux = astro_norm * astro_flux_nominal *
(E / 100 TeV)^astro_delta` per parameter change, `astro_weights = initial_weights * astro_flux` per run -- all three
on the device (`pisa_hip_power_law`, `pisa_hip_bin_scale`)."""
import numpy as np

from pisa_amd import FTYPE
from pisa_amd import kernels as K
from pisa_amd.core.stage import Stage

__all__ = ["astrophysical", "PIVOT"]

PIVOT = FTYPE(100.0e3)


class astrophysical(Stage):  # pylint: disable=invalid-name
    def __init__(self, **std_kwargs):
        self._central_gamma = FTYPE(-2.5)
        self._central_norm = FTYPE(0.787e-18)
        super().__init__(expected_params=("astro_delta", "astro_norm"),
                         expected_container_keys=("true_energy", "true_coszen", "initial_weights"), **std_kwargs)

    def setup_function(self):
        for container in self.data:
            container["astro_weights"] = np.ones(container.size, dtype=FTYPE)
            container["astro_flux"] = np.ones(container.size, dtype=FTYPE)
            container["astro_flux_nominal"] = K.power_law(container.device("true_energy"), PIVOT, self._central_gamma,
                                                          self._central_norm)

    def compute_function(self):
        delta = self.params.astro_delta.value.m_as("dimensionless")
        norm = self.params.astro_norm.value.m_as("dimensionless")
        for container in self.data:
            container["astro_flux"] = K.power_law(container.device("true_energy"), PIVOT, delta, norm,
                                                  nominal=container.device("astro_flux_nominal"))
            container.mark_valid("astro_flux")

    def apply_function(self):
        for container in self.data:
            container["astro_weights"] = K.bin_scale(container.device("initial_weights"), container.device("astro_flux"))


def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    from pisa_amd.core.param import Param, ParamSet

    return astrophysical(params=ParamSet([Param(name="astro_norm", value=1.0, **param_kwargs),
                                          Param(name="astro_delta", value=0.0, **param_kwargs)]))
