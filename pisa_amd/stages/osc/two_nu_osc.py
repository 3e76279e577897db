"""Two-flavour vacuum oscillation weights (counterpart of pisa/stages/osc/two_nu_osc.py:18-127): containers whose
name holds 'numu' get `weights *= nu_flux[:, 1] * (1 - P)`, 'nutau' `weights *= nu_flux[:, 1] * P`, 'nue'
`weights *= nu_flux[:, 0]`, with P = theta23 * sin^2(1.267 dm31 L / E) and L the path from a production height of
19 km through an Earth of 6378.2 km (:101-110).  As in the reference `theta23` enters by its magnitude in radians
(`m_as('dimensionless')`), not through sin^2(2 theta).  One launch of `pisa_hip_two_nu_osc` per container."""
from pisa_amd import kernels as K
from pisa_amd.core.stage import Stage

__all__ = ["two_nu_osc"]


class two_nu_osc(Stage):  # pylint: disable=invalid-name
    def __init__(self, **std_kwargs):
        super().__init__(expected_params=("theta23", "deltam31"),
                         expected_container_keys=("true_energy", "true_coszen", "nu_flux", "weights"), **std_kwargs)

    def apply_function(self):
        theta = self.params.theta23.value.m_as("dimensionless")
        deltam31 = self.params.deltam31.value.m_as("eV**2")
        for container in self.data:
            # two_nu_osc.py:70-97: three independent tests on the name, in this order
            for tag, flav in (("numu", 1), ("nutau", 2), ("nue", 0)):
                if tag in container.name:
                    weights = container.device("weights").clone()
                    K.two_nu_osc(container.device("nu_flux"), theta, deltam31, container.device("true_energy"),
                                 container.device("true_coszen"), flav, weights)
                    container["weights"] = weights


def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    from pisa_amd.core.param import Param, ParamSet
    from pisa_amd.core.units import ureg

    return two_nu_osc(params=ParamSet([Param(name="theta23", value=45 * ureg.degree, **param_kwargs),
                                       Param(name="deltam31", value=2.5e-3 * ureg.eV ** 2, **param_kwargs)]))
