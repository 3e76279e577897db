"""Pre-calculated GENIE cross-section systematics (counterpart of pisa/stages/xsec/genie_sys.py:15-113): per
interaction k the events carry a linear and a quadratic coefficient, and
`weights *= max(0, prod_k (1 + (linear_k + quad_k p_k) p_k))` with p_k the interaction's parameter --
`pisa_hip_poly_scale`, one launch per container."""
import re

from pisa_amd import kernels as K
from pisa_amd.core.stage import Stage
from pisa_amd.utils.log import logging

__all__ = ["genie_sys"]


class genie_sys(Stage):  # pylint: disable=invalid-name
    def __init__(self, interactions="Genie_Ma_QE, Genie_Ma_RES", names="maccqe, maccres", **std_kwargs):
        interactions = re.split(r"\W+", interactions)
        names = re.split(r"\W+", names)
        assert len(interactions) == len(names), "Specify a name for each interaction"
        self.interactions = interactions
        self.names = names
        keys = ["linear_fit_" + n for n in names] + ["quad_fit_" + n for n in names] + ["weights"]
        super().__init__(expected_params=tuple(interactions), expected_container_keys=keys, **std_kwargs)

    def setup_function(self):
        for name in self.interactions:
            rng = self.params[name].range
            if rng is not None and (rng[0] < -2.0 or rng[1] > 2.0):
                logging.warning(name + " parameter bounds have been set larger than the range used to produce"
                                " interpolation points ([-2.,2]). This will void the warranty...")

    def apply_function(self):
        values = [self.params[name].m_as("dimensionless") for name in self.interactions]
        for container in self.data:
            weights = container.device("weights").clone()
            K.poly_scale([container.device("linear_fit_" + n) for n in self.names],
                         [container.device("quad_fit_" + n) for n in self.names], values, weights)
            container["weights"] = weights


def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    from pisa_amd.core.param import Param, ParamSet

    param_kwargs.pop("range", None)
    return genie_sys(params=ParamSet([Param(name="Genie_Ma_QE", value=0.0, range=[-1.0, 1.0], **param_kwargs),
                                      Param(name="Genie_Ma_RES", value=0.0, range=[-1.0, 1.0], **param_kwargs)]))
