"""Apply a livetime / overall scale to already calculated weights (counterpart of
pisa/stages/aeff/weight.py:14-65): `weights *= weight_scale * livetime_s`, and `errors` likewise where
a container has them (:57-65) -- on events or on maps, whatever the stage's mode is.  The product runs
on the device (`pisa_hip_bin_scale`); in an event representation a pending reweighting chain is
materialised first, as for any stage that touches `weights` directly."""
from pisa_amd import kernels as K
from pisa_amd.core.stage import Stage

__all__ = ["weight"]


class weight(Stage):  # pylint: disable=invalid-name
    def __init__(self, **std_kwargs):
        super().__init__(expected_params=("livetime", "weight_scale"), expected_container_keys=("weights",),
                         **std_kwargs)

    def apply_function(self):
        scale = self.params.weight_scale.m_as("dimensionless") * self.params.livetime.m_as("sec")
        for container in self.data:
            container["weights"] = K.bin_scale(container.device("weights"), None, scale)
            if "errors" in container.keys:
                container["errors"] = K.bin_scale(container.device("errors"), None, scale)

def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    from pisa_amd.core.param import Param, ParamSet
    from pisa_amd.core.units import ureg

    return weight(params=ParamSet([Param(name="livetime", value=3 * ureg.year, **param_kwargs),
                                   Param(name="weight_scale", value=1.0, **param_kwargs)]))
