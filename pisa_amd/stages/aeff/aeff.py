"""Effective-area weighting (counterpart of pisa/stages/aeff/aeff.py:20-101):
`weights *= weighted_aeff * aeff_scale*livetime_s*[norms by container name]`
(:78-88).  In an event representation the multiplication is recorded as a
deferred operation (fused by `utils.hist`); on maps it is applied at once."""
from pisa_amd import kernels as K
from pisa_amd.core.stage import Stage
from pisa_amd.stages import deferred

__all__ = ["aeff"]


class aeff(Stage):  # pylint: disable=invalid-name
    def __init__(self, **std_kwargs):
        expected_params = ("livetime", "aeff_scale", "nutau_cc_norm", "nutau_norm", "nu_nc_norm")
        expected_container_keys = ("weights", "weighted_aeff")
        super().__init__(expected_params=expected_params,
                         expected_container_keys=expected_container_keys, **std_kwargs)

    def scale_for(self, name):
        p = self.params
        scale = p.aeff_scale.m_in("dimensionless") * p.livetime.m_in("sec")
        if name in ("nutau_cc", "nutaubar_cc"):
            scale *= p.nutau_cc_norm.m_in("dimensionless")
        if "nutau" in name:
            scale *= p.nutau_norm.m_in("dimensionless")
        if "nc" in name:
            scale *= p.nu_nc_norm.m_in("dimensionless")
        return scale

    def apply_function(self):
        for container in self.data:
            scale = self.scale_for(container.name)
            if not container.is_map:
                deferred.aeff(container, scale)
            else:
                w = container.device("weights")
                K.apply_aeff(container.device("weighted_aeff"), scale, w)
                container["weights"] = w
