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

    def scale_for(self, name, values=None):
        """aeff.py:78-86; `values` = the five magnitudes read once (`_magnitudes`) when many containers are asked for"""
        a, lt, tcc, tau, nc = values or self._magnitudes()
        scale = a * lt
        if name in ("nutau_cc", "nutaubar_cc"):
            scale *= tcc
        if "nutau" in name:
            scale *= tau
        if "nc" in name:
            scale *= nc
        return scale

    def _magnitudes(self):
        p = self.params
        return (p.aeff_scale.m_in("dimensionless"), p.livetime.m_in("sec"), p.nutau_cc_norm.m_in("dimensionless"),
                p.nutau_norm.m_in("dimensionless"), p.nu_nc_norm.m_in("dimensionless"))

    def scales_for(self, names):
        """`scale_for` of several containers, the parameters looked up once"""
        v = self._magnitudes()
        return [self.scale_for(n, v) for n in names]

    def _scales(self):
        """`scale_for` of every container, recomputed when one of this stage's parameters moved"""
        from pisa_amd.core.param import ParamSet

        # (the structural counter: after `select_params` another parameter OBJECT may stand behind a name, its change
        # counter by chance the same)
        key = tuple(p._ver for p in self.params) + (len(self.data.containers), ParamSet.struct_clock)
        c = getattr(self, "_scale_cache", None)
        if c is None or c[0] != key or c[2] is not self.params:
            names = [cont.name for cont in self.data]
            c = self._scale_cache = (key, dict(zip(names, self.scales_for(names))), self.params)
        return c[1]

    def apply_function(self):
        scales = self._scales()
        for container in self.data:
            scale = scales[container.name]
            if not container.is_map or deferred.chain_open(container):
                deferred.aeff(container, scale)
            else:
                w = container.device("weights")
                K.apply_aeff(container.device("weighted_aeff"), scale, w)
                container["weights"] = w

def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    from pisa_amd.core.param import Param, ParamSet
    from pisa_amd.core.units import ureg

    return aeff(params=ParamSet([
        Param(name="livetime", value=10 * ureg.s, **param_kwargs),
        Param(name="aeff_scale", value=1.0, **param_kwargs),
        Param(name="nutau_cc_norm", value=1.0, **param_kwargs),
        Param(name="nutau_norm", value=1.0, **param_kwargs),
        Param(name="nu_nc_norm", value=1.0, **param_kwargs)]))
