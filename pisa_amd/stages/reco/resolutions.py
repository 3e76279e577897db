"""Improved reconstruction resolutions (counterpart of pisa/stages/reco/resolutions.py:14-96), once at setup: the
reconstructed energy and coszen move the fraction `*_improvement` of the way to the truth (coszen clipped to
[-1, 1]); `pid` moves by `pid_improvement` towards 1 for numu(bar)_cc and towards 0 for all others -- as a shift, or
with `relative_pid` as that fraction of the distance.  `pisa_hip_shift_toward` per column."""
from pisa_amd import kernels as K
from pisa_amd.core.stage import Stage

__all__ = ["resolutions"]


class resolutions(Stage):  # pylint: disable=invalid-name
    def __init__(self, relative_pid=False, **std_kwargs):
        super().__init__(expected_params=("energy_improvement", "coszen_improvement", "pid_improvement"),
                         expected_container_keys=("true_energy", "true_coszen", "reco_energy", "reco_coszen", "pid"),
                         supported_reps={"calc_mode": "events"}, **std_kwargs)
        self.relative_pid = relative_pid

    def setup_function(self):
        e_imp = self.params.energy_improvement.m_as("dimensionless")
        cz_imp = self.params.coszen_improvement.m_as("dimensionless")
        pid_imp = self.params.pid_improvement.m_as("dimensionless")
        for container in self.data:
            container["reco_energy"] = K.shift_toward(container.device("reco_energy"), container.device("true_energy"),
                                                      e_imp)
            container["reco_coszen"] = K.shift_toward(container.device("reco_coszen"), container.device("true_coszen"),
                                                      cz_imp, clip=(-1.0, 1.0))
            track = container.name in ("numu_cc", "numubar_cc")
            pid = container.device("pid")
            if self.relative_pid:
                container["pid"] = K.shift_toward(pid, 1.0 if track else 0.0, pid_imp)
            else:
                # pid +- improvement = pid + ((pid +- 1) - pid) * improvement up to the rounding of (pid +- 1) - pid;
                # the sum itself is one rounding: x + (t - x) * f with t = improvement, x -> 0 is not that -- add directly
                container["pid"] = pid + (pid_imp if track else -pid_imp)


def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    from pisa_amd.core.param import Param, ParamSet

    return resolutions(params=ParamSet([Param(name="energy_improvement", value=0.9, **param_kwargs),
                                        Param(name="coszen_improvement", value=0.5, **param_kwargs),
                                        Param(name="pid_improvement", value=0.02, **param_kwargs)]))
