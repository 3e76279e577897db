"""Detector systematics from a SnowStorm simulation by splitting and histogramming (counterpart of
pisa/stages/cont_sys/snowstorm_hist.py:32-236).  Every event carries the value each systematic had when it was
simulated.  Per systematic the events are split at the central value of the simulated distribution and histogrammed
in the output binning (h1 above, h2 below: two passes of the weighted-histogram kernel with the weights masked on the
device); the gradient per bin is 2 (h1 - h2) c / (h1 + h2) with c = sqrt(pi/2) / sigma for a Gaussian and
2 / (max - min) for a uniform distribution (NaN -> 0), the scale per bin prod_s (1 + (value_s - central_s) grad_s)
clipped at 0, and the binned weights are multiplied by it (`pisa_hip_bin_scale`).  As in the reference the stage reads
`container["weights"]` in the event representation AFTER utils.hist wrote the binned ones, so what it splits are the
bin contents looked up at the events (the Container's translation rule).  Gradients are re-made when an
`additional_params` value moved by more than its tolerance since they were last made."""
import numpy as np

from pisa_amd import FTYPE
from pisa_amd import kernels as K
from pisa_amd.core.binning import MultiDimBinning
from pisa_amd.core.container import regularized
from pisa_amd.core.stage import Stage

__all__ = ["snowstorm_hist"]


def _listed(x, default=None):
    if isinstance(x, str):
        return eval(x)  # pylint: disable=eval-used
    return default if x is None else x


class snowstorm_hist(Stage):  # pylint: disable=invalid-name
    def __init__(self, systematics, simulation_dists, simulation_dists_params, additional_params=None, tolerances=None,
                 **std_kwargs):
        self.systematics = _listed(systematics)
        assert isinstance(self.systematics, list)
        self.simulation_dists = _listed(simulation_dists)
        assert isinstance(self.simulation_dists, list) and len(self.simulation_dists) == len(self.systematics)
        for sd in self.simulation_dists:
            assert sd.lower() in ["gauss", "uniform"]
        self.simulation_dists_params = _listed(simulation_dists_params)
        assert isinstance(self.simulation_dists_params, list)
        assert len(self.simulation_dists_params) == len(self.systematics)
        self.additional_params = _listed(additional_params, [])
        assert isinstance(self.additional_params, list)
        self.tol = _listed(tolerances, [0] * len(self.additional_params))
        assert isinstance(self.tol, list) and len(self.tol) == len(self.additional_params)
        self.tol = np.array(self.tol, dtype=FTYPE)
        self.grads = {}
        self.central_values = []
        super().__init__(expected_params=self.systematics + self.additional_params,
                         expected_container_keys=["weights"] + self.systematics,
                         supported_reps={"calc_mode": "events", "apply_mode": [None, MultiDimBinning]}, **std_kwargs)

    def setup_function(self):
        if self.apply_mode is None:
            self.apply_mode = self.data["output_binning"]
        else:
            assert self.apply_mode == self.data["output_binning"]
        self.central_values = []
        for sd, prm in zip(self.simulation_dists, self.simulation_dists_params):
            self.central_values.append(prm[0] if sd.lower() == "gauss" else sum(prm) / 2)
        self._samples = {}
        for container in self.data:
            self.grads[container.name] = {}
            container.representation = "events"
            self._reg_binning, cols = regularized(self.apply_mode, lambda n, log, c=container: (np.log(c[n]) if log else c[n]))
            self._samples[container.name] = [K.to_device(np.ascontiguousarray(col, dtype=FTYPE)) for col in cols]
        self.additional_params_values = None

    def _gradients(self, container):
        """snowstorm_hist.py:189-217 for one container, all systematics"""
        container.representation = self.calc_mode
        weights = container.device("weights")
        cols = self._samples[container.name]
        for i, sys in enumerate(self.systematics):
            value = container.device(sys)
            h1 = K.histogram_regular(cols, weights * (value > self.central_values[i]), self._reg_binning).cpu().numpy()
            h2 = K.histogram_regular(cols, weights * (value < self.central_values[i]), self._reg_binning).cpu().numpy()
            with np.errstate(divide="ignore", invalid="ignore"):
                if self.simulation_dists[i].lower() == "gauss":
                    correction_factor = 1 / self.simulation_dists_params[i][1] * np.sqrt(np.pi / 2)
                    grad = np.nan_to_num(2 * (h1 - h2) * correction_factor / (h1 + h2))
                else:
                    diff = (self.simulation_dists_params[i][1] - self.simulation_dists_params[i][0]) / 2
                    grad = np.nan_to_num(2 * (h1 - h2) / diff / (h1 + h2))
            self.grads[container.name][sys] = grad

    def compute_function(self):
        values = np.array([self.params[p].m for p in self.additional_params], dtype=FTYPE)
        first = self.data.names[0]
        if self.additional_params_values is None or np.any(np.abs(values - self.additional_params_values) > self.tol):
            calc_grads = True
            self.additional_params_values = values
        else:
            calc_grads = self.apply_mode.size != len(self.grads[first].get(self.systematics[0], ()))
        for container in self.data:
            if calc_grads:
                self._gradients(container)
            container.representation = self.apply_mode
            scale = np.ones(self.apply_mode.size)
            for i, sys in enumerate(self.systematics):
                scale *= 1 + (self.params[sys].m - self.central_values[i]) * self.grads[container.name][sys]
            container["syst_scale"] = np.clip(scale, a_min=0, a_max=np.inf)

    def apply_function(self):
        for container in self.data:
            container["weights"] = K.bin_scale(container.device("weights"), container.device("syst_scale"))


def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    from pisa_amd.core.param import Param, ParamSet
    from pisa_amd.core.units import ureg

    return snowstorm_hist(systematics=["dom_eff"], simulation_dists=["gauss"], simulation_dists_params=[(1.0, 0.1)],
                          additional_params=["deltam31"], calc_mode="events",
                          params=ParamSet([Param(name="dom_eff", value=1.0, **param_kwargs),
                                           Param(name="deltam31", value=3e-3 * ureg.eV ** 2, **param_kwargs)]))
