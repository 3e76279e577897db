"""Replace the errors by a manually chosen variance (counterpart of
pisa/stages/utils/set_variance.py:23-104): variance = weights * variance_scale
[* expected_total_mc / n_mc_events] [floored], errors = sqrt(variance)."""
import numpy as np

from pisa_amd import FTYPE
from pisa_amd import kernels as K
from pisa_amd.core.binning import MultiDimBinning
from pisa_amd.core.stage import Stage

__all__ = ["set_variance"]


class set_variance(Stage):  # pylint: disable=invalid-name
    def __init__(self, variance_scale=1.0, variance_floor=None, expected_total_mc=None,
                 divide_total_mc=False, **std_kwargs):
        super().__init__(expected_params=(), expected_container_keys=(),
                         supported_reps={"calc_mode": MultiDimBinning, "apply_mode": MultiDimBinning},
                         **std_kwargs)
        self.variance_scale = variance_scale
        self.variance_floor = variance_floor
        assert self.variance_scale is not None
        assert expected_total_mc is not None
        self.expected_total_mc = int(expected_total_mc)
        self.divide_n = divide_total_mc
        self.total_mc = {}

    def setup_function(self):
        if self.divide_n:
            self.data.representation = "events"
            for container in self.data:
                self.total_mc[container.name] = container.size
        self.data.representation = self.calc_mode
        for container in self.data:
            container["manual_variance"] = np.empty(container.size, dtype=FTYPE)
            if "errors" not in container.keys:
                container["errors"] = np.empty(container.size, dtype=FTYPE)

    def compute_function(self):
        for container in self.data:
            scalar = float(self.variance_scale)
            var = K.bin_scale(container.device("weights"), None, scalar)
            if self.divide_n:
                var = K.bin_scale(var, None, self.expected_total_mc / self.total_mc[container.name])
            if self.variance_floor is not None:
                var = K.bin_scale(var, None, 1.0, floor=float(self.variance_floor))
            container["manual_variance"] = var
            container.mark_valid("manual_variance")

    def apply_function(self):
        for container in self.data:
            container["errors"] = K.bin_sqrt(container.device("manual_variance"))

def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    from pisa_amd.stages.utils.kde import service_test_binning

    b = service_test_binning()
    return set_variance(expected_total_mc=100, calc_mode=b, apply_mode=b)
