"""Re-weight events by factors from an external data/MC comparison (counterpart of
pisa/stages/utils/adhoc_sys.py:23-97): the JSON holds, per variable, a binning and one factor per bin; every event
is multiplied by the factor of the bin its `variable_name` falls into (the Container's own map -> events lookup:
0 outside the binning)."""
import numpy as np

from pisa_amd import FTYPE
from pisa_amd import kernels as K
from pisa_amd.core.binning import MultiDimBinning
from pisa_amd.core.stage import Stage
from pisa_amd.utils.jsons import from_json
from pisa_amd.utils.resources import find_resource

__all__ = ["adhoc_sys"]


class adhoc_sys(Stage):  # pylint: disable=invalid-name
    def __init__(self, variable_name=None, scale_file=None, **std_kwargs):
        super().__init__(expected_params=(), expected_container_keys=["weights", variable_name],
                         supported_reps={"calc_mode": "events", "apply_mode": "events"}, **std_kwargs)
        assert self.calc_mode == "events"
        assert self.apply_mode == "events"
        self.scale_file = scale_file
        self.variable = variable_name

    def setup_function(self):
        scaling_dict = from_json(find_resource(self.scale_file))
        scale_binning = MultiDimBinning(**scaling_dict[self.variable]["binning"])
        scale_factors = np.array(scaling_dict[self.variable]["scales"], dtype=FTYPE)
        self.data.representation = scale_binning
        for container in self.data:
            container["adhoc_scale_factors"] = scale_factors

    def apply_function(self):
        for container in self.data:
            container["weights"] = K.bin_scale(container.device("weights"), container.device("adhoc_scale_factors"))


def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    import os
    import tempfile

    from pisa_amd.core.binning import OneDimBinning
    from pisa_amd.utils.jsons import to_json

    var = "reco_length"
    bin_edges = [0, 0.5, 1.0]
    var_binning = MultiDimBinning(name="adhoc_sys_test_binning",
                                  dimensions=[OneDimBinning(name=var, bin_edges=bin_edges, is_lin=True)])
    scales = np.random.RandomState(0).random_sample(len(bin_edges) - 1).astype(FTYPE)
    path = os.path.join(tempfile.gettempdir(), "pisa_amd_test_scale_file_%d.json" % os.getpid())
    to_json({var: {"binning": var_binning, "scales": scales}}, path)
    return adhoc_sys(variable_name=var, scale_file=path, calc_mode="events", apply_mode="events")
