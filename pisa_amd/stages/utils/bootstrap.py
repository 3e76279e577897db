"""Bootstrap samples of the events (counterpart of pisa/stages/utils/bootstrap.py:38-151): every container draws
`size` indices with replacement from numpy's `default_rng(seed)` -- ONE generator over the containers in order, as in
the reference, so a seed gives the same sample -- and multiplies the weights by how often each event was drawn."""
from collections import OrderedDict
from copy import deepcopy

import numpy as np

from pisa_amd import FTYPE
from pisa_amd import kernels as K
from pisa_amd.core.stage import Stage

__all__ = ["bootstrap", "insert_bootstrap_after_data_loader"]


class bootstrap(Stage):  # pylint: disable=invalid-name
    def __init__(self, seed=None, **std_kwargs):
        super().__init__(expected_params=(), expected_container_keys=("weights",),
                         supported_reps={"calc_mode": "events"}, **std_kwargs)
        assert self.calc_mode == "events"
        self.seed = None if seed is None else int(seed)

    def setup_function(self):
        rng = np.random.default_rng(self.seed)
        for container in self.data:
            size = container.size
            drawn = rng.integers(size, size=size)
            container["bootstrap_weights"] = np.bincount(drawn, minlength=size).astype(FTYPE)

    def apply_function(self):
        for container in self.data:
            container["weights"] = K.bin_scale(container.device("weights"), container.device("bootstrap_weights"))


def insert_bootstrap_after_data_loader(cfg_dict, seed=None):
    """a copy of a parsed pipeline configuration with this stage placed right after `data.simple_data_loader`
    (bootstrap.py:109-146)"""
    stage_cfg = OrderedDict([("apply_mode", "events"), ("calc_mode", "events"), ("seed", seed)])
    out = OrderedDict()
    for k, v in deepcopy(cfg_dict).items():
        out[k] = v
        if k == ("data", "simple_data_loader"):
            out[("utils", "bootstrap")] = stage_cfg
    return out


def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    return bootstrap(calc_mode="events")
