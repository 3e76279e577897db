"""Toy event generator (counterpart of pisa/stages/data/toy_event_generator.py:19-104).

Creates one container per `output_names` entry, either on the calc grid or with
`n_events` random events drawn exactly like the reference (one
`RandomState(seed)`: `true_energy = 10**(rand*3)`, then `true_coszen = rand*2-1`
per container, :56-76).  `apply_function` resets `weights` to `initial_weights`
every evaluation (:101-104) -- here as a deferred operation so that the fused
kernel can read `initial_weights` directly.
"""
import numpy as np

from pisa_amd import FTYPE
from pisa_amd.core.binning import MultiDimBinning
from pisa_amd.core.container import Container
from pisa_amd.core.stage import Stage
from pisa_amd.stages import deferred

__all__ = ["toy_event_generator"]


class toy_event_generator(Stage):  # pylint: disable=invalid-name
    def __init__(self, output_names, **std_kwargs):
        self.output_names = output_names
        super().__init__(expected_params=("n_events", "random", "seed"),
                         expected_container_keys=(), **std_kwargs)

    def setup_function(self):
        n_events = int(self.params.n_events.value.m)
        seed = int(self.params.seed.value.m)
        self.random_state = np.random.RandomState(seed)
        for name in self.output_names:
            container = Container(name, representation=self.calc_mode)
            nubar = -1 if "bar" in name else 1
            if "e" in name:
                flav = 0
            if "mu" in name:
                flav = 1
            if "tau" in name:
                flav = 2
            if not isinstance(self.calc_mode, MultiDimBinning):
                container["true_energy"] = np.power(10, self.random_state.rand(n_events).astype(FTYPE) * 3)
                container["true_coszen"] = self.random_state.rand(n_events).astype(FTYPE) * 2 - 1
            size = container.size
            if self.params.random.value:
                container["initial_weights"] = self.random_state.rand(size).astype(FTYPE)
            else:
                container["initial_weights"] = np.ones(size, dtype=FTYPE)
            container.set_aux_data("nubar", nubar)
            container.set_aux_data("flav", flav)
            container["weights"] = np.ones(size, dtype=FTYPE)
            container["weighted_aeff"] = np.ones(size, dtype=FTYPE)
            flux = np.stack([np.zeros(size, dtype=FTYPE), np.ones(size, dtype=FTYPE)], axis=1)
            container["nu_flux_nominal"] = flux
            container["nubar_flux_nominal"] = flux.copy()
            self.data.add_container(container)

    def apply_function(self):
        for container in self.data:
            deferred.reset_weights(container)

def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    from pisa_amd.core.param import Param, ParamSet

    return toy_event_generator(output_names=["numu", "nue_bar"], params=ParamSet([
        Param(name="n_events", value=100, **param_kwargs),
        Param(name="random", value=1, **param_kwargs),
        Param(name="seed", value=666, **param_kwargs)]))
