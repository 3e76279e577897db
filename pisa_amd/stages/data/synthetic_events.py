"""Synthetic MC events with reco variables (builder-defined loader; the
reference's `toy_event_generator` has no reco variables and its file loaders
need h5py/pandas).  Events are drawn by `pisa_amd.synthetic.make_events`
(toy_event_generator-style true_energy / true_coszen from one RandomState(seed),
then smeared reco variables, pid, power-law fluxes, E-dependent weighted_aeff).
Container keys match what `simple_data_loader` provides for example.cfg
(pisa_examples/resources/settings/pipeline/example.cfg data_dict).
"""
import numpy as np

from pisa_amd import FTYPE
from pisa_amd.core.container import Container
from pisa_amd.core.stage import Stage
from pisa_amd.stages import deferred

__all__ = ["synthetic_events"]


class synthetic_events(Stage):  # pylint: disable=invalid-name
    def __init__(self, output_names, **std_kwargs):
        self.output_names = output_names
        super().__init__(expected_params=("n_events", "seed"), expected_container_keys=(),
                         supported_reps={"calc_mode": [None, "events"], "apply_mode": ["events"]},
                         **std_kwargs)

    def setup_function(self):
        from pisa_amd import synthetic

        n_per = int(self.params.n_events.value.m) // len(self.output_names)
        events = synthetic.make_events(n_per, seed=int(self.params.seed.value.m), names=self.output_names)
        for ev in events:
            c = Container(ev["name"], representation="events")
            for key in ("true_energy", "true_coszen", "reco_energy", "reco_coszen",
                        "weighted_aeff", "initial_weights"):
                c[key] = np.ascontiguousarray(ev[key], dtype=FTYPE)
            c["pid"] = np.ascontiguousarray(2.0 * ev["pid"] - 1.0, dtype=FTYPE)  # -1 cascade, +1 track
            c["nu_flux_nominal"] = ev["nu_flux"]
            c["nubar_flux_nominal"] = ev["nu_flux"] * 0.7
            c["weights"] = np.ones(n_per, dtype=FTYPE)
            c.set_aux_data("nubar", ev["nubar"])
            c.set_aux_data("flav", ev["flav"])
            self.data.add_container(c)

    def apply_function(self):
        for container in self.data:
            deferred.reset_weights(container)

def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    from pisa_amd.core.param import Param, ParamSet

    return synthetic_events(output_names=["numu_cc", "nuebar_nc"], params=ParamSet([
        Param(name="n_events", value=1000, **param_kwargs),
        Param(name="seed", value=3, **param_kwargs)]))
