"""Loader for PISA's HDF5 event files (counterpart of pisa/stages/data/simple_data_loader.py:20-263): one container
per output name with the variables of `data_dict`, optional cuts and reproducible sub-sampling; `apply_function`
resets the weights every evaluation (:258-260) -- here as a deferred operation so that the fused
reweight+histogram kernel can absorb it.  The file is read by this package's own HDF5 reader (`utils/hdf.py`)."""
import numpy as np

from pisa_amd import FTYPE
from pisa_amd.core.container import Container
from pisa_amd.core.events_pi import EventsPi
from pisa_amd.core.stage import Stage
from pisa_amd.stages import deferred

__all__ = ["simple_data_loader", "init_test"]


def _split(spec):
    if spec is None:
        return []
    if isinstance(spec, str):
        return [s.strip() for s in spec.split(",") if s.strip()]
    return list(spec)


class simple_data_loader(Stage):  # pylint: disable=invalid-name
    def __init__(self, events_file, mc_cuts, data_dict, neutrinos=True, required_metadata=None,
                 fraction_events_to_keep=None, events_subsample_index=0, seed=123456, output_names=None,
                 **std_kwargs):
        self.events_file = _split(events_file)
        self.mc_cuts = mc_cuts
        self.data_dict = eval(data_dict) if isinstance(data_dict, str) else data_dict  # pylint: disable=eval-used
        self.neutrinos = neutrinos
        self.required_metadata = _split(required_metadata) if required_metadata is not None else None
        self.fraction_events_to_keep = fraction_events_to_keep
        self.events_subsample_index = int(events_subsample_index)
        self.seed = int(seed)
        self.output_names = _split(output_names)
        super().__init__(expected_params=(), expected_container_keys=(),
                         supported_reps={"calc_mode": None, "apply_mode": "events"}, **std_kwargs)
        if len(self.output_names) != len(set(self.output_names)):
            raise ValueError("Found duplicates in `output_names`, but each name must be unique.")
        self.load_events()
        self.apply_cuts_to_events()

    def load_events(self):
        self.evts = EventsPi(name="Events", neutrinos=self.neutrinos,
                             fraction_events_to_keep=self.fraction_events_to_keep,
                             events_subsample_index=self.events_subsample_index)
        self.evts.load_events_file(events_file=self.events_file, variable_mapping=self.data_dict,
                                   required_metadata=self.required_metadata, seed=self.seed)
        self.metadata = self.evts.metadata

    def apply_cuts_to_events(self):
        if self.mc_cuts:
            self.evts = self.evts.apply_cut(self.mc_cuts)

    def setup_function(self):
        names = self.output_names if self.output_names else list(self.evts.keys())
        for name in names:
            if name not in self.evts:
                raise ValueError('Output name "%s" not found in events. Only found %s.' % (name, list(self.evts.keys())))
            container = Container(name)
            container.representation = "events"
            for key, val in self.evts[name].items():
                container[key] = np.ascontiguousarray(val, dtype=FTYPE)
            if "weights" in container.keys:
                raise KeyError('Found an existing `weights` array in "%s" which would be overwritten. Consider renaming'
                               " it to `initial_weights`." % name)
            container["weights"] = np.ones(container.size, dtype=FTYPE)
            if "initial_weights" not in container.keys:
                w0 = 1.0
                if self.fraction_events_to_keep is not None and ("nu" in name or "mu" in name):
                    w0 = 1.0 / float(self.fraction_events_to_keep)      # the sub-sample stands for the whole
                container["initial_weights"] = np.full(container.size, w0, dtype=FTYPE)
            if self.neutrinos:
                if name.startswith("nutau"):
                    flav = 2
                elif name.startswith("numu"):
                    flav = 1
                elif name.startswith("nue"):
                    flav = 0
                else:
                    raise ValueError("Cannot determine flavour of %s" % name)
                container.set_aux_data("nubar", -1 if "bar" in name else 1)
                container.set_aux_data("flav", flav)
            self.data.add_container(container)
        if len(self.data.names) == 0:
            raise ValueError("No containers created during data loading for some reason.")

    def apply_function(self):
        for container in self.data:
            deferred.reset_weights(container)


def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    return simple_data_loader(
        events_file="events/events__vlvnt__toy_1_to_80GeV_spidx1.0_cz-1_to_1_1e2evts_set0__unjoined__with_fluxes_"
                    "honda-2015-spl-solmin-aa.hdf5",
        mc_cuts="(true_coszen <= 0.5) & (true_energy <= 70)",
        data_dict={"true_energy": "true_energy", "true_coszen": "true_coszen", "reco_energy": "reco_energy",
                   "reco_coszen": "reco_coszen", "pid": "pid", "weighted_aeff": "weighted_aeff",
                   "nu_flux_nominal": ["nominal_nue_flux", "nominal_numu_flux"],
                   "nubar_flux_nominal": ["nominal_nuebar_flux", "nominal_numubar_flux"]},
        output_names=["nue_cc", "numu_cc", "nutau_cc", "nuebar_cc", "numubar_cc", "nutaubar_cc",
                      "nue_nc", "numu_nc", "nutau_nc", "nuebar_nc", "numubar_nc", "nutaubar_nc"])
