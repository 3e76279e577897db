"""Loader for the CSV data-release format (counterpart of
pisa/stages/data/csv_loader.py:19-171): one container per output name, selected
from the file by PDG code and interaction type; `apply_function` resets the
weights every evaluation (:168-171) -- here as a deferred operation so that the
fused reweight+histogram kernel can absorb it.
"""
import numpy as np
import pandas as pd

from pisa_amd import FTYPE
from pisa_amd.core.container import Container
from pisa_amd.core.stage import Stage
from pisa_amd.stages import deferred
from pisa_amd.utils.resources import find_resource

__all__ = ["csv_loader"]


def _split(spec):
    if isinstance(spec, str):
        return [s.strip() for s in spec.split(",") if s.strip()]
    return list(spec)


class csv_loader(Stage):  # pylint: disable=invalid-name
    def __init__(self, events_file, data_dict, output_names, neutrinos=True, dis_idx=None,
                 scale_aeff=False, **std_kwargs):
        self.events_file = [find_resource(f) for f in _split(events_file)]
        if isinstance(data_dict, str):
            self.data_dict = eval(data_dict)  # pylint: disable=eval-used  (csv_loader.py:73)
        elif isinstance(data_dict, dict):
            self.data_dict = data_dict
        else:
            raise ValueError(f"Unsupported type {type(data_dict)} for data_dict.")
        self.output_names = _split(output_names)
        if len(self.output_names) != len(set(self.output_names)):
            raise ValueError("Found duplicates in `output_names`, but each name must be unique.")
        self.neutrinos = neutrinos
        self.dis_idx = int(dis_idx) if dis_idx is not None else None
        self.scale_aeff = scale_aeff
        super().__init__(expected_params=(), expected_container_keys=(),
                         supported_reps={"calc_mode": "events", "apply_mode": "events"},
                         **std_kwargs)

    def setup_function(self):
        raw_data = pd.concat([pd.read_csv(f) for f in self.events_file])
        for name in self.output_names:
            container = Container(name)
            if self.neutrinos:
                nubar = -1 if "bar" in name else 1
                if "e" in name:
                    flav = 0
                if "mu" in name:
                    flav = 1
                if "tau" in name:
                    flav = 2
                container.set_aux_data("nubar", nubar)
                container.set_aux_data("flav", flav)
                pdg = nubar * (12 + 2 * flav)
                if "pdg_code" in raw_data:
                    mask = raw_data["pdg_code"] == pdg
                elif "pdg" in raw_data:
                    mask = raw_data["pdg"] == pdg
                else:
                    raise ValueError("Either 'pdg' or 'pdg_code' must be in file.")
                if "cc" in name:
                    mask = np.logical_and(mask, raw_data["type"] >= 1)
                else:
                    mask = np.logical_and(mask, raw_data["type"] == 0)
                events = raw_data[mask]
            else:
                events = raw_data
            container["initial_weights"] = np.ones(len(events), dtype=FTYPE)
            container["weights"] = np.ones(len(events), dtype=FTYPE)
            for key, val in self.data_dict.items():
                container[key] = np.ascontiguousarray(events[val].values.astype(FTYPE))
            if self.scale_aeff and "weighted_aeff" in container.keys:
                container["weighted_aeff"] = container["weighted_aeff"] * 1.0e-4
            if ("dis" not in container.keys and "interaction" in container.keys
                    and self.dis_idx is not None):
                container["dis"] = (container["interaction"] == self.dis_idx).astype(FTYPE)
            self.data.add_container(container)
        if len(self.data.names) == 0:
            raise ValueError("No containers created during data loading for some reason.")

    def apply_function(self):
        for container in self.data:
            deferred.reset_weights(container)

def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    data_dict = {"true_energy": "true_energy", "true_coszen": "true_coszen", "weighted_aeff": "weight",
                 "reco_energy": "reco_energy", "reco_coszen": "reco_coszen", "pid": "pid"}
    return csv_loader(events_file="events/IceCube_3y_oscillations/neutrino_mc.csv.bz2", data_dict=data_dict,
                      output_names=["nue_cc", "numu_cc"])
