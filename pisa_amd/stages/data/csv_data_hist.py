"""Binned observed counts of the CSV data release (counterpart of
pisa/stages/data/csv_data_hist.py:20-60); implements no apply."""
import pandas as pd

from pisa_amd import FTYPE
from pisa_amd.core.container import Container
from pisa_amd.core.stage import Stage
from pisa_amd.utils.resources import find_resource

__all__ = ["csv_data_hist"]


class csv_data_hist(Stage):  # pylint: disable=invalid-name
    def __init__(self, events_file, **std_kwargs):
        self.events_file = find_resource(events_file)
        super().__init__(expected_params=(), expected_container_keys=(), **std_kwargs)

    def setup_function(self):
        events = pd.read_csv(self.events_file)
        container = Container("total")
        container.representation = self.calc_mode
        container["weights"] = events["count"].values.astype(FTYPE)
        container["reco_energy"] = events["reco_energy"].values.astype(FTYPE)
        container["reco_coszen"] = events["reco_coszen"].values.astype(FTYPE)
        container["pid"] = events["pid"].values.astype(FTYPE)
        self.data.add_container(container)

def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    return csv_data_hist(events_file="events/IceCube_3y_oscillations/data.csv.bz2")
