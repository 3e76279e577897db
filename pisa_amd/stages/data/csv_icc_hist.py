"""Binned atmospheric-muon template of the CSV data release (counterpart of
pisa/stages/data/csv_icc_hist.py:22-84): weights = count * atm_muon_scale."""
import numpy as np
import pandas as pd

from pisa_amd import FTYPE
from pisa_amd import kernels as K
from pisa_amd.core.container import Container
from pisa_amd.core.stage import Stage
from pisa_amd.utils.resources import find_resource

__all__ = ["csv_icc_hist"]


class csv_icc_hist(Stage):  # pylint: disable=invalid-name
    def __init__(self, events_file, **std_kwargs):
        self.events_file = find_resource(events_file)
        super().__init__(expected_params=("atm_muon_scale",), expected_container_keys=(),
                         **std_kwargs)

    def setup_function(self):
        events = pd.read_csv(self.events_file)
        # one pseudo-event per bin, placed at the bin midpoint; the binned
        # representation is obtained by the container's events -> binned translation,
        # as in the reference (no assumption on the row order of the file)
        container = Container("icc")
        container["count"] = events["count"].values.astype(FTYPE)
        container["weights"] = np.ones(container.size, dtype=FTYPE)
        key = "abs_uncert" if "abs_uncert" in events else "abs_uncertainty"
        container["errors"] = events[key].values.astype(FTYPE)
        container["reco_energy"] = events["reco_energy"].values.astype(FTYPE)
        container["reco_coszen"] = events["reco_coszen"].values.astype(FTYPE)
        container["pid"] = events["pid"].values.astype(FTYPE)
        self.data.add_container(container)

    def apply_function(self):
        scale = self.params.atm_muon_scale.m_as("dimensionless")
        for container in self.data:
            container["weights"] = K.bin_scale(container.device("count"), None, float(scale))

def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    from pisa_amd.core.param import Param, ParamSet

    return csv_icc_hist(events_file="events/IceCube_3y_oscillations/muons.csv.bz2",
                        params=ParamSet([Param(name="atm_muon_scale", value=0.2, **param_kwargs)]))
