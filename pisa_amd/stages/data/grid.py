"""Events on the nodes of a grid (counterpart of pisa/stages/data/grid.py:15-95): one container per output name
whose columns are the flattened `grid_binning.meshgrid(entity)`, unit weights, `nubar` / `flav` from the name."""
import numpy as np

from pisa_amd import FTYPE
from pisa_amd.core.container import Container
from pisa_amd.core.stage import Stage

__all__ = ["grid"]


class grid(Stage):  # pylint: disable=invalid-name
    def __init__(self, grid_binning, entity="midpoints", output_names=None, **std_kwargs):
        self.grid_binning = grid_binning
        self.entity = entity
        self.output_names = output_names
        super().__init__(expected_params=(), expected_container_keys=(), supported_reps={"calc_mode": "events"},
                         **std_kwargs)
        assert self.output_names is not None

    def setup_function(self):
        for name in self.output_names:
            container = Container(name, self.calc_mode)
            nubar = -1 if "bar" in name else 1
            # grid.py:65-70: three independent tests, the last that matches wins ('nue_nc' -> 0, 'nutau_cc' -> 2)
            for tag, code in (("e", 0), ("mu", 1), ("tau", 2)):
                if tag in name:
                    flav = code
            mesh = self.grid_binning.meshgrid(entity=self.entity, attach_units=False)
            size = mesh[0].size
            for var_name, var_vals in zip(self.grid_binning.names, mesh):
                container[var_name] = np.ascontiguousarray(var_vals.flatten(), dtype=FTYPE)
            container.set_aux_data("nubar", nubar)
            container.set_aux_data("flav", flav)
            container["initial_weights"] = np.ones(size, dtype=FTYPE)
            container["weights"] = np.ones(size, dtype=FTYPE)
            self.data.add_container(container)

    def apply_function(self):
        for container in self.data:
            container["weights"] = np.copy(container["initial_weights"])


def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    from pisa_amd.stages.utils.kde import service_test_binning

    return grid(grid_binning=service_test_binning(), calc_mode="events",
                output_names=["nue_cc", "numu_cc", "nutau_cc", "nuebar_cc", "numubar_cc", "nutaubar_cc",
                              "nue_nc", "numu_nc", "nutau_nc", "nuebar_nc", "numubar_nc", "nutaubar_nc"])
