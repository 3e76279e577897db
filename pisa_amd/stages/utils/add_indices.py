"""The output bin of every event as a column, and one mask per bin (counterpart of
pisa/stages/utils/add_indices.py:25-80): `bin_indices` (events; -1 below / n_bins above the binning,
`core.bin_indexing.lookup_indices`) and, in the binned representation, `bin_<i>_mask` = (`bin_indices` seen
through that binning == i) -- the reference builds the masks after switching the representation, so each is a
per-BIN array of the events' average index (:76-80)."""
from pisa_amd.core.bin_indexing import lookup_indices
from pisa_amd.core.binning import MultiDimBinning
from pisa_amd.core.stage import Stage

__all__ = ["add_indices"]


class add_indices(Stage):  # pylint: disable=invalid-name
    def __init__(self, **std_kwargs):
        super().__init__(expected_params=(), expected_container_keys=(),
                         supported_reps={"calc_mode": "events", "apply_mode": MultiDimBinning}, **std_kwargs)

    def setup_function(self):
        if self.calc_mode != "events":
            raise ValueError('calc mode must be set to "events" for this module')
        if not isinstance(self.apply_mode, MultiDimBinning):
            raise ValueError("apply mode must be set to a binning")
        for container in self.data:
            self.data.representation = self.calc_mode
            container["bin_indices"] = lookup_indices([container.device(n) for n in self.apply_mode.names],
                                                      self.apply_mode)
            self.data.representation = self.apply_mode
            seen = container["bin_indices"]
            for i in range(self.apply_mode.tot_num_bins):
                container["bin_%d_mask" % i] = seen == i


def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    from pisa_amd.stages.utils.kde import service_test_binning

    return add_indices(calc_mode="events", apply_mode=service_test_binning())
