"""Per-bin scale factors from hypersurface fits to discrete systematics sets
(counterpart of pisa/stages/discr_sys/hypersurfaces.py:39-257).

`compute_function` evaluates the hyperplanes for the current detector-systematics
parameters (a [n_bins x n_params] product per linked container class, on the host:
it depends only on parameters); `apply_function` scales the binned `weights`,
`errors` and `bin_unc2` on the device (`pisa_hip_bin_scale`).
"""
import ast
from collections.abc import Mapping

import numpy as np

from pisa_amd import FTYPE
from pisa_amd import kernels as K
from pisa_amd.core.binning import MultiDimBinning
from pisa_amd.core.stage import Stage
from pisa_amd.utils import hypersurface as hs

__all__ = ["hypersurfaces"]


class hypersurfaces(Stage):  # pylint: disable=invalid-name
    def __init__(self, fit_results_file, propagate_uncertainty=False, interpolated=False,
                 links=None, fluctuate=False, fluctuate_seed=12345, **std_kwargs):
        if interpolated or fluctuate or propagate_uncertainty:
            raise NotImplementedError("interpolated / fluctuated hypersurfaces and their uncertainty "
                                      "propagation are not part of this build")
        self.fit_results_file = fit_results_file
        self.propagate_uncertainty = False
        self.hypersurfaces = hs.load_hypersurfaces(fit_results_file,
                                                   expected_binning=std_kwargs["calc_mode"])
        self.hypersurface_param_names = list(self.hypersurfaces.values())[0].param_names
        keys = ["weights"] + (["errors"] if std_kwargs.get("error_method") else [])
        super().__init__(expected_params=self.hypersurface_param_names,
                         expected_container_keys=keys,
                         supported_reps={"calc_mode": MultiDimBinning}, **std_kwargs)
        if links is None:
            self.links = {}
        elif not isinstance(links, Mapping):
            self.links = ast.literal_eval(links)
        else:
            self.links = links

    def _link(self):
        for key, val in self.links.items():
            self.data.link_containers(key, val)

    def setup_function(self):
        self._link()
        for container in self.data:
            container["hs_scales"] = np.empty(container.size, dtype=FTYPE)
        for container in self.data:
            assert container.name in self.hypersurfaces, \
                f"No match for map {container.name} found in the hypersurfaces"
        self.data.unlink_containers()

    def compute_function(self):
        self._link()
        param_values = {n: float(self.params[n].m) for n in self.hypersurface_param_names}
        for container in self.data:
            scales = self.hypersurfaces[container.name].evaluate(param_values).reshape(container.size)
            scales[~np.isfinite(scales)] = 1.0  # empty bins (:201-208)
            container["hs_scales"] = scales
            container.mark_valid("hs_scales")
        self.data.unlink_containers()

    def apply_function(self):
        for container in self.data:
            scales = container.device("hs_scales")
            if self.error_method == "sumw2":
                if self.data.representation != "events":
                    container["errors"] = K.bin_scale(container.device("errors"), scales)
                if "bin_unc2" in container.keys:
                    container["bin_unc2"] = K.bin_scale(container.device("bin_unc2"), scales, floor=0.0)
            container["weights"] = K.bin_scale(container.device("weights"), scales, floor=0.0)
