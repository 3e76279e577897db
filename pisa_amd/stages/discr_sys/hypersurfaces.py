"""Per-bin scale factors from hypersurface fits to discrete systematics sets
(counterpart of pisa/stages/discr_sys/hypersurfaces.py:39-257).

`compute_function` evaluates the hypersurfaces (data-release hyperplanes or `fit_hypersurfaces`
JSON files; optional uncertainty from the fit covariance, optional fluctuation) for the current
detector-systematics parameters -- per linked container class, on the host: it depends only on
parameters and a few hundred bins; `apply_function` scales the binned `weights`,
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
        self.fit_results_file = fit_results_file
        self.propagate_uncertainty = bool(propagate_uncertainty)
        self.interpolated = bool(interpolated)
        self.fluctuate = bool(fluctuate)
        self.fluctuate_seed = fluctuate_seed
        if self.fluctuate:
            assert self.fluctuate_seed is not None
        # the expected parameters depend on the file: the hypersurfaces' own parameters and, for
        # interpolated hypersurfaces, the parameters they are interpolated in (:97-106)
        self.inter_params = []
        if self.interpolated:
            self.hypersurfaces = hs.load_interpolated_hypersurfaces(fit_results_file,
                                                                    expected_binning=std_kwargs["calc_mode"])
            self.inter_params = list(self.hypersurfaces.values())[0].interpolation_param_names
        else:
            self.hypersurfaces = hs.load_hypersurfaces(fit_results_file,
                                                       expected_binning=std_kwargs["calc_mode"])
        self.hypersurface_param_names = list(self.hypersurfaces.values())[0].param_names
        keys = ["weights"] + (["errors"] if std_kwargs.get("error_method") else [])
        super().__init__(expected_params=self.hypersurface_param_names + self.inter_params,
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
            if self.propagate_uncertainty:
                container["hs_scales_uncertainty"] = np.empty(container.size, dtype=FTYPE)
        for container in self.data:
            assert container.name in self.hypersurfaces, \
                f"No match for map {container.name} found in the hypersurfaces"
        self.data.unlink_containers()

    def _linear_block(self, conts):
        """All surfaces linear in every parameter, nothing interpolated / fluctuated / propagated (the data-release
        hyperplanes of the published 3-year analysis): intercepts and coefficients of all containers stacked once,
        so that one evaluation is `len(params)` numpy operations on [n_surfaces, bins] instead of that many per
        surface -- element by element the very additions of `Hypersurface.evaluate`, in its order."""
        blk = getattr(self, "_lin_block", None)
        names = tuple(c.name for c in conts)
        if blk is not None and blk["names"] == names:
            return blk
        surfaces = [self.hypersurfaces[n] for n in names]
        ok = (not self.interpolated and not self.fluctuate and not self.propagate_uncertainty
              and all(list(sf.params.keys()) == self.hypersurface_param_names for sf in surfaces)
              and all(p.func_name == "linear" for sf in surfaces for p in sf.params.values())
              and len({sf.log for sf in surfaces}) == 1 and len({sf.using_legacy_data for sf in surfaces}) == 1
              and all(sf.intercept.size == conts[0].size for sf in surfaces))
        blk = self._lin_block = dict(names=names, ok=ok)
        if ok:
            blk["intercept"] = np.stack([np.asarray(sf.intercept, dtype=FTYPE).reshape(-1) for sf in surfaces])
            blk["coef"] = [np.stack([sf.params[n].fit_coeffts[..., 0].reshape(-1) for sf in surfaces])
                           for n in self.hypersurface_param_names]
            blk["nominal"] = [[sf.params[n].nominal_value for sf in surfaces] for n in self.hypersurface_param_names]
            blk["log"], blk["legacy"] = surfaces[0].log, surfaces[0].using_legacy_data
        return blk

    def compute_function(self):
        self._link()
        param_values = {n: float(self.params[n].m) for n in self.hypersurface_param_names}
        conts = list(self.data)
        blk = self._linear_block(conts)
        if blk["ok"]:
            out = blk["intercept"].copy()
            for n, coef, nominal in zip(self.hypersurface_param_names, blk["coef"], blk["nominal"]):
                v = param_values[n]
                if blk["legacy"]:
                    out += coef * v                      # _lin(p, m) = m * p, out += ...
                elif len(set(nominal)) == 1:
                    out += coef * (v - nominal[0])
                else:
                    out += coef * (v - np.asarray(nominal, dtype=FTYPE))[:, None]
            scales = np.exp(out) if blk["log"] else out
            scales[~np.isfinite(scales)] = 1.0           # empty bins (:201-210)
            for i, container in enumerate(conts):
                container["hs_scales"] = scales[i]
                container.mark_valid("hs_scales")
            self.data.unlink_containers()
            return
        # the same fluctuation on every call (:176-178)
        rs = np.random.RandomState(self.fluctuate_seed) if self.fluctuate else None
        inter = {n: self.params[n].value for n in self.inter_params}
        for container in self.data:
            surface = self.hypersurfaces[container.name]
            if self.interpolated:
                # the hypersurface at the current values of the interpolation parameters (:186-189)
                surface = surface.get_hypersurface(**inter)
            if self.fluctuate:
                surface = surface.fluctuate(random_state=rs)
            if self.propagate_uncertainty:
                scales, unc = surface.evaluate(param_values, return_uncertainty=True)
                scales, unc = scales.reshape(container.size), unc.reshape(container.size)
            else:
                scales = surface.evaluate(param_values).reshape(container.size)
            empty = ~np.isfinite(scales)     # empty bins (:201-210)
            scales[empty] = 1.0
            container["hs_scales"] = scales
            container.mark_valid("hs_scales")
            if self.propagate_uncertainty:
                unc[empty] = 0.0
                container["hs_scales_uncertainty"] = unc
                container.mark_valid("hs_scales_uncertainty")
        self.data.unlink_containers()

    def apply_function(self):
        # all containers at once: the maps are a few hundred bins each, so the cost is the
        # number of launches and copies, not the arithmetic
        import torch

        conts = list(self.data)
        scales = torch.stack([c.device("hs_scales") for c in conts])

        def scaled(key, floor):
            x = torch.stack([c.device(key) for c in conts])
            out = K.bin_scale(x.reshape(-1), scales.reshape(-1), floor=floor).reshape(x.shape)
            host = out.cpu().numpy()  # the maps are read on the host right after (get_outputs)
            for i, c in enumerate(conts):
                c.set_mirrored(key, out[i], host[i])

        if self.error_method == "sumw2":
            if self.data.representation == "events":
                pass                                       # error propagation skipped in events mode (:243-246)
            elif self.propagate_uncertainty:
                # errors = weights * hs_scales_uncertainty, with the UNSCALED weights (:248-249)
                w = torch.stack([c.device("weights") for c in conts])
                u = torch.stack([c.device("hs_scales_uncertainty") for c in conts])
                out = K.bin_scale(w.reshape(-1), u.reshape(-1), floor=None).reshape(w.shape)
                host = out.cpu().numpy()
                for i, c in enumerate(conts):
                    c.set_mirrored("errors", out[i], host[i])
            else:
                scaled("errors", None)                     # errors *= hs_scales (:251)
            if all("bin_unc2" in c.keys for c in conts):
                scaled("bin_unc2", 0.0)                    # clip(bin_unc2 * hs_scales, 0, inf) (:254-256)
        scaled("weights", 0.0)                             # clip(weights * hs_scales, 0, inf) (:259)

def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    from pisa_amd.core.binning import OneDimBinning
    from pisa_amd.core.param import Param, ParamSet
    from pisa_amd.core.units import ureg

    param_set = ParamSet([
        Param(name="opt_eff_overall", value=1.0, **param_kwargs),
        Param(name="opt_eff_lateral", value=25, **param_kwargs),
        Param(name="opt_eff_headon", value=0.0, **param_kwargs),
        Param(name="ice_scattering", value=0.0, **param_kwargs),
        Param(name="ice_absorption", value=0.0, **param_kwargs)])
    dd_en = OneDimBinning("reco_energy", is_log=True,
                          bin_edges=[5.62341325, 7.49894209, 10.0, 13.33521432, 17.7827941, 23.71373706, 31.6227766,
                                     42.16965034, 56.23413252] * ureg.GeV)
    dd_cz = OneDimBinning("reco_coszen", num_bins=8, is_lin=True, domain=[-1, 1])
    dd_pid = OneDimBinning("pid", bin_edges=[-0.5, 0.5, 1.5])
    return hypersurfaces(params=param_set, fit_results_file="events/IceCube_3y_oscillations/hyperplanes_*.csv.bz2",
                         error_method="sumw2", calc_mode=MultiDimBinning([dd_en, dd_cz, dd_pid], name="dragon_datarelease"),
                         links={"nue_cc+nuebar_cc": ["test1_cc", "test2_nc"]})
