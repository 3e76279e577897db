"""Weighted histogramming of events into the output binning (counterpart of
pisa/stages/utils/hist.py:19-223; events calc_mode).

hist = sum(w), errors = sqrt(sum(w^2)), bin_unc2 = sum(unc^2 * w) on the
regularised (all-linear) binning: log dimensions are binned in ln(x), irregular
ones are pre-digitised at setup (hist.py:86-127).

Fast path: if every container's `weights` carry the deferred chain
[reset, osc, aeff] (loader -> osc.prob3 on a 2-D calc grid -> aeff.aeff), the
three apply_functions and the three histogram passes of the reference
(hist.py:198-209) run as ONE pass over HBM through `HotPathEngine`
(`pisa_hip_reweight_hist`), with exact, order-independent accumulation.
Otherwise the weights are materialised and histogrammed by
`pisa_hip_histogram_regular` -- same results, more passes.
"""
import numpy as np
import torch

from pisa_amd import FTYPE
from pisa_amd import kernels as K
from pisa_amd.core.binning import MultiDimBinning
from pisa_amd.core.container import regularized
from pisa_amd.core.stage import Stage
from pisa_amd.stages import deferred

__all__ = ["hist"]


class hist(Stage):  # pylint: disable=invalid-name
    def __init__(self, apply_unc_weights=False, unweighted=False, **std_kwargs):
        keys = ["weights"] + (["unc_weights"] if apply_unc_weights else [])
        supported_reps = {"calc_mode": [MultiDimBinning, "events"],
                          "apply_mode": [None, MultiDimBinning]}
        super().__init__(expected_params=(), expected_container_keys=keys,
                         supported_reps=supported_reps, **std_kwargs)
        self.apply_unc_weights = apply_unc_weights
        self.unweighted = unweighted
        self._engine = None
        self._engine_versions = None
        self._rows = []           # BlockRows handed to the containers by the last fused evaluation
        self.fused_last_eval = False

    def setup_function(self):
        if self.apply_mode is None:
            self.apply_mode = self.data["output_binning"]
        else:
            assert self.apply_mode == self.data["output_binning"]
        if isinstance(self.calc_mode, MultiDimBinning):
            self._setup_transforms()
            return
        # regularised binning + per-container sample columns (device), once
        self._samples = {}
        for container in self.data.containers:
            container.representation = "events"
            binning, cols = regularized(self.apply_mode, lambda n, log, c=container: (
                np.log(c[n]) if log else c[n]))
            self._samples[container.name] = [K.to_device(np.asarray(col, dtype=FTYPE)) for col in cols]
        self._reg_binning = binning
        self.data["regularized_output_binning"] = binning
        self._engine = None

    # ------------------------------------------------------------------ binned calc_mode
    def _setup_transforms(self):
        """hist.py:69-84: `hist_transform[i, j]` = number of events in calc bin i and output bin j
        (the histogram of the events in the joint binning calc_mode + apply_mode).  Kept dense in
        the container, as the reference does, and as its non-zeros grouped by output bin for
        `pisa_hip_transform_apply`."""
        assert len(set(self.calc_mode.names) & set(self.apply_mode.names)) == 0, \
            "calc_mode and apply_mode must not share dimensions"
        n_calc, n_out = self.calc_mode.size, self.apply_mode.size
        self._csr = {}
        for container in self.data.containers:
            container.representation = "events"
            getcol = lambda n, log, c=container: (np.log(c[n]) if log else c[n])  # noqa: E731
            b_calc, c_calc = regularized(self.calc_mode, getcol)
            b_out, c_out = regularized(self.apply_mode, getcol)
            i = K.event_indices([K.to_device(np.asarray(x, dtype=FTYPE)) for x in c_calc], b_calc).long()
            j = K.event_indices([K.to_device(np.asarray(x, dtype=FTYPE)) for x in c_out], b_out).long()
            ok = (i >= 0) & (j >= 0)
            keys, counts = torch.unique(j[ok] * n_calc + i[ok], return_counts=True)   # sorted: by j, then i
            col = (keys % n_calc).to(torch.int32).contiguous()
            rows = keys // n_calc
            ptr = torch.zeros(n_out + 1, dtype=torch.int64, device=keys.device)
            ptr[1:] = torch.cumsum(torch.bincount(rows, minlength=n_out), 0)
            val = counts.to(torch.float64).contiguous()
            self._csr[container.name] = (ptr.to(torch.int32).contiguous(), col, val)
            dense = torch.zeros(n_calc * n_out, dtype=torch.float64, device=keys.device)
            dense[(keys % n_calc) * n_out + rows] = val
            container.representation = self.calc_mode
            container["hist_transform"] = dense.view(n_calc, n_out)

    def _apply_transforms(self):
        """hist.py:132-160"""
        if self.unweighted:
            raise NotImplementedError("Unweighted hist only implemented in event-wise calculation")
        sumw2 = self.error_method == "sumw2"
        for container in self.data.containers:
            container.representation = self.calc_mode
            weights = container.device("weights")
            if "astro_weights" in container.keys:
                weights = weights + container.device("astro_weights")
            unc = container.device("unc_weights") if self.apply_unc_weights else None
            ptr, col, val = self._csr[container.name]
            h, s2, u2 = K.transform_apply(weights.contiguous(), unc, ptr, col, val, self.apply_mode.size,
                                          errors=sumw2)
            container.representation = self.apply_mode
            container["weights"] = h
            if sumw2:
                container["errors"] = torch.sqrt(s2)
                container["bin_unc2"] = u2

    # ------------------------------------------------------------------ fused
    def _find_prob3(self):
        stage = self.data._glob_aux_data.get("_prob3_stage")
        return stage if (stage is not None and stage.grid is not None and stage.pepmu is not None) else None

    def _fused(self):
        if self.unweighted or self.apply_unc_weights:
            return False
        conts = self.data.containers
        chains = [deferred.fusable_chain(c) for c in conts]
        if any(ch is None for ch in chains) or any("astro_weights" in c.all_keys for c in conts):
            return False
        osc = self._find_prob3()
        if osc is None:
            return False
        cm = osc.calc_mode
        e_dim, cz_dim = cm["true_energy"], cm["true_coszen"]
        # the engine's event -> node index assumes the usual oscillogram grid (log-uniform
        # energy, lin-uniform coszen, `GridSpec`); any other calc grid takes the unfused path,
        # whose lookups go through `regularized()` and handle every binning
        if (e_dim.is_irregular or cz_dim.is_irregular or not e_dim.is_log or not cz_dim.is_lin):
            return False
        from pisa_amd.engine import GridSpec, HotPathEngine

        flux_key = chains[0][0]
        static_keys = ("weighted_aeff", "initial_weights", "true_energy", "true_coszen")
        if self._engine is not None:
            for c, v in zip(conts, self._engine_versions):
                cv = c._version
                if (cv["weighted_aeff"] != v["weighted_aeff"] or cv["initial_weights"] != v["initial_weights"]
                        or cv["true_energy"] != v["true_energy"] or cv["true_coszen"] != v["true_coszen"]):
                    self._engine = None   # a column folded / digitised at engine build was rewritten
                    break
        # flux computed on the oscillation grid (flux stages with calc_mode = osc.prob3's, e.g. the
        # IceCube 3-year cfgs): every event would look up flux AND probabilities at the same node
        # (container.py:981-1012), so the engine multiplies them per node instead of per event
        cm_hash = hash(cm)
        node_flux = all(c.validity[flux_key].get(cm_hash, False) for c in conts)
        if self._engine is not None and self._engine.node_flux != node_flux:
            self._engine = None

        def flux_on_nodes(c):
            c.representation = cm
            try:
                return c.device(flux_key)
            finally:
                c.representation = "events"

        if self._engine is None:
            g = osc.grid
            grid = GridSpec(tuple(e_dim.domain.m_as("GeV")), e_dim.num_bins,
                            tuple(cz_dim.domain.magnitude), cz_dim.num_bins, energy_first=g["e_major"])
            import torch.distributed as dist

            world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
            rank = dist.get_rank() if world > 1 else 0
            evs = []
            for c in conts:
                c.representation = "events"
                ev = dict(name=c.name, flav=int(c["flav"]), nubar=int(c["nubar"]),
                          true_energy=c["true_energy"], true_coszen=c["true_coszen"],
                          weighted_aeff=c["weighted_aeff"], initial_weights=c["initial_weights"],
                          sample=[s.cpu().numpy() for s in self._samples[c.name]], scale=1.0)
                if node_flux:
                    ev["nu_flux_nodes"] = flux_on_nodes(c)
                else:
                    ev["nu_flux"] = c[flux_key]
                evs.append(ev)
            # compact columns: initial_weights*weighted_aeff folded into the flux pair once
            # (refreshed by update_flux below whenever a flux systematic moved)
            from pisa_amd.engine import configured_points

            self._engine = HotPathEngine(evs, grid, self._reg_binning, None, 0, rank=rank,
                                         world_size=world, external_tables=True, compact=True,
                                         node_flux=node_flux, points=configured_points())
            self._engine_versions = [{k: c.version(k) for k in static_keys + (flux_key,)}
                                     for c in conts]
            self._node_flux_src = (flux_key, cm) if node_flux else None
        eng = self._engine
        moved = []
        for i, (c, ch) in enumerate(zip(conts, chains)):
            if eng.cont[i].scale != ch[1]:
                eng.set_scale(c.name, ch[1])
            # the container's change counter, not object identity: a stage that edits the flux in
            # place and calls mark_changed (container.py:638-649) keeps the same array object
            if c._version[flux_key] != self._engine_versions[i][flux_key]:
                c.representation = "events"
                if node_flux:                            # flux systematics changed
                    eng.update_flux_nodes(i, flux_on_nodes(c))
                else:
                    moved.append((i, c.device(flux_key)))
                self._engine_versions[i][flux_key] = c.version(flux_key)
        eng.update_flux_many(moved)                      # one launch for all rewritten columns
        eng.pepmu = osc.pepmu
        # the rows handed out at the previous evaluation that nobody read are void from here on (their
        # table is about to be overwritten; the containers get new rows below); Maps somebody still HOLDS
        # keep the old block alive and are brought to the host by the engine before the launch
        for row in self._rows:
            if row.pristine:
                row.block = None
        self._rows = []
        eng.accumulate()
        eng.allreduce()
        # Nothing is finalised or copied here: the maps stay in HBM as the int64 limbs of this evaluation.
        # The containers receive ROWS of that table (core/container.py: BlockRow) -- host arrays / device
        # tensors on first access, device-backed Maps through `get_mapset`, whose total and metric run on the
        # device (the tail kernel) without the maps travelling at all.
        import weakref

        from pisa_amd.core.container import BlockRow
        from pisa_amd.core.fastplan import DeviceMapBlock

        sumw2 = self.error_method == "sumw2"
        block = DeviceMapBlock(eng, sumw2)
        block.rows_published = True
        eng._out_block = weakref.ref(block)
        n_bins = eng.n_bins
        rows = self._rows
        for i, c in enumerate(conts):
            ops = c.pending.pop(deferred.KEY, None)  # consumed by the fused kernel
            c.representation = self.apply_mode
            r = BlockRow(block, i, 0, n_bins)
            rows.append(r)
            c.publish("weights", r)
            # histogramming does not invalidate the event-wise weights (hist.py:213): they are
            # what the consumed chain gives, computed if anybody reads them
            c.keep_lazy("weights", ops, "events")
            if sumw2:
                r1, r2 = BlockRow(block, i, 1, n_bins), BlockRow(block, i, 2, n_bins)
                rows += [r1, r2]
                c.publish("errors", r1)
                c.publish("bin_unc2", r2)          # sum(1^2 * w), hist.py:207-209
        return True

    def sync_node_flux(self):
        """node-flux engine: hand over the flux of every container whose flux column was rewritten
        since the engine saw it (used by the evaluation plan after it re-ran a flux stage)"""
        key, cm = self._node_flux_src
        eng = self._engine
        for i, c in enumerate(self.data.containers):
            if c.version(key) != self._engine_versions[i][key]:
                keep = c.representation
                c.representation = cm
                try:
                    eng.update_flux_nodes(i, c.device(key))
                finally:
                    c.representation = keep
                self._engine_versions[i][key] = c.version(key)

    # ------------------------------------------------------------------ apply
    def apply_function(self):
        if isinstance(self.calc_mode, MultiDimBinning):
            self.fused_last_eval = False
            self._apply_transforms()
            return
        self.fused_last_eval = self._fused()
        if self.fused_last_eval:
            return
        for container in self.data:
            container.representation = "events"
            sample = self._samples[container.name]
            weights = container.device("weights")  # materialises any deferred chain
            if "astro_weights" in container.keys:
                weights = weights + container.device("astro_weights")
            if self.unweighted:
                weights = torch.ones_like(weights)
            unc = container.device("unc_weights") if self.apply_unc_weights else None
            w = weights if unc is None else unc * weights
            h = K.histogram_regular(sample, w, self._reg_binning)
            if self.error_method == "sumw2":
                sumw2 = K.histogram_regular(sample, w * w, self._reg_binning)
                bin_unc2 = h if unc is None else K.histogram_regular(sample, unc * unc * weights,
                                                                    self._reg_binning)
            container.representation = self.apply_mode
            container["weights"] = h
            # histogramming does not invalidate the event-wise weights (hist.py:213)
            container.validity["weights"][hash("events")] = True
            if self.error_method == "sumw2":
                container["errors"] = torch.sqrt(sumw2)
                container["bin_unc2"] = bin_unc2

def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    return hist(calc_mode="events")
