"""KDE-smoothed output maps (counterpart of pisa/stages/utils/kde.py:19-305).

Same constructor kwargs and defaults as the reference (:52-66); log dimensions
are smoothed in ln-space (`linearize_log_dims`, :106-130); the event weights are
whatever the previous stages produced (a deferred reweighting chain is
materialised first).  `stash_hists` memoises the maps (:157-164, 280-293);
`bootstrap` gives per-bin errors from `bootstrap_niter` resampled estimates
(:189-258: the same `numpy.random.default_rng(seed).integers` / `bincount` draws,
so the same seed resamples the same events).

The sample columns never change between evaluations: they -- and, with
`stack_pid`, the per-channel event indices -- are kept in HBM from the first
evaluation on; per evaluation only the weights move.  The estimator itself is
native code (`csrc/kde.hip`: cell-list Gaussian cut-off at kernel value `tol`,
extra kwarg of this build, 0 = all pairs).  The stage's default `tol` is 1e-14
(`KDE_STAGE_TOL`: differences to the all-pairs evaluation at rounding level); a
looser cut-off is an explicit choice of the cfg (`tol = 1e-12`, `KDE_FAST_TOL`:
the pilot's series on the matrix cores, see INTEGRATION.md for what it costs in
accuracy and when it is safe).
"""
import os

import numpy as np
import torch

from pisa_amd import FTYPE
from pisa_amd.core.binning import MultiDimBinning, OneDimBinning
from pisa_amd.core.stage import Stage
from pisa_amd.utils import kde_hist

__all__ = ["kde", "KDE_STAGE_TOL", "KDE_FAST_TOL"]

# Cut-off of the stage's estimators.  Measured against the all-pairs evaluation (tol = 0) on the C3 workload (1e7 events, 24
# estimators, 2 400 bins spanning 8 decades of content; scripts/dev/kde_tol_budget.py, EXPERIMENTS R5-3), largest relative
# difference of any bin: 1.3e-13 at tol = 1e-14 (rounding), 4.9e-13 at 1e-13, 6.2e-12 at 1e-12, 6.3e-11 at 1e-11,
# 4.5e-10 at 1e-10.  The reference evaluates all pairs, so the DEFAULT is the cut-off that is indistinguishable from it
# (1e-14, the one the estimator objects `kernels.KdeEstimator` / `kde_hist.gaussian_kde` use as well).  1e-12 kept a factor
# 16 to the 1e-10 budget on that one workload; the truncated tails weigh more in a bin the sparser it is, so a sample whose
# sparsest bins lie two more decades down would leave the budget (round-5 advisor): 1e-12 is `KDE_FAST_TOL`, chosen in the
# cfg (`tol = 1e-12`) by whoever has checked it against `tol = 0` on their sample -- bench.py's C3 leg does and says so.
KDE_STAGE_TOL = 1e-14
KDE_FAST_TOL = 1e-12


class kde(Stage):  # pylint: disable=invalid-name
    def __init__(self, bw_method="silverman", coszen_name="reco_coszen", oversample=10,
                 coszen_reflection=0.25, adaptive=True, alpha=0.1, stack_pid=True,
                 stash_hists=False, bootstrap=False, bootstrap_niter=10, bootstrap_seed=None,
                 linearize_log_dims=True, tol=None, **std_kargs):
        self.bw_method = bw_method
        self.coszen_name = coszen_name
        self.oversample = int(oversample)
        self.coszen_reflection = float(coszen_reflection)
        self.alpha = float(alpha)
        self.adaptive = adaptive
        self.stack_pid = stack_pid
        self.stash_hists = stash_hists
        self.stash_valid = False
        self.stashed_hists = None
        self.stashed_errors = None
        self.linearize_log_dims = linearize_log_dims
        self.bootstrap = bootstrap
        self.bootstrap_niter = int(bootstrap_niter)
        self.bootstrap_seed = int(bootstrap_seed) if bootstrap_seed is not None else None
        self.tol = KDE_STAGE_TOL if tol is None else float(tol)
        if self.bootstrap and self.oversample > 1:
            # errors inside a bin are highly correlated (kde_hist.py:67-70)
            raise ValueError("Bootstrapping cannot be combined with oversampling.")
        super().__init__(expected_params=(), expected_container_keys=(),
                         supported_reps={"calc_mode": "events", "apply_mode": MultiDimBinning},
                         **std_kargs)
        self.regularized_apply_mode = None
        self._static = {}
        self.stats = {}

    def setup_function(self):
        self._static = {}
        self.stash_valid = False
        if not self.linearize_log_dims:
            self.regularized_apply_mode = self.apply_mode
            return
        dims = []
        for dim in self.apply_mode:
            if dim.is_lin:
                dims.append(dim)
            elif dim.is_irregular:
                dims.append(OneDimBinning(dim.name, bin_edges=np.log(dim.edge_magnitudes)))
            else:
                dims.append(OneDimBinning(dim.name, domain=np.log(dim.domain.magnitude),
                                          num_bins=dim.num_bins))
        self.regularized_apply_mode = MultiDimBinning(dims)

    def _static_sample(self, container):
        """device sample [N, D] (+ pid channels) of a container; rebuilt when a column changed"""
        names = [d.name for d in self.regularized_apply_mode]
        versions = tuple(container.version(n) for n in names)
        hit = self._static.get(container.name)
        if hit is not None and hit["versions"] == versions:
            return hit
        cols = []
        for dim, orig in zip(self.regularized_apply_mode, self.apply_mode):
            container.representation = ("log_events" if (orig.is_log and self.linearize_log_dims)
                                        else "events")
            cols.append(container.device(dim.name))
        container.representation = "events"
        sample = torch.stack(cols, dim=1).contiguous()
        chans = kde_hist.pid_channels(sample, self.regularized_apply_mode) if self.stack_pid else None
        hit = dict(versions=versions, sample=sample, channels=chans)
        self._static[container.name] = hit
        return hit

    def _bootstrap_weights(self, rng, st, size):
        """how often each event is drawn when the sample (each pid channel, if stacking) is
        resampled with replacement (stages/utils/kde.py:197-238)"""
        if self.stack_pid:
            sw = torch.zeros(size, dtype=torch.float64, device=st["sample"].device)
            for idx, _ in st["channels"][2]:
                n_ch = int(idx.numel())
                draws = np.bincount(rng.integers(n_ch, size=n_ch), minlength=n_ch)
                sw[idx] += torch.as_tensor(draws, dtype=torch.float64).to(sw.device)
            return sw
        draws = np.bincount(rng.integers(size, size=size), minlength=size)
        return torch.as_tensor(draws, dtype=torch.float64).to(st["sample"].device)

    # -- the estimators of one evaluation are independent (one per container and pid channel) and
    #    each is a chain of small launches with a few host round trips (moments, cell heads,
    #    bandwidth range): they go to the library in ONE call (`kde_hist.kde_histogramdd_batch` ->
    #    `pisa_hip_kde_lattice_batch`), which runs them side by side on `kde_workers` threads of its
    #    own, each with its own HIP stream.  Every estimator is deterministic by itself, so the maps
    #    do not depend on the interleaving.
    kde_workers = int(os.environ.get("PISA_KDE_WORKERS", "8"))

    def _job_kwargs(self, st):
        return dict(sample=st["sample"], binning=self.regularized_apply_mode,
                    bw_method=self.bw_method, coszen_name=self.coszen_name,
                    coszen_reflection=self.coszen_reflection, adaptive=self.adaptive,
                    alpha=self.alpha, oversample=self.oversample, stack_pid=self.stack_pid,
                    tol=self.tol, channels=st["channels"])

    def _bootstrap_map(self, st, weights, kw):
        """mean and standard deviation of the maps of `bootstrap_niter` resampled samples
        (stages/utils/kde.py:189-258)"""
        rng = np.random.default_rng(self.bootstrap_seed)
        # the resampled samples are independent: their estimators go to the library as one job list (the draws are
        # made in the reference's order first); maps = those of kde_histogramdd one resample after the other
        resampled = []
        for _ in range(self.bootstrap_niter):
            sw = self._bootstrap_weights(rng, st, int(weights.numel()))
            resampled.append(dict(sample=kw["sample"], weights=weights * sw, channels=kw.get("channels")))
        bkw = {k: v for k, v in kw.items() if k not in ("sample", "channels")}
        try:
            maps = kde_hist.kde_histogramdd_batch(resampled, stats=self.stats, n_threads=self.kde_workers, **bkw)
        except Exception as exc:
            raise RuntimeError(
                "Could not calculate KDE with the given sample. This can happen if the "
                "bootstrap selects too few distinct events in one of the PID channels."
            ) from exc
        for m in maps:
            if not np.all(np.isfinite(m)):
                raise RuntimeError("Could not calculate KDE with the given sample (non-finite map).")
        maps = np.stack(maps)
        return np.mean(maps, axis=0), np.std(maps, axis=0)

    @staticmethod
    def owned_containers(n_containers, rank, world_size, sizes=None):
        """Multi-GPU: the estimators of one evaluation are independent objects (one per container
        and pid channel), so the CONTAINERS are dealt to the ranks -- no event ever crosses a rank,
        bandwidths and pilot densities stay exact -- and the finished maps (n_bins doubles per
        container) are exchanged in one all-reduce in which every entry has exactly one non-zero
        contribution: the same bits as on one GPU.
        `sizes` (events per container): longest-processing-time assignment -- containers in order of
        decreasing size, each to the rank with the least events so far (ties: lowest rank), the
        estimator's cost growing with its sample; a real sample is far from balanced (nu_mu CC >>
        nu_tau NC) and 12 containers on 8 ranks dealt round-robin leave a 2:1 imbalance even for equal
        sizes.  Without sizes: round-robin.  Every rank computes the same assignment."""
        if sizes is None:
            return [i for i in range(n_containers) if i % world_size == rank]
        assert len(sizes) == n_containers
        load = [0] * world_size
        owner = [0] * n_containers
        for i in sorted(range(n_containers), key=lambda j: (-int(sizes[j]), j)):
            r = min(range(world_size), key=lambda q: (load[q], q))
            owner[i] = r
            load[r] += int(sizes[i])
        return [i for i in range(n_containers) if owner[i] == rank]

    @staticmethod
    def exchange_maps(local, n_containers, n_bins, with_errors, group=None):
        """`local`: {container index: (map[n_bins], errors[n_bins] or None)} of this rank ->
        the same for all containers on every rank (all-reduce SUM; the ranks' entries are disjoint)"""
        import torch.distributed as dist

        buf = np.zeros((n_containers, 2 if with_errors else 1, n_bins), dtype=np.float64)
        for i, (m, e) in local.items():
            buf[i, 0] = np.asarray(m).ravel()
            if with_errors:
                buf[i, 1] = np.asarray(e).ravel()
        t = torch.from_numpy(buf)
        on_gpu = dist.get_backend(group) == "nccl"
        if on_gpu:
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        out = t.cpu().numpy()
        return {i: (out[i, 0].copy(), out[i, 1].copy() if with_errors else None) for i in range(n_containers)}

    def apply_function(self):
        self.stats = {}
        conts = list(self.data)
        if self.stash_valid:
            self.data.representation = self.apply_mode
            for container in conts:
                container["weights"] = self.stashed_hists[container.name].copy()
                if self.bootstrap:
                    container["errors"] = self.stashed_errors[container.name].copy()
            return
        import torch.distributed as dist

        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank() if world > 1 else 0
        owned = self.owned_containers(len(conts), rank, world, sizes=[c.size for c in conts] if world > 1 else None)
        # deferred reweighting chains are materialised on the caller's thread and stream -- with the batched path
        # container by container while the estimators of the earlier ones already run (`weights` callables)
        def event_weights(i):
            conts[i].representation = "events"
            return conts[i].device("weights")

        inputs = {}
        for i in owned:
            st = self._static_sample(conts[i])
            lazy = not self.bootstrap
            inputs[i] = (st, (lambda i=i: event_weights(i)) if lazy else event_weights(i), self._job_kwargs(st))
        results = {}
        if self.bootstrap:
            for i, (st, weights, kw) in inputs.items():
                results[i] = self._bootstrap_map(st, weights, kw)
        else:
            order = list(inputs)
            if order:
                kw = dict(self._job_kwargs(inputs[order[0]][0])) if order else {}
                for key in ("sample", "channels"):
                    kw.pop(key, None)
                samples = [dict(sample=inputs[i][0]["sample"], weights=inputs[i][1], channels=inputs[i][0]["channels"])
                           for i in order]
                maps = kde_hist.kde_histogramdd_batch(samples, stats=self.stats, n_threads=self.kde_workers, **kw)
                for i, m in zip(order, maps):
                    results[i] = (m, None)
        if world > 1:
            results = self.exchange_maps(results, len(conts), int(self.apply_mode.size), self.bootstrap)
        self.data.representation = self.apply_mode
        if self.stash_hists:
            self.stashed_hists, self.stashed_errors = {}, {}
        for i, container in enumerate(conts):
            kde_map = np.ascontiguousarray(np.asarray(results[i][0]).ravel(), dtype=FTYPE)
            container["weights"] = kde_map
            if self.bootstrap:
                kde_errors = np.ascontiguousarray(np.asarray(results[i][1]).ravel(), dtype=FTYPE)
                container["errors"] = kde_errors
            if self.stash_hists:
                self.stashed_hists[container.name] = kde_map.copy()
                if self.bootstrap:
                    self.stashed_errors[container.name] = kde_errors.copy()
        self.stash_valid = self.stash_hists


def service_test_binning():
    """TEST_BINNING of the reference's service tests (pisa_tests/test_services.py:71-75)"""
    from pisa_amd.core.binning import MultiDimBinning, OneDimBinning
    from pisa_amd.core.units import ureg

    return MultiDimBinning([
        OneDimBinning(name="reco_energy", is_log=True, num_bins=3, domain=[0.1, 1] * ureg.GeV),
        OneDimBinning(name="reco_coszen", is_lin=True, num_bins=3, domain=[0.1, 1]),
        OneDimBinning(name="pid", is_lin=True, num_bins=3, domain=[0.1, 1])])

def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    return kde(calc_mode="events", apply_mode=service_test_binning())
