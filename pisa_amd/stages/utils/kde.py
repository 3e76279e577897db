"""KDE-smoothed output maps (counterpart of pisa/stages/utils/kde.py:19-305).

Same constructor kwargs and defaults as the reference (:52-66); log dimensions
are smoothed in ln-space (`linearize_log_dims`, :106-130); the event weights are
whatever the previous stages produced (a deferred reweighting chain is
materialised first).  `stash_hists` memoises the maps (:157-164, 280-293).
Bootstrap errors are not part of this build.
"""
import numpy as np

from pisa_amd import FTYPE
from pisa_amd.core.binning import MultiDimBinning, OneDimBinning
from pisa_amd.core.stage import Stage
from pisa_amd.utils import kde_hist

__all__ = ["kde"]


class kde(Stage):  # pylint: disable=invalid-name
    def __init__(self, bw_method="silverman", coszen_name="reco_coszen", oversample=10,
                 coszen_reflection=0.25, adaptive=True, alpha=0.1, stack_pid=True,
                 stash_hists=False, bootstrap=False, bootstrap_niter=10, bootstrap_seed=None,
                 linearize_log_dims=True, **std_kargs):
        if bootstrap:
            raise NotImplementedError("bootstrap KDE errors are not part of this build")
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
        self.linearize_log_dims = linearize_log_dims
        super().__init__(expected_params=(), expected_container_keys=(),
                         supported_reps={"calc_mode": "events", "apply_mode": MultiDimBinning},
                         **std_kargs)
        self.regularized_apply_mode = None

    def setup_function(self):
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

    def apply_function(self):
        for container in self.data:
            if self.stash_valid:
                self.data.representation = self.apply_mode
                container["weights"] = self.stashed_hists[container.name].copy()
                continue
            sample = []
            for dim, orig in zip(self.regularized_apply_mode, self.apply_mode):
                container.representation = ("log_events" if (orig.is_log and self.linearize_log_dims)
                                            else "events")
                sample.append(container[dim.name])
            container.representation = "events"
            sample = np.stack(sample).T
            weights = container["weights"]
            kde_map = kde_hist.kde_histogramdd(
                sample=sample, binning=self.regularized_apply_mode, weights=weights,
                bw_method=self.bw_method, coszen_name=self.coszen_name,
                coszen_reflection=self.coszen_reflection, adaptive=self.adaptive, alpha=self.alpha,
                oversample=self.oversample, stack_pid=self.stack_pid)
            kde_map = np.ascontiguousarray(kde_map.ravel(), dtype=FTYPE)
            self.data.representation = self.apply_mode
            container["weights"] = kde_map
            if self.stash_hists:
                if self.stashed_hists is None:
                    self.stashed_hists = {}
                self.stashed_hists[container.name] = kde_map.copy()
        self.stash_valid = self.stash_hists
