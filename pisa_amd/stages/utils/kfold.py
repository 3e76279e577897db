"""Keep one of `n_splits` folds of the events (counterpart of pisa/stages/utils/kfold.py:36-124): the weights of all
events outside fold `select_split` become 0, those inside stay (or, `renormalize`, are multiplied by n_splits).  The
folds are scikit-learn's `KFold(n_splits, shuffle, random_state=seed)` -- restated here for shuffle=False (the first
n % n_splits folds have one event more, contiguous blocks) and taken from scikit-learn itself, as the reference does,
when the events are shuffled."""
import numpy as np

from pisa_amd import FTYPE
from pisa_amd import kernels as K
from pisa_amd.core.stage import Stage

__all__ = ["kfold"]


def _fold(n, n_splits, select, shuffle, seed):
    if n_splits < 2:
        raise ValueError("k-fold cross-validation requires at least one train/test split by setting n_splits=2 or more,"
                         " got n_splits=%d." % n_splits)
    if n_splits > n:
        raise ValueError("Cannot have number of splits n_splits=%d greater than the number of samples: n_samples=%d."
                         % (n_splits, n))
    if shuffle:
        from sklearn.model_selection import KFold

        for i, (_, test) in enumerate(KFold(n_splits=n_splits, shuffle=True, random_state=seed).split(np.empty(n))):
            if i == select:
                break
        return test
    sizes = np.full(n_splits, n // n_splits, dtype=int)
    sizes[: n % n_splits] += 1
    # kfold.py:96-100: the loop breaks AT the selected split, a `select_split` beyond the last one keeps the last
    i = min(select, n_splits - 1)
    start = int(sizes[:i].sum())
    return np.arange(start, start + sizes[i])


class kfold(Stage):  # pylint: disable=invalid-name
    def __init__(self, n_splits, select_split=0, seed=None, renormalize=False, shuffle=False, save_mask=False,
                 **std_kwargs):
        super().__init__(expected_params=(), expected_container_keys=("weights",),
                         supported_reps={"calc_mode": "events"}, **std_kwargs)
        assert self.calc_mode == "events"
        self.n_splits = int(n_splits)
        self.select_split = int(select_split)
        self.seed = None if seed is None else int(seed)
        self.renormalize = bool(renormalize)
        self.shuffle = bool(shuffle)
        self.save_mask = save_mask

    def setup_function(self):
        if not self.shuffle and self.seed is not None:
            # scikit-learn refuses this combination (KFold.__init__), and so does the reference through it
            raise ValueError("Setting a random_state has no effect since shuffle is False. You should leave random_state"
                             " to its default (None), or set shuffle=True.")
        for container in self.data:
            keep = _fold(container.size, self.n_splits, self.select_split, self.shuffle, self.seed)
            fold_weight = np.zeros(container.size, dtype=FTYPE)
            fold_weight[keep] = self.n_splits if self.renormalize else 1.0
            container["fold_weight"] = fold_weight
            if self.save_mask:
                mask = np.zeros(container.size, dtype=bool)
                mask[keep] = True
                container["kfold_mask"] = mask

    def apply_function(self):
        for container in self.data:
            container["weights"] = K.bin_scale(container.device("weights"), container.device("fold_weight"))


def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    return kfold(n_splits=2, calc_mode="events")
