"""The metrics of the hot path as free functions: the interface of pisa/utils/stats.py (`chi2` :98-167, `llh`
:169-253, `poisson_llh` :255-326, `mcllh_mean` :328-382, `mcllh_eff` :384-438, `conv_llh` :558-596, `mod_chi2`
:651-695, `correct_chi2` :697-730, `signed_sqrt_mod_chi2` :762-786) for code that calls them on arrays rather than
through `Map.metric`.  Each returns the PER-BIN values in the shape of its inputs (the caller sums, `np.nansum` in
`Map.metric`, map.py:1601-1604); the arithmetic is `pisa_hip_metric`'s on the GPU -- there is no host
implementation.  `barlow_llh`, `generalized_poisson_llh` and `weighted_chi2` are not built.

Where the reference reads expected values AND their standard deviations out of an `uncertainties` array
(every metric here but chi2 / llh / poisson_llh), pass a `Map`, or the values and `sigma=`.
"""
import numpy as np

from pisa_amd import FTYPE

__all__ = ["SMALL_POS", "CHI2_METRICS", "LLH_METRICS", "ALL_METRICS", "chi2", "llh", "poisson_llh", "mod_chi2",
           "correct_chi2", "signed_sqrt_mod_chi2", "mcllh_mean", "mcllh_eff", "conv_llh"]

SMALL_POS = 1e-10       # expected values are clipped to [SMALL_POS, inf) before logarithms and divisions (stats.py:74)
CHI2_METRICS = ("chi2", "mod_chi2", "correct_chi2", "signed_sqrt_mod_chi2")
LLH_METRICS = ("llh", "poisson_llh", "conv_llh", "mcllh_mean", "mcllh_eff")
ALL_METRICS = LLH_METRICS + CHI2_METRICS


def _values(x):
    from pisa_amd.core.map import Map

    if isinstance(x, Map):
        return x.hist, x._var
    return np.asarray(x, dtype=FTYPE), None


def _per_bin(kind, actual_values, expected_values, sigma=None):
    from pisa_amd import kernels as K

    a, _ = _values(actual_values)
    e, var = _values(expected_values)
    if a.shape != e.shape:
        raise ValueError("Shape mismatch: actual_values.shape = %s, expected_values.shape = %s" % (a.shape, e.shape))
    if sigma is not None:
        var = np.square(np.asarray(sigma, dtype=FTYPE))
        if var.shape != e.shape:
            raise ValueError("Shape mismatch: sigma.shape = %s, expected_values.shape = %s" % (var.shape, e.shape))
    s2 = None
    if kind in K.VARIANCE_METRICS:
        s2 = K.to_device(np.ascontiguousarray(np.zeros_like(e) if var is None else var).ravel())
    _, per_bin = K.metric(kind, K.to_device(np.ascontiguousarray(a).ravel()), K.to_device(np.ascontiguousarray(e).ravel()),
                          s2, per_bin=True)
    return per_bin.cpu().numpy().reshape(a.shape)


def chi2(actual_values, expected_values):
    """Pearson's (N_actual - N_exp)^2 / N_exp per bin"""
    return _per_bin("chi2", actual_values, expected_values)


def llh(actual_values, expected_values):
    """N_actual ln(N_exp) - N_exp - (N_actual ln(N_actual) - N_actual) per bin (Stirling; NaN where a count is 0)"""
    return _per_bin("llh", actual_values, expected_values)


def poisson_llh(actual_values, expected_values):
    """the Poisson log-probability per bin, with the log-gamma function (stats.py:255-326)"""
    return _per_bin("poisson_llh", actual_values, expected_values)


def mod_chi2(actual_values, expected_values, sigma=None):
    """(N_actual - N_exp)^2 / (sigma^2 + N_exp) per bin; sigma from a `Map`'s errors or given (0 otherwise)"""
    return _per_bin("mod_chi2", actual_values, expected_values, sigma)


def correct_chi2(actual_values, expected_values, sigma=None):
    """(N_actual - N_exp)^2 / (sigma^2 + N_exp) + ln(sigma^2 + N_exp) per bin"""
    return _per_bin("correct_chi2", actual_values, expected_values, sigma)


def signed_sqrt_mod_chi2(actual_values, expected_values, sigma=None):
    """the pull (N_actual - N_exp) / sqrt(sigma^2 + N_exp) per bin"""
    return _per_bin("signed_sqrt_mod_chi2", actual_values, expected_values, sigma)


def mcllh_mean(actual_values, expected_values, sigma=None):
    """L_Mean of JHEP06(2019)030, table 2 (Poisson-gamma mixture, a = 0, b = 0); Poisson where sigma = 0"""
    return _per_bin("mcllh_mean", actual_values, expected_values, sigma)


def mcllh_eff(actual_values, expected_values, sigma=None):
    """L_Eff of JHEP06(2019)030, eq. 3.16 (a = 1, b = 0); Poisson where sigma = 0"""
    return _per_bin("mcllh_eff", actual_values, expected_values, sigma)


def conv_llh(actual_values, expected_values, sigma=None):
    """Poisson smeared with a normal of width sigma (101 steps over +-3 sigma), normalised to the value at
    N_actual = N_exp, minus the same at N_exp := N_actual"""
    return _per_bin("conv_llh", actual_values, expected_values, sigma)
