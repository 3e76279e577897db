"""Extended-precision referee of the Poisson LLH scalar (TEST INFRASTRUCTURE: imported by tests/ and by bench.py's oracle
check only, never by pisa_amd/).

The reference's formula (pisa/utils/stats.py:169-253, summed with np.nansum in map.py:1604)

    llh = sum_b  k_b ln(lam_b) - lam_b - (k_b ln(k_b) - k_b)

cancels terms of ~1e5 ... 1e7 per bin to a total of ~ -60 on the headline workload, so two correct fp64 evaluations of it
(glibc's log, the device's log) differ by ulps of the TERMS, which is more than 1e-10 of the TOTAL.  Round 4 accepted
such a difference through a floor of 8 eps sum|terms| -- 180 x the observed difference: not a gate.  The referee
separates the two sources of a difference instead:

  (i)  the MAPS: the formula evaluated in extended precision (np.longdouble: 64-bit mantissa on x86-64; a double-double
       log where longdouble is only 53 bits) on the oracle's summed map and on the device's summed map; the two values
       must agree to the north star's pure 1e-10 relative -- rounding of the evaluation is out of the picture;
  (ii) the EVALUATION: the device's fp64 LLH against the extended value on ITS OWN map, and the oracle's against its
       own: each within 2 eps sum|terms| (every term is built from at most four rounded operations of its magnitude,
       half an ulp each).  The observed deviations are recorded beside the bound and beside the statistical expectation
       eps sqrt(sum terms^2).
"""
import numpy as np

EPS = float(np.finfo(np.float64).eps)


def _log_ext(x):
    """ln(x) for a float64 array in extended precision, returned as np.longdouble.  Where longdouble carries no more
    than 53 bits (non-x86 hosts) a double-double correction is applied: ln x = y + (x e^-y - 1) with y = fp64 log."""
    ld = np.longdouble
    x = np.asarray(x, dtype=np.float64)
    if np.finfo(ld).nmant > 52:
        return np.log(x.astype(ld))
    y = np.log(x)
    # one Newton step of e^y = x in exact-product arithmetic (fma-free Dekker splitting is overkill here: the residual
    # x * exp(-y) - 1 is ~1e-16 and is itself computed to ~1e-16 relative, i.e. the correction is good to ~1e-32)
    return y.astype(ld) + (x * np.exp(-y) - 1.0).astype(ld)


def llh_extended(data, lam):
    """the reference's llh formula in extended precision -> (value as float, sum of |terms| as float, sqrt(sum terms^2) as
    float).  A bin with k = 0 contributes NOTHING: 0 * log(0) is NaN in stats.py:249 and map.py:1604 sums with np.nansum
    (the whole bin, its -lam included, drops out)."""
    ld = np.longdouble
    k = np.asarray(data, dtype=np.float64).ravel()
    lam = np.maximum(np.asarray(lam, dtype=np.float64).ravel(), 1e-10)     # SMALL_POS, stats.py:40
    pos = k > 0
    ln_lam, ln_k = _log_ext(lam), _log_ext(np.where(pos, k, 1.0))
    kl = k.astype(ld)
    t1 = np.where(pos, kl * ln_lam, ld(0.0))
    t2 = np.where(pos, lam.astype(ld), ld(0.0))
    t3 = np.where(pos, kl * ln_k, ld(0.0))
    total = (t1 - t2 - (t3 - kl)).sum()
    mags = np.abs(t1) + t2 + np.abs(t3) + kl
    return float(total), float(mags.sum()), float(np.sqrt((mags * mags).sum()))


def llh_referee(data, lam_device, lam_oracle, llh_device, llh_oracle, rtol=1e-10):
    """see module docstring -> dict with the three comparisons and `met`"""
    e_dev, terms_dev, rms_dev = llh_extended(data, lam_device)
    e_orc, terms_orc, rms_orc = llh_extended(data, lam_oracle)
    bound_dev, bound_orc = 2.0 * EPS * terms_dev, 2.0 * EPS * terms_orc
    maps_rel = abs(e_dev - e_orc) / abs(e_orc)
    out = {
        "extended_precision_bits": int(np.finfo(np.longdouble).nmant) + 1,
        "maps": {"llh_extended_on_device_map": e_dev, "llh_extended_on_oracle_map": e_orc, "rel_diff": maps_rel,
                 "gate": rtol, "met": bool(maps_rel <= rtol)},
        "device_evaluation": {"llh_fp64": float(llh_device), "abs_dev_from_extended": abs(llh_device - e_dev),
                              "bound_2eps_sum_terms": bound_dev, "expected_eps_rms_terms": EPS * rms_dev,
                              "met": bool(abs(llh_device - e_dev) <= bound_dev)},
        "oracle_evaluation": {"llh_fp64": float(llh_oracle), "abs_dev_from_extended": abs(llh_oracle - e_orc),
                              "bound_2eps_sum_terms": bound_orc, "expected_eps_rms_terms": EPS * rms_orc,
                              "met": bool(abs(llh_oracle - e_orc) <= bound_orc)},
        "pure_1e-10_relative_met": bool(abs(llh_device - llh_oracle) <= rtol * abs(llh_oracle)),
        "fp64_abs_diff": abs(llh_device - llh_oracle),
    }
    out["met"] = bool(out["maps"]["met"] and out["device_evaluation"]["met"] and out["oracle_evaluation"]["met"])
    out["applied"] = ("1e-10 relative on the fp64 values" if out["pure_1e-10_relative_met"] else
                      ("referee: maps 1e-10 in extended precision + each fp64 evaluation within 2 eps sum|terms| of its own "
                       "extended value" if out["met"] else "NONE MET"))
    return out
