"""Extended-precision referee of the Poisson LLH scalar (TEST INFRASTRUCTURE: imported by tests/ and by bench.py's oracle
check only, never by pisa_amd/).

The reference's formula (pisa/utils/stats.py:169-253, summed with np.nansum in map.py:1604)

    llh = sum_b  k_b ln(lam_b) - lam_b - (k_b ln(k_b) - k_b)

cancels terms of ~1e5 ... 1e7 per bin to a total of ~ -60 on the headline workload, so two correct fp64 evaluations of it
(glibc's log, the device's log) differ by ulps of the TERMS, which is more than 1e-10 of the TOTAL.  Round 4 accepted
such a difference through a floor of 8 eps sum|terms| -- 180 x the observed difference: not a gate.  The referee
separates the two sources of a difference instead:

  (i)  the MAPS: the formula evaluated in extended precision (np.longdouble: 64-bit mantissa on x86-64) on the oracle's
       summed map and on the device's summed map; the two values must agree to the north star's pure 1e-10 relative --
       rounding of the evaluation is out of the picture;
  (ii) the EVALUATION: the device's fp64 LLH against the extended value on ITS OWN map, and the oracle's against its
       own: each within 8 eps sqrt(sum terms^2) (round 6).  A fp64 evaluation is a sum of ~4 n_bins terms each carrying
       an independent rounding error of up to ~an ulp of its magnitude (log within 1-2 ulp, a product, a difference): the
       deviations add like a random walk, eps sqrt(sum terms^2) is their scale and 8 of those a bound no correct
       evaluation reaches -- over the 50 parameter points of config C4 at the headline size the largest observed ratio
       |fp64 - extended| / (eps sqrt(sum terms^2)) is recorded by tests/test_gpu_fullsize.py and bench.py (`c4_llh_gate`;
       device <= 0.4, oracle <= 0.6 when this was written).  Round 5's bound, 2 eps sum|terms|, was the worst case of
       every error having the same sign: 45-680 x what is observed, i.e. not a gate.

Where np.longdouble carries no more than fp64's 53 bits (non-x86 hosts) there is no extended arithmetic to referee
with: `llh_referee` then reports `applicable: False` and `met: False` instead of comparing fp64 with fp64 (round-5
advisor: the former double-double fallback computed its Newton residual in plain fp64 and gained no bits).
"""
import numpy as np

EPS = float(np.finfo(np.float64).eps)


EVAL_SIGMAS = 8.0     # evaluation gate: |fp64 - extended| <= EVAL_SIGMAS * eps * sqrt(sum terms^2)


def extended_available():
    """True where np.longdouble has more mantissa bits than float64 (x86-64: 64)"""
    return int(np.finfo(np.longdouble).nmant) > 52


def _log_ext(x):
    """ln(x) for a float64 array as np.longdouble (extended precision where `extended_available()`)"""
    return np.log(np.asarray(x, dtype=np.float64).astype(np.longdouble))


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
    bound_dev, bound_orc = EVAL_SIGMAS * EPS * rms_dev, EVAL_SIGMAS * EPS * rms_orc
    maps_rel = abs(e_dev - e_orc) / abs(e_orc)
    applicable = extended_available()
    out = {
        "extended_precision_bits": int(np.finfo(np.longdouble).nmant) + 1,
        "applicable": applicable,
        "maps": {"llh_extended_on_device_map": e_dev, "llh_extended_on_oracle_map": e_orc, "rel_diff": maps_rel,
                 "gate": rtol, "met": bool(maps_rel <= rtol)},
        "device_evaluation": {"llh_fp64": float(llh_device), "abs_dev_from_extended": abs(llh_device - e_dev),
                              "bound_8eps_rms_terms": bound_dev, "expected_eps_rms_terms": EPS * rms_dev,
                              "worst_case_2eps_sum_terms": 2.0 * EPS * terms_dev,
                              "over_eps_rms": abs(llh_device - e_dev) / (EPS * rms_dev),
                              "met": bool(abs(llh_device - e_dev) <= bound_dev)},
        "oracle_evaluation": {"llh_fp64": float(llh_oracle), "abs_dev_from_extended": abs(llh_oracle - e_orc),
                              "bound_8eps_rms_terms": bound_orc, "expected_eps_rms_terms": EPS * rms_orc,
                              "worst_case_2eps_sum_terms": 2.0 * EPS * terms_orc,
                              "over_eps_rms": abs(llh_oracle - e_orc) / (EPS * rms_orc),
                              "met": bool(abs(llh_oracle - e_orc) <= bound_orc)},
        "pure_1e-10_relative_met": bool(abs(llh_device - llh_oracle) <= rtol * abs(llh_oracle)),
        "fp64_abs_diff": abs(llh_device - llh_oracle),
    }
    out["met"] = bool(applicable and out["maps"]["met"] and out["device_evaluation"]["met"] and out["oracle_evaluation"]["met"])
    out["applied"] = ("1e-10 relative on the fp64 values" if out["pure_1e-10_relative_met"] else
                      ("referee: maps 1e-10 in extended precision + each fp64 evaluation within 8 eps sqrt(sum terms^2) of its "
                       "own extended value" if out["met"] else
                       ("NONE MET" if applicable else "NONE MET (no extended precision on this host: referee not applicable)")))
    return out
