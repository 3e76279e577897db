"""CPU restatement of the small services around the hot path.  TEST INFRASTRUCTURE ONLY: imported by tests/ and
scripts/dev as the checker, never by pisa_amd/.

PINNED against the reference's own functions executed in the build container (`oracle/gen_golden.py side` /
`stats_wide` -> tests/golden/side_stages_ref.npz, stats_wide_ref.npz; tests/test_oracle_stages.py): find_index /
lookup_indices, two_nu_prob / two_nu_weights, power_law, poly_scale (genie form), decoherence_disappearance /
decoherence_table / tau_row_sq (3-flavour form), metric_wide (five metrics).  `numpy.interp` stands for scipy's
`interp1d(kind='linear')` (checked equal in the same test file).
PARITY UNPINNED (restated from the reference's text, nothing of it executed): decoherence_disappearance_2flav (its
unit conversions run through pint, absent here), csv_hypersurface_scales (pandas table arithmetic of the stage itself),
shift_toward and atm_muon_weights (one-line numpy expressions of resolutions.py / atm_muons.py); the expectations of
the ultrasurfaces and snowstorm_hist tests are written out in those tests the same way."""
import numpy as np


def find_index(val, edges):
    """pisa/core/translation.py:504-553: half-open bins, the last edge inside; -1 below / NaN, n_bins above"""
    edges = np.asarray(edges, dtype=np.float64)
    v = np.asarray(val, dtype=np.float64)
    nb = len(edges) - 1
    with np.errstate(invalid="ignore"):
        idx = np.searchsorted(edges, v, side="right") - 1
        idx = np.where(v == edges[-1], nb - 1, idx)
        idx = np.where(v > edges[-1], nb, idx)
        idx = np.where(v >= edges[0], idx, -1)
    return idx.astype(np.int64)


def lookup_indices(sample, edges):
    """pisa/core/bin_indexing.py:46-101: C-order flat index; any dimension below -> -1, else any above -> n_bins"""
    per_dim = [find_index(x, e) for x, e in zip(sample, edges)]
    nbins = [len(e) - 1 for e in edges]
    under = np.zeros(len(per_dim[0]), dtype=bool)
    over = np.zeros(len(per_dim[0]), dtype=bool)
    flat = np.zeros(len(per_dim[0]), dtype=np.int64)
    for k, nb in zip(per_dim, nbins):
        under |= k == -1
        over |= k == nb
        flat = flat * nb + np.clip(k, 0, nb - 1)
    return np.where(under, -1, np.where(over, int(np.prod(nbins)), flat))


def two_nu_prob(t23, dm31, true_energy, true_coszen):
    """pisa/stages/osc/two_nu_osc.py:101-110"""
    L1 = 19.0
    R = 6378.2 + L1
    phi = np.arcsin((1 - L1 / R) * np.sin(np.arccos(true_coszen)))
    psi = np.arccos(true_coszen) - phi
    propdist = np.sqrt((R - L1) ** 2 + R ** 2 - (2 * (R - L1) * R * np.cos(psi)))
    return t23 * np.sin(1.267 * dm31 * propdist / true_energy) ** 2


def two_nu_weights(nu_flux, t23, dm31, true_energy, true_coszen, flav, weights):
    """pisa/stages/osc/two_nu_osc.py:122-127; flav 0 e, 1 mu, 2 tau (the reference's codes 0, 1, 3)"""
    if flav == 0:
        return weights * nu_flux[:, 0]
    p = two_nu_prob(t23, dm31, true_energy, true_coszen)
    return weights * (nu_flux[:, 1] * ((1.0 - p) if flav == 1 else p))


def power_law(true_energy, pivot, index, norm=1.0, nominal=None):
    """pisa/stages/flux/astrophysical.py:69-72 (nominal=None) and :121-149"""
    scale = np.power(true_energy / pivot, index)
    return norm * scale if nominal is None else norm * nominal * scale


def shift_toward(x, target, fraction, clip=None):
    """pisa/stages/reco/resolutions.py:76-92"""
    out = x + (target - x) * fraction
    return out if clip is None else np.clip(out, clip[0], clip[1])


def poly_scale(params, linear, quad, weights):
    """pisa/stages/xsec/genie_sys.py:103-113"""
    factor = 1
    for p, lin, q in zip(params, linear, quad):
        factor = factor * (1.0 + (lin + q * p) * p)
    return weights * np.maximum(0, factor)


def decoherence_disappearance(coef, gamma, delta, energy, baseline):
    """pisa/stages/osc/decoherence.py:229-269: pairs (1,0), (2,0), (2,1); gamma in GeV, delta in eV^2, E GeV, L km"""
    prob_dec = np.zeros(np.shape(energy))
    for c, g, d in zip(coef, gamma, delta):
        prob_dec += c * (1.0 - np.exp(-g * baseline * 5.07e+18) * np.cos(d * 1.0e-18 / (2.0 * energy) * baseline * 5.07e+18))
    return 2.0 * prob_dec


def decoherence_disappearance_2flav(theta23, gamma32_ev, dm32, energy, baseline):
    """pisa/stages/osc/decoherence.py:112-139 (parity unpinned: the reference converts units through pint, absent here)"""
    norm_term = 0.5 * (np.sin(2.0 * theta23) ** 2)
    decoh_term = np.exp(-gamma32_ev * (baseline * 1000.0 / 1.97e-7))
    osc_term = np.cos((2.0 * 1.27 * dm32 * baseline) / energy)
    return norm_term * (1.0 - (decoh_term * osc_term))


def decoherence_table(disappearance):
    """pisa/stages/osc/decoherence.py:90-106, 449-466: P[n, 3, 3]"""
    n = len(disappearance)
    p = np.zeros((n, 3, 3))
    p[:, 0, 0] = 1.0
    p[:, 1, 1] = 1.0 - disappearance
    p[:, 1, 2] = 1.0 - p[:, 1, 0] - p[:, 1, 1]
    p[:, 2, 1] = p[:, 1, 2]
    p[:, 2, 2] = p[:, 1, 1]
    return p


def tau_row_sq(theta12, theta13, theta23):
    """|U[2][k]|^2 of pisa/stages/osc/decoherence.py:176-227 (its delta_cp phases are 0.0)"""
    c12, c13, c23 = np.cos(theta12), np.cos(theta13), np.cos(theta23)
    s12, s13, s23 = np.sin(theta12), np.sin(theta13), np.sin(theta23)
    eid = 0.0
    row = [(s12 * s23) - (c12 * c23 * s13 * eid), (0.0 - c12 * s23) - (s12 * c23 * s13 * eid), c23 * c13]
    return [abs(v) ** 2 for v in row]


def atm_muon_weights(weights, cr_rw_array, delta_gamma_mu, atm_muon_scale):
    """pisa/stages/background/atm_muons.py:95-101"""
    weight_mod = 1 + (delta_gamma_mu * cr_rw_array)
    return weights * np.clip(weight_mod * atm_muon_scale, a_min=0, a_max=np.inf)


# ---- the metrics beyond llh / poisson_llh / chi2 / mod_chi2 (pisa/utils/stats.py), per bin -------------------------
SMALL_POS = 1e-10


def _poisson_gamma(data, sum_w, sum_w2, a, b=0):
    """pisa/utils/likelihood_functions.py:22-63"""
    from scipy import special

    llh = np.ones(data.shape) * -np.inf
    bad = np.logical_or(sum_w <= 0, sum_w2 < 0)
    llh[np.logical_and(data == 0, bad)] = 0
    good = ~bad
    pois = np.logical_and(sum_w2 == 0, good)
    llh[pois] = data[pois] * np.log(sum_w[pois]) - sum_w[pois] - special.loggamma(data[pois] + 1)
    reg = np.logical_and(good, ~pois)
    alpha = sum_w[reg] ** 2.0 / sum_w2[reg] + a
    beta = sum_w[reg] / sum_w2[reg] + b
    k = data[reg]
    llh[reg] = (alpha * np.log(beta) + special.loggamma(k + alpha).real - special.loggamma(k + 1.0).real
                - (k + alpha) * np.log1p(beta) - special.loggamma(alpha).real)
    return llh


def _conv_poisson(k, l, s, nsigma=3, steps=50):
    """pisa/utils/stats.py:479-527"""
    from scipy.special import gammaln

    l, k, s = max(SMALL_POS, l), max(SMALL_POS, k), max(SMALL_POS, s)
    st = 2 * (steps + 1)
    conv_x = np.linspace(-nsigma * s, +nsigma * s, st)[:-1] + nsigma * s / (st - 1.0)
    conv_y = -np.log(s) - 0.5 * np.log(2 * np.pi) - conv_x ** 2 / (2 * s ** 2)
    f_x = conv_x + l
    idx = np.argmax(f_x > 0)
    f_y = np.nan_to_num(k * np.log(f_x[idx:]) - f_x[idx:] - gammaln(k + 1))
    return np.exp(conv_y[idx:] + f_y).sum() / np.sum(np.exp(conv_y))


def _norm_conv_poisson(k, l, s):
    """pisa/utils/stats.py:529-556"""
    from scipy.special import gammaln

    with np.errstate(all="ignore"):
        n1 = np.exp(l * np.log(l) - l - gammaln(l + 1))
    return _conv_poisson(k, l, s) * n1 / _conv_poisson(l, l, s)


def metric_wide(kind, actual, expected, sigma):
    """per-bin values of stats.correct_chi2 (:697-730), signed_sqrt_mod_chi2 (:762-786), mcllh_mean (:328-382),
    mcllh_eff (:384-438), conv_llh (:558-596)"""
    actual, expected, sigma = (np.array(x, dtype=np.float64).ravel() for x in (actual, expected, sigma))
    if kind == "conv_llh":
        out = []
        for k, l, s in zip(actual, expected, sigma):
            with np.errstate(all="ignore"):
                out.append(np.log(max(SMALL_POS, _norm_conv_poisson(k, l, s))) - np.log(max(SMALL_POS, _norm_conv_poisson(k, k, s))))
        return np.array(out)
    expected = np.clip(expected, SMALL_POS, np.inf)
    if kind == "correct_chi2":
        tv = sigma ** 2 + expected
        return (actual - expected) ** 2 / tv + np.log(tv)
    if kind == "signed_sqrt_mod_chi2":
        return (actual - expected) / np.sqrt(sigma ** 2 + expected)
    return _poisson_gamma(actual, expected, sigma ** 2, a=1 if kind == "mcllh_eff" else 0)


def csv_hypersurface_scales(table, inter_param, inter_value, nominal, values):
    """per-bin scale factors of pisa/stages/discr_sys/csv_hypersurfaces.py:167-206 from the CSV table given as a
    dict of columns (numpy arrays): the slices at the two nodes of `inter_param` around `inter_value`, interpolated
    linearly column by column, then intercept + sum_p gradient_p (value_p - nominal_p); non-finite -> 1.
    (Restated without pandas; parity with the reference anchored on its formulae, not on an execution of it.)"""
    col = np.asarray(table[inter_param], dtype=np.float64)
    if inter_value < col.min() or col.max() < inter_value:
        raise ValueError("outside of interpolation range")
    nodes = np.unique(col)
    lower, upper = nodes[nodes <= inter_value].max(), nodes[nodes > inter_value].min()
    lo, up = col == lower, col == upper
    binlen = upper - lower

    def interpolated(p):
        a, b = np.asarray(table[p], dtype=np.float64)[lo], np.asarray(table[p], dtype=np.float64)[up]
        return (b - a) / binlen * (inter_value - lower) + a

    scales = interpolated("intercept") + sum([interpolated(p) * (values[p] - nominal[p]) for p in values])
    scales = np.array(scales)
    scales[~np.isfinite(scales)] = 1.0
    return scales
