"""Diagnosis of the reference's one quantitative KDE criterion (TEST INFRASTRUCTURE ONLY, round-3 verdict "Next #2").

pisa_tests/test_kde_stage.py:45-153 builds toy_event_generator(n_events=1000, seed=0, random=False) -> aeff.weight
(livetime 12345 s) -> utils.kde(bw_method="silverman", adaptive=True, alpha=0.1, oversample=1, coszen_reflection=0.25)
on a 15 (log, 10..100 GeV) x 16 (lin, -1..0) binning and asserts that the totals of the map made with and without
`linearize_log_dims` differ, by less than 5 %.  The estimator is the un-vendored `kde` package (setup.py:88,
git+https://github.com/icecubeopensource/kde.git, no version pin) and is a REQUIRED module of the reference's test
runner (pisa_tests/run_unit_tests.py:87), i.e. the real package meets the criterion.

This script restates that set-up on plain numpy (the sample: toy_event_generator.py:75-76 with
RandomState(0); the wrapper: pisa/utils/kde_hist.py:122-217; all weights equal, so every choice that only matters
for unequal weights -- effective sample size in the bandwidth factor, weights in the pilot or in its geometric mean,
weighted vs unweighted covariance -- CANNOT be what separates this build from the package here) and evaluates the
criterion for every variant a Gaussian product-kernel / full-covariance adaptive KDE can differ by.  Result
(`python -m oracle.kde_variants`, table in EXPERIMENTS.md R4-2): at alpha = 0.1 NO structural variant meets the 5 %;
the textbook form (this build, scipy's covariance convention + Abramson-type lambda_i = (pilot_i / geometric
mean)^-alpha entering as lambda^-d * exp(-r^2 / (2 lambda^2))) gives 9.8 %, and the only lever that brings the
number below 5 % is a stronger adaptation (alpha >= 0.26; the package's own default alpha = 0.3 gives 3.5 %).
No family is singled out => nothing to adopt; the KDE core stays PARITY UNPINNED.

Round 5 (verdict "Next #1a", time-boxed): the combinations the first table lacked -- float32 storage of sample and
pilot, pilot on the sample mirrored at the coszen edges, separate exponents for the kernel width (a_w) and for its
normalisation (a_n), and the bandwidth factor scaled.  Result: storage precision and a mirrored pilot do not move the
number (9.8 / 9.5 %); the criterion IS met by several mutually exclusive forms -- a consistent effective alpha of 0.4
(a_w = a_n = 0.4: 0.4 %), an inconsistent normalisation (a_w = 0.1, a_n = 0.2: 1.2 %, not a density), and a bandwidth
factor 0.84 x Silverman's (1.8 %; 0.84 happens to be the ONE-dimensional Silverman rule (3n/4)^(-1/5) over the
two-dimensional one at n = 1000).  The criterion is a weak discriminator: three unrelated changes pass it, nothing in
the reference says which (if any) the package makes.  Closed: parity stays UNPINNED, criterion failing for the
textbook form, no further time goes here.
"""
import numpy as np


def sample():
    rs = np.random.RandomState(0)
    n = 1000
    energy = np.power(10, rs.rand(n) * 3)
    coszen = rs.rand(n) * 2 - 1
    return energy, coszen, np.full(n, 12345.0)


def _kde_eval(x, coef, s2, pts, inv_cov):
    out = np.empty(pts.shape[1])
    for a in range(0, pts.shape[1], 256):
        d = pts[:, None, a:a + 256] - x[:, :, None]
        q = np.einsum("inm,ij,jnm->nm", d, inv_cov, d)
        out[a:a + 256] = (coef[:, None] * np.exp(-0.5 * q * s2[:, None])).sum(0)
    return out


def estimator(x, weights, pts, bw="silverman", alpha=0.1, bias="unbiased", diag=False, self_term=True,
              lam_norm="d", lam_exp=2.0, glob="geometric", alpha_times_d=False, two_pass=False,
              f32=False, mirror_pilot=(), a_w=None, a_n=None, hfac=1.0):
    d, n = x.shape
    if f32:
        x, pts = x.astype(np.float32).astype(np.float64), pts.astype(np.float32).astype(np.float64)
    wn = weights / weights.sum()
    factor = (n * (d + 2) / 4.0) ** (-1.0 / (d + 4)) if bw == "silverman" else n ** (-1.0 / (d + 4))
    factor *= hfac
    xc = x - (x * wn).sum(1, keepdims=True)
    cov = (xc * wn) @ xc.T
    if bias == "unbiased":
        cov /= 1 - np.sum(wn ** 2)
    if diag:
        cov = np.diag(np.diag(cov))
    covh = cov * factor ** 2
    inv = np.linalg.inv(covh)
    norm = np.sqrt(np.linalg.det(2 * np.pi * covh))
    a = alpha * d if alpha_times_d else alpha
    s = np.ones(n)
    if f32 or mirror_pilot or a_w is not None:
        # round-5 forms: pilot over the sample plus its mirror images about coszen = edge (dimension 0), pilot kept
        # in float32, separate exponents of (pilot / g) for the kernel width and for its normalisation
        xa = np.concatenate([x] + [np.vstack([2 * e - x[:1], x[1:]]) for e in mirror_pilot], axis=1)
        wa = np.concatenate([wn] * (1 + len(mirror_pilot)))
        pilot = _kde_eval(xa, wa / norm, np.ones(xa.shape[1]), x, inv)
        if f32:
            pilot = pilot.astype(np.float32).astype(np.float64)
        r = pilot / np.exp(np.mean(np.log(pilot)))
        a_w = a if a_w is None else a_w
        a_n = a_w if a_n is None else a_n
        return _kde_eval(x, wn * (r ** a_n) ** d / norm, (r ** a_w) ** 2, pts, inv)
    for _ in range(2 if two_pass else 1):
        pilot = _kde_eval(x, wn * s ** d / norm, s ** 2, x, inv)
        if not self_term:
            pilot = pilot - wn * s ** d / norm
        g = {"geometric": np.exp(np.mean(np.log(pilot))), "arithmetic": pilot.mean(), "median": np.median(pilot)}[glob]
        s = (pilot / g) ** a                       # 1 / lambda
    coef = wn * {"d": s ** d, "1": s, "0": np.ones(n)}[lam_norm]
    return _kde_eval(x, coef / norm, s ** lam_exp, pts, inv)


def total(linearise, centres="weighted", **kw):
    energy, coszen, w = sample()
    ne, ncz = 15, 16
    if linearise:
        ee = np.linspace(np.log(10), np.log(100), ne + 1)
        ec, xe = 0.5 * (ee[:-1] + ee[1:]), np.log(energy)
    else:
        ee = np.logspace(1, 2, ne + 1)
        ec = np.sqrt(ee[:-1] * ee[1:]) if centres == "weighted" else 0.5 * (ee[:-1] + ee[1:])
        xe = energy
    ce = np.linspace(-1, 0, ncz + 1)
    cc = 0.5 * (ce[:-1] + ce[1:])
    l = int(len(cc) * 0.25)
    c = np.concatenate([2 * cc[0] - cc[1:l + 1][::-1], cc])
    x = np.array([coszen, xe])
    pts = np.array([g.ravel() for g in np.meshgrid(c, ec, indexing="ij")])
    h = estimator(x, w, pts, **kw).reshape(ncz + l, ne)
    h = h[l:] + np.flipud(np.concatenate([np.zeros((ncz - l, ne)), h[:l]]))
    return float((h * np.multiply.outer(np.diff(ce), np.diff(ee))).sum() * w.sum())


VARIANTS = [
    ("this build: scipy covariance convention, lambda = (pilot/geo.mean)^-alpha, lambda^-d exp(-r^2/2lambda^2)", {}),
    ("Scott factor (identical to Silverman for d = 2)", dict(bw="scott")),
    ("biased (1/sum w) covariance", dict(bias="biased")),
    ("diagonal covariance (product kernel)", dict(diag=True)),
    ("pilot without the self term (leave-one-out)", dict(self_term=False)),
    ("normalisation lambda^-1 instead of lambda^-d", dict(lam_norm="1")),
    ("no lambda in the normalisation", dict(lam_norm="0")),
    ("lambda entering the exponent once (h^2 * lambda)", dict(lam_exp=1.0)),
    ("lambda entering the exponent as lambda^4", dict(lam_exp=4.0)),
    ("pilot normalised by its arithmetic mean", dict(glob="arithmetic")),
    ("pilot normalised by its median", dict(glob="median")),
    ("exponent alpha * d (d = 2)", dict(alpha_times_d=True)),
    ("two passes (lambdas from an adaptive pilot)", dict(two_pass=True)),
    ("non-linearised map evaluated at arithmetic bin centres", dict(centres="arithmetic")),
    ("alpha = 0.2", dict(alpha=0.2)),
    ("alpha = 0.26", dict(alpha=0.26)),
    ("alpha = 0.3 (default of the `kde` package and of kde_hist.get_hist)", dict(alpha=0.3)),
    ("alpha = 0.5 (Abramson)", dict(alpha=0.5)),
    ("fixed bandwidth (alpha = 0)", dict(alpha=0.0)),
    # round 5
    ("float32 storage of sample, evaluation points and pilot", dict(f32=True)),
    ("pilot over the sample mirrored at coszen = -1", dict(mirror_pilot=(-1.0,))),
    ("pilot over the sample mirrored at coszen = -1 and +1", dict(mirror_pilot=(-1.0, 1.0))),
    ("width exponent 0.2, normalisation exponent 0.1", dict(a_w=0.2, a_n=0.1)),
    ("width exponent 0.4 (alpha * d with lambda^2 in h^2), normalisation exponent 0.4", dict(a_w=0.4, a_n=0.4)),
    ("width exponent 0.1, normalisation exponent 0.2 (not a density)", dict(a_w=0.1, a_n=0.2)),
    ("bandwidth factor x 0.84 (the 1-D Silverman rule used in 2-D at n = 1000)", dict(hfac=0.84)),
    ("bandwidth factor x 0.7", dict(hfac=0.7)),
    ("bandwidth factor x 1.2", dict(hfac=1.2)),
]


def table():
    energy, coszen, w = sample()
    truth = float(w[(energy >= 10) & (energy < 100) & (coszen >= -1) & (coszen < 0)].sum())
    rows = []
    for name, kw in VARIANTS:
        kw = dict(kw)
        centres = kw.pop("centres", "weighted")
        a, b = total(False, centres=centres, **kw), total(True, **kw)
        rows.append((name, a / b - 1, a / truth - 1, b / truth - 1, abs(a / b - 1) < 0.05 and a != b))
    return rows


if __name__ == "__main__":
    print("| variant (alpha = 0.1 unless stated) | no-lin / lin - 1 | no-lin / histogram - 1 | lin / histogram - 1 | < 5 % |")
    print("|---|---|---|---|---|")
    for name, r, ra, rb, ok in table():
        print("| %s | %+.4f | %+.4f | %+.4f | %s |" % (name, r, ra, rb, "yes" if ok else "no"))
