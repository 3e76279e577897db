"""KDE stage on the GPU vs the CPU oracle + the reference's own invariants.

The KDE core lives in the un-vendored `kde` package: PARITY UNPINNED (see
DESIGN.md).  What is checked: (1) the all-pairs HIP kernel and the cell-list cut-off
estimator (`pisa_hip_kde_create/evaluate`) against the oracle's double loop; (2) the whole map chain (oversampling, coszen reflection, pid
stacking) against the oracle chain on identical inputs; (3) invariants the
reference itself tests (pisa_tests/test_kde_stage.py:148-174, 198-313):
normalisation close to the sum of weights, scale-then-KDE == KDE-then-scale."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_kde_kernel_vs_oracle(oracle):
    from pisa_amd import kernels as K

    rs = np.random.RandomState(1)
    for dim, n, m in ((1, 300, 257), (2, 1500, 1030), (3, 700, 300)):
        src, qry = rs.randn(dim, n), rs.randn(dim, m) * 1.5
        coef, s2 = rs.rand(n), 0.5 + rs.rand(n)
        a = rs.randn(dim, dim)
        inv_cov = a @ a.T + np.eye(dim)
        got = K.kde_eval(K.to_device(src), K.to_device(coef), K.to_device(s2), K.to_device(qry),
                         inv_cov).cpu().numpy()
        want = oracle.kde_eval(src, coef, s2, qry, inv_cov)
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-300)


def _estimator_vs_oracle(oracle, dim, n, m, adaptive, bw, alpha, tol, seed, weighted=True):
    from oracle import kde_oracle
    from pisa_amd import kernels as K

    rs = np.random.RandomState(seed)
    # correlated, non-Gaussian cloud (a coszen-like uniform dimension, a ln E-like skewed one)
    x = np.empty((dim, n))
    x[0] = rs.rand(n) * 2 - 1
    if dim > 1:
        x[1] = 1.5 + rs.gamma(3.0, 0.6, n) + 0.4 * x[0]
    if dim > 2:
        x[2] = rs.randn(n) * 0.3 + 0.2 * x[1]
    w = rs.rand(n) * 2 + 0.1 if weighted else None
    lo, hi = x.min(axis=1, keepdims=True), x.max(axis=1, keepdims=True)
    q = lo + (hi - lo) * (rs.rand(dim, m) * 1.3 - 0.15)       # also outside the cloud
    est = K.KdeEstimator(K.to_device(x), None if w is None else K.to_device(w), bw_method=bw,
                         adaptive=adaptive, alpha=alpha, tol=tol)
    got = est(K.to_device(q)).cpu().numpy()
    want = kde_oracle.gaussian_kde_eval(x, w, q, bw, adaptive, alpha)
    return est, got, want


def test_kde_estimator_cutoff_vs_oracle_large(oracle):
    """the cell-list cut-off estimator (pilot + adaptive evaluation) against the oracle's plain
    double loop at 1.2e5 weighted sources: every density value to 1e-10
    relative (+ an absolute floor of 1e-12 of the peak, the documented truncation bound)"""
    est, got, want = _estimator_vs_oracle(oracle, 2, 120000, 6000, True, "silverman", 0.1, 1e-14, 11)
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-12 * want.max())
    assert want.max() > 0 and np.count_nonzero(want > 1e-6 * want.max()) > 1000
    # the cut-off really removed most of the pairs
    assert est.pairs_pilot < 0.45 * 120000.0 ** 2
    assert 0 < est.pairs_eval < 0.6 * 120000 * 6000
    # local bandwidths themselves (pilot -> lambda): compare s2 in the estimator's order
    ys, coef, s2 = est.arrays()
    assert ys.shape == (2, 120000) and float(s2.min()) > 0
    # sum of the coefficients = sum(wn lam^d)/norm: same as the oracle's chain
    from oracle import kde_oracle  # noqa: F401


@pytest.mark.parametrize("dim,n,m,adaptive,bw,alpha,tol", [
    (1, 30000, 2000, True, "scott", 0.3, 1e-14),
    (3, 20000, 3000, True, "silverman", 0.1, 1e-14),
    (2, 20000, 3000, False, "scott", 0.3, 1e-14),
    (2, 5000, 1000, True, "silverman", 0.5, 0.0),       # no cut-off: one cell, all pairs
    (2, 300, 200, True, "silverman", 0.1, 1e-12),       # tiny: cells nearly empty
    (2, 60000, 3000, True, "scott", 0.3, 1e-12),        # series order 16: the pilot on the matrix cores (one MFMA tile)
    (2, 60000, 3000, True, "silverman", 0.2, 1e-10),    # series order 14: the same kernels, padded to the tile
])
def test_kde_estimator_vs_oracle(oracle, dim, n, m, adaptive, bw, alpha, tol):
    est, got, want = _estimator_vs_oracle(oracle, dim, n, m, adaptive, bw, alpha, tol, 5 + dim)
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=max(tol, 1e-16) * 100 * want.max())
    if tol == 0.0:
        assert est.pairs_pilot == n * n and est.pairs_eval == n * m


def test_kde_hermite_expansion_matches_direct_sums():
    """the 2-D pilot sums dense cells through a truncated Hermite series (fast Gauss transform);
    with the expansion switched off every pair is evaluated.  Both must agree far below the
    parity bar, on the pilot-derived bandwidths and on the final densities."""
    from pisa_amd import _lib
    from pisa_amd import kernels as K

    lib = _lib.lib()
    rs = np.random.RandomState(8)
    n = 150000
    x = np.stack([np.clip(rs.rand(n) * 2 - 1 + rs.randn(n) * 0.15, -1, 1), rs.gamma(4.0, 0.5, n) + 0.6 * rs.rand(n)])
    w = rs.rand(n) * 2 + 0.05
    q = np.array([g.ravel() for g in np.meshgrid(np.linspace(-1.4, 1.4, 90), np.linspace(0.0, 7.0, 80), indexing="ij")])
    xd, wd, qd = K.to_device(x), K.to_device(w), K.to_device(q)
    out = {}
    old = lib.pisa_hip_kde_configure(-1)
    try:
        for tol in (1e-14, 1e-12):
            for flag in (2, 1, 0):
                lib.pisa_hip_kde_configure(flag)
                est = K.KdeEstimator(xd, wd, adaptive=True, alpha=0.3, tol=tol)
                assert (est.n_dense > 100) == bool(flag)
                ys, coef, s2 = est.arrays()
                out[tol, flag] = (s2.cpu().numpy(), est(qd).cpu().numpy(), est.pairs_pilot)
            s2b, fb, wb = out[tol, 0]
            for flag in (2, 1):   # local expansions / Hermite series per target vs every pair
                s2a, fa, wa = out[tol, flag]
                np.testing.assert_allclose(s2a, s2b, rtol=3e-12 if tol < 1e-13 else 3e-11)
                np.testing.assert_allclose(fa, fb, rtol=1e-11 if tol < 1e-13 else 1e-10, atol=1e-13 * fb.max())
                assert wa < 0.2 * wb      # and it is what makes the pilot cheap
            assert out[tol, 2][2] < 0.2 * out[tol, 1][2]
    finally:
        lib.pisa_hip_kde_configure(old)


@pytest.mark.parametrize("alpha,counts,span", [
    (0.1, (120, 80), 1.0),      # a map's shape: oversampled 8 x 8 bins with reflection; strips of 32
    (0.3, (333, 47), 1.0),      # count not a multiple of the strip, strongly varying bandwidths
    (0.3, (41, 300), 3.0),      # coarser lattice in dimension 0: shorter strips
    (0.5, (7, 9), 6.0),         # lattice far coarser than the narrowest kernels: written-out points
])
def test_kde_lattice_evaluation_matches_point_evaluation(alpha, counts, span):
    """`pisa_hip_kde_evaluate_lattice` (Gaussian recurrence along lattice lines) against
    `pisa_hip_kde_evaluate` on the same points written out: far below the parity bar, and far fewer
    instructions.  Also 1-D and 3-D lattices (always written out) and bit-reproducibility."""
    from pisa_amd import kernels as K

    rs = np.random.RandomState(11)
    n = 120000
    x = np.stack([np.clip(rs.rand(n) * 2 - 1 + rs.randn(n) * 0.1, -1, 1), rs.gamma(4.0, 0.5, n) + 0.6 * rs.rand(n)])
    w = rs.rand(n) * 2 + 0.05
    est = K.KdeEstimator(K.to_device(x), K.to_device(w), adaptive=True, alpha=alpha, tol=1e-14)
    n0, n1 = counts
    a0 = np.linspace(-1.45 * span, 1.45 * span, n0)
    a1 = np.linspace(0.2, 6.5, n1)
    origin, step = [a0[0], a1[0]], [(a0[-1] - a0[0]) / (n0 - 1), (a1[-1] - a1[0]) / (n1 - 1)]
    lat = est.evaluate_lattice(origin, step, counts).cpu().numpy()
    pairs_lat = est.pairs_eval
    pts = np.array([g.ravel() for g in np.meshgrid(origin[0] + step[0] * np.arange(n0),
                                                   origin[1] + step[1] * np.arange(n1), indexing="ij")])
    direct = est(K.to_device(pts)).cpu().numpy()
    assert direct.max() > 0
    np.testing.assert_allclose(lat, direct, rtol=2e-12, atol=1e-13 * direct.max())
    assert pairs_lat > 0
    again = est.evaluate_lattice(origin, step, counts).cpu().numpy()
    np.testing.assert_array_equal(lat, again)


def test_kde_lattice_other_dimensions_and_no_cutoff():
    from pisa_amd import kernels as K

    rs = np.random.RandomState(12)
    for dim, n, counts in ((1, 5000, (200,)), (3, 4000, (9, 8, 7))):
        x = rs.randn(dim, n)
        est = K.KdeEstimator(K.to_device(x), None, adaptive=True, alpha=0.3)
        origin, step = [-2.0] * dim, [4.0 / (c - 1) for c in counts]
        lat = est.evaluate_lattice(origin, step, counts).cpu().numpy()
        axes = [origin[d] + step[d] * np.arange(counts[d]) for d in range(dim)]
        pts = np.array([g.ravel() for g in np.meshgrid(*axes, indexing="ij")])
        np.testing.assert_array_equal(lat, est(K.to_device(pts)).cpu().numpy())
    # tol = 0 (every pair): the lattice form is not used, identical bits
    x = rs.randn(2, 3000)
    est = K.KdeEstimator(K.to_device(x), None, adaptive=True, alpha=0.3, tol=0.0)
    lat = est.evaluate_lattice([-2.0, -2.0], [0.1, 0.1], (41, 41)).cpu().numpy()
    pts = np.array([g.ravel() for g in np.meshgrid(-2.0 + 0.1 * np.arange(41), -2.0 + 0.1 * np.arange(41), indexing="ij")])
    np.testing.assert_array_equal(lat, est(K.to_device(pts)).cpu().numpy())
    with pytest.raises(ValueError):
        est.evaluate_lattice([0.0, 0.0], [0.1, 0.1], (0, 5))


def test_kde_estimator_properties():
    """bit-reproducible; unweighted == unit weights; scaling the weights changes nothing (the
    density is normalised); integral over a fine grid = 1"""
    from pisa_amd import kernels as K

    rs = np.random.RandomState(2)
    n = 200000
    x = np.stack([rs.rand(n) * 2 - 1, rs.randn(n) * 0.8 + 3.0])
    w = rs.rand(n) + 0.5
    gx, gy = np.linspace(-1.6, 1.6, 321), np.linspace(-1.5, 7.5, 451)
    q = np.array([g.ravel() for g in np.meshgrid(gx, gy, indexing="ij")])
    xd, wd, qd = K.to_device(x), K.to_device(w), K.to_device(q)
    a = K.KdeEstimator(xd, wd, adaptive=True, alpha=0.1)(qd).cpu().numpy()
    b = K.KdeEstimator(xd, wd, adaptive=True, alpha=0.1)(qd).cpu().numpy()
    np.testing.assert_array_equal(a, b)
    c = K.KdeEstimator(xd, 3.0 * wd, adaptive=True, alpha=0.1)(qd).cpu().numpy()
    np.testing.assert_allclose(c, a, rtol=1e-12, atol=1e-300)
    integral = a.sum() * (gx[1] - gx[0]) * (gy[1] - gy[0])
    assert abs(integral - 1.0) < 2e-3
    u = K.KdeEstimator(xd, None, adaptive=False)(qd).cpu().numpy()
    v = K.KdeEstimator(xd, K.to_device(np.ones(n)), adaptive=False)(qd).cpu().numpy()
    np.testing.assert_allclose(u, v, rtol=1e-13, atol=1e-300)


def test_kde_maps_vs_oracle_and_invariants(oracle):
    from oracle import kde_oracle
    from pisa_amd.core.binning import MultiDimBinning, OneDimBinning
    from pisa_amd.utils import kde_hist

    rs = np.random.RandomState(4)
    n = 6000
    reco_e = 10 ** (0.7 + rs.rand(n) * 1.3)
    reco_cz = np.clip(rs.rand(n) * 2 - 1 + rs.randn(n) * 0.1, -1, 1)
    pid = (rs.rand(n) < 0.35) * 2.0 - 1.0
    w = rs.rand(n) * 3.0
    # regularised binning of example.cfg's reco_binning: ln(E) lin, coszen lin, pid
    e_edges = np.linspace(np.log(5.0), np.log(100.0), 11)
    cz_edges = np.linspace(-1, 1, 11)
    pid_edges = np.array([-1000.0, 0.0, 1000.0])
    binning = MultiDimBinning([OneDimBinning("reco_energy", bin_edges=e_edges),
                               OneDimBinning("reco_coszen", bin_edges=cz_edges),
                               OneDimBinning("pid", bin_edges=pid_edges)])
    sample = np.stack([np.log(reco_e), reco_cz, pid]).T
    kw = dict(bw_method="silverman", adaptive=True, alpha=0.1, coszen_reflection=0.25,
              coszen_name="reco_coszen", oversample=4)
    got = kde_hist.kde_histogramdd(sample=sample, binning=binning, weights=w, stack_pid=True, **kw)
    dims = [("reco_energy", e_edges, False), ("reco_coszen", cz_edges, False), ("pid", pid_edges, False)]
    want = kde_oracle.kde_histogramdd(sample, dims, w, **kw)
    assert got.shape == (10, 10, 2)
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-300)
    # normalisation: events bleed out of the energy range only
    inside = (np.log(reco_e) >= e_edges[0]) & (np.log(reco_e) < e_edges[-1])
    assert abs(got.sum() / w[inside].sum() - 1.0) < 0.1
    # linearity in the weights (test_kde_stage.py: scale-then-KDE == KDE-then-scale)
    got2 = kde_hist.kde_histogramdd(sample=sample, binning=binning, weights=2.5 * w, stack_pid=True, **kw)
    np.testing.assert_allclose(got2, 2.5 * got, rtol=1e-12)
    # non-adaptive, unweighted, 2-D (no pid stacking), Scott factor
    b2 = MultiDimBinning([binning["reco_coszen"], binning["reco_energy"]])
    s2 = np.stack([reco_cz, np.log(reco_e)]).T
    g = kde_hist.kde_histogramdd(sample=s2, binning=b2, weights=None, stack_pid=False, bw_method="scott",
                                 adaptive=False, coszen_name="reco_coszen", oversample=2)
    o = kde_oracle.get_hist(s2, [("reco_coszen", cz_edges, False), ("reco_energy", e_edges, False)], None,
                            "scott", False, 0.3, 0.25, "reco_coszen", 2)
    np.testing.assert_allclose(g, o, rtol=1e-10)


def test_kde_stage_in_pipeline():
    """utils.kde in place of utils.hist (pisa_tests/test_kde_stage.py pattern):
    maps are smooth versions of the histogram maps with nearly the same total"""
    from collections import OrderedDict

    from pisa_amd.core.config_parser import parse_pipeline_config
    from pisa_amd.core.pipeline import Pipeline

    cfg = parse_pipeline_config("settings/pipeline/example_hip.cfg")
    hist_maps = Pipeline(cfg).get_outputs()
    cfg2 = parse_pipeline_config("settings/pipeline/example_hip.cfg")
    cfg3 = OrderedDict()
    for k, v in cfg2.items():
        if k == ("utils", "hist"):
            cfg3[("utils", "kde")] = OrderedDict(calc_mode="events", apply_mode=v["apply_mode"],
                                                 oversample=2, stash_hists=True)
        else:
            cfg3[k] = v
    cfg3["pipeline"]["output_key"] = "weights"
    cfg3[("data", "synthetic_events")]["params"].params.n_events.value = 2.4e4
    pipe = Pipeline(cfg3)
    kde_maps = pipe.get_outputs()
    hist_small = Pipeline(_with_events(parse_pipeline_config("settings/pipeline/example_hip.cfg"), 2.4e4)).get_outputs()
    tot_k = sum(m.hist.sum() for m in kde_maps)
    tot_h = sum(m.hist.sum() for m in hist_small)
    assert abs(tot_k / tot_h - 1) < 0.1
    assert all(np.all(m.hist >= 0) and np.all(np.isfinite(m.hist)) for m in kde_maps)
    # stash: second call returns the memoised maps without recomputing
    again = pipe.get_outputs()
    for a, b in zip(kde_maps, again):
        np.testing.assert_array_equal(a.hist, b.hist)
    assert len(hist_maps) == 12


def _with_events(cfg, n):
    cfg[("data", "synthetic_events")]["params"].params.n_events.value = n
    return cfg


def test_kde_stage_bootstrap():
    """stages/utils/kde.py:189-258: bootstrap_niter resampled KDE maps -> mean map + std errors;
    same seed -> same maps; refuses oversampling (kde_hist.py:67-70)"""
    from collections import OrderedDict

    from pisa_amd.core.config_parser import parse_pipeline_config
    from pisa_amd.core.pipeline import Pipeline
    from pisa_amd.stages.utils.kde import kde

    with pytest.raises(ValueError):
        kde(bootstrap=True, oversample=2, calc_mode="events")

    def make(seed, niter=6):
        cfg = _with_events(parse_pipeline_config("settings/pipeline/example_hip.cfg"), 2.4e4)
        out = OrderedDict()
        for k, v in cfg.items():
            if k == ("utils", "hist"):
                out[("utils", "kde")] = OrderedDict(calc_mode="events", apply_mode=v["apply_mode"],
                                                    oversample=1, bootstrap=True, bootstrap_niter=niter,
                                                    bootstrap_seed=seed)
            else:
                out[k] = v
        out["pipeline"]["output_key"] = ("weights", "errors")
        return Pipeline(out)

    a, b, c = make(3).get_outputs(), make(3).get_outputs(), make(4).get_outputs()
    plain_cfg = _with_events(parse_pipeline_config("settings/pipeline/example_hip.cfg"), 2.4e4)
    for ma, mb, mc in zip(a, b, c):
        np.testing.assert_array_equal(ma.hist, mb.hist)
        np.testing.assert_array_equal(ma.std_devs, mb.std_devs)
        assert np.all(ma.std_devs > 0) and np.all(np.isfinite(ma.std_devs))
        assert not np.array_equal(ma.hist, mc.hist)
        # errors are the bootstrap spread: relative size ~ 1/sqrt(events per bin), well below 1
        sel = ma.hist > 0.05 * ma.hist.max()
        assert np.median(ma.std_devs[sel] / ma.hist[sel]) < 0.5
    del plain_cfg


def test_kde_stage_concurrent_estimators_are_bit_identical_to_sequential():
    """`utils.kde` hands the estimators of one evaluation to the library's thread pool (`kde_workers` threads, each
    with its own stream); every estimator is deterministic by itself, so the maps must not depend on the
    interleaving: repeated concurrent evaluations and a single-thread evaluation give the same bits."""
    from collections import OrderedDict

    from pisa_amd.core.config_parser import parse_pipeline_config
    from pisa_amd.core.pipeline import Pipeline

    cfg = OrderedDict()
    for k, v in parse_pipeline_config("settings/pipeline/example_hip.cfg").items():
        cfg[("utils", "kde") if k == ("utils", "hist") else k] = (
            OrderedDict(calc_mode="events", apply_mode=v["apply_mode"]) if k == ("utils", "hist") else v)
    cfg["pipeline"]["output_key"] = "weights"
    cfg[("data", "synthetic_events")]["params"].params.n_events.value = 4.8e5
    pipe = Pipeline(cfg)
    stage = pipe["kde"]
    assert stage.kde_workers > 1
    runs = []
    for workers in (stage.kde_workers, stage.kde_workers, 1, stage.kde_workers):
        stage.kde_workers = workers
        for s in pipe.stages:
            s.param_hash = None        # evaluate again at the same parameters
        runs.append([m.hist.copy() for m in pipe.get_outputs()])
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            np.testing.assert_array_equal(a, b)
    assert sum(m.sum() for m in runs[0]) > 0


def test_get_hist_bootstrap_through_bootstrap_kde():
    """`get_hist(bootstrap=True)` / `kde_histogramdd(bootstrap=True)` (kde_hist.py:108-109, 155-217: the
    external package's `bootstrap_kde`; parity unpinned, this build's form documented at the class):
    mean map close to the plain KDE map, errors positive and of the bootstrap's size, reproducible,
    oversampling refused, pid stacking handled"""
    from pisa_amd.core.binning import MultiDimBinning, OneDimBinning
    from pisa_amd.utils import kde_hist

    rs = np.random.RandomState(3)
    n = 20000
    cz = rs.rand(n) * 2 - 1
    e = rs.randn(n) * 0.5 + 2.0
    pid = (rs.rand(n) < 0.4).astype(float)
    b2 = MultiDimBinning([OneDimBinning("coszen", domain=[-1, 1], num_bins=10, is_lin=True),
                          OneDimBinning("energy", domain=[0.5, 3.5], num_bins=8, is_lin=True)])
    w = rs.rand(n) + 0.5
    kw = dict(binning=b2, weights=w, bw_method="silverman", coszen_name="coszen", oversample=1, alpha=0.1)
    plain = kde_hist.get_hist(np.stack([cz, e], axis=1), **kw)
    mean, err = kde_hist.get_hist(np.stack([cz, e], axis=1), bootstrap=True, bootstrap_niter=8, **kw)
    mean2, err2 = kde_hist.get_hist(np.stack([cz, e], axis=1), bootstrap=True, bootstrap_niter=8, **kw)
    np.testing.assert_array_equal(mean, mean2)
    np.testing.assert_array_equal(err, err2)
    assert mean.shape == plain.shape == err.shape
    assert np.all(err > 0) and np.all(err < 0.2 * mean.max())
    assert np.abs(mean - plain).max() < 5 * err.max()
    np.testing.assert_allclose(mean.sum(), plain.sum(), rtol=2e-2)
    with pytest.raises(ValueError):
        kde_hist.get_hist(np.stack([cz, e], axis=1), bootstrap=True, **dict(kw, oversample=2))
    b3 = MultiDimBinning(list(b2) + [OneDimBinning("pid", bin_edges=[-0.5, 0.5, 1.5])])
    h3, e3 = kde_hist.kde_histogramdd(np.stack([cz, e, pid], axis=1), b3, weights=w, bw_method="silverman",
                                      coszen_name="coszen", oversample=1, alpha=0.1, stack_pid=True,
                                      bootstrap=True, bootstrap_niter=4)
    assert h3.shape == (10, 8, 2) == e3.shape and np.all(e3 > 0)
    np.testing.assert_allclose(h3.sum(), w.sum() * plain.sum() / w.sum(), rtol=5e-2)


def test_kde_histogramdd_batch_equals_one_by_one():
    """`kde_histogramdd_batch` (all estimators of several samples in one `pisa_hip_kde_lattice_batch` call, on the
    library's own threads and streams) returns the maps of `kde_histogramdd` sample by sample, bit for bit --
    with and without weights, NaN weights zeroed, with and without pid stacking, whatever the thread count"""
    from pisa_amd import kernels as K
    from pisa_amd.core.binning import MultiDimBinning, OneDimBinning
    from pisa_amd.utils import kde_hist

    rs = np.random.RandomState(11)
    b3 = MultiDimBinning([OneDimBinning("energy", domain=[0.5, 3.5], num_bins=6, is_lin=True),
                          OneDimBinning("coszen", domain=[-1, 1], num_bins=8, is_lin=True),
                          OneDimBinning("pid", bin_edges=[0.0, 0.5, 1.0], is_lin=True)])
    samples = []
    for n in (30000, 21000, 45000):
        s = np.stack([rs.randn(n) * 0.6 + 2.0, np.clip(rs.rand(n) * 2.2 - 1.1, -1, 1), (rs.rand(n) < 0.35) * 0.75], axis=1)
        w = rs.rand(n) + 0.05
        w[rs.rand(n) < 0.01] = np.nan
        samples.append(dict(sample=K.to_device(s), weights=K.to_device(w)))
    samples.append(dict(sample=samples[0]["sample"], weights=None))
    kw = dict(bw_method="silverman", adaptive=True, alpha=0.3, coszen_name="coszen", coszen_reflection=0.25, oversample=5)
    for stack_pid, binning in ((True, b3), (False, MultiDimBinning(list(b3)[:2]))):
        smp = [dict(sample=s["sample"] if stack_pid else s["sample"][:, :2].contiguous(), weights=s["weights"]) for s in samples]
        ref = [kde_hist.kde_histogramdd(sample=s["sample"], weights=s["weights"], binning=binning, stack_pid=stack_pid, **kw)
               for s in smp]
        for threads in (1, 3, 8):
            got = kde_hist.kde_histogramdd_batch(smp, binning, stack_pid=stack_pid, n_threads=threads, **kw)
            assert len(got) == len(ref)
            for a, b in zip(got, ref):
                assert a.shape == b.shape and np.all(np.isfinite(a))
                np.testing.assert_array_equal(a, b)
    # (a 3-D binning WITHOUT pid stacking is refused by the one-by-one path and by the batch alike: the evaluation
    # grid of kde_hist.get_hist is two-dimensional, pisa/utils/kde_hist.py:137-143 `megashape`)
    bz = MultiDimBinning([OneDimBinning("energy", domain=[0.5, 3.5], num_bins=4, is_lin=True),
                          OneDimBinning("coszen", domain=[-1, 1], num_bins=6, is_lin=True),
                          OneDimBinning("z", domain=[0.0, 1.0], num_bins=3, is_lin=True)])
    s3 = K.to_device(np.stack([rs.randn(5000) * 0.6 + 2.0, rs.rand(5000) * 2 - 1, rs.rand(5000)], axis=1))
    for call in (lambda: kde_hist.kde_histogramdd(sample=s3, weights=None, binning=bz, stack_pid=False, **kw),
                 lambda: kde_hist.kde_histogramdd_batch([dict(sample=s3, weights=None)] * 2, bz, stack_pid=False, **kw)):
        with pytest.raises((ValueError, AssertionError, RuntimeError)):
            call()
    with pytest.raises(ValueError):
        kde_hist.kde_histogramdd_batch([dict(sample=samples[0]["sample"], weights=samples[1]["weights"])], b3, **kw)


@pytest.mark.parametrize("rho,counts,zero_frac", [(0.85, (37, 11), 0.0), (-0.6, (200, 3), 0.0), (0.0, (301, 203), 0.3),
                                                   (0.4, (64, 64), 0.0), (0.2, (33, 130), 0.0)])
def test_kde_lattice_kernel_edge_shapes(rho, counts, zero_frac):
    """The lattice kernel on shapes that stress its patch / share / pairing logic: strongly correlated samples (a large
    shear between lattice lines in whitened coordinates), lattices smaller than one patch, long thin ones, more lines than
    one row of patches, a third of the weights exactly zero -- always against the point evaluation of the same estimator,
    plus bit-reproducibility."""
    from pisa_amd import kernels as K

    rs = np.random.RandomState(17)
    n = 90000
    z = rs.randn(2, n)
    x = np.stack([z[0], rho * z[0] + np.sqrt(1 - rho * rho) * z[1]]) * np.array([[0.7], [1.3]])
    x[:, : n // 10] *= 2.5                      # a sparse halo: wide local bandwidths
    w = rs.rand(n) + 0.1
    w[rs.rand(n) < zero_frac] = 0.0
    est = K.KdeEstimator(K.to_device(x), K.to_device(w), adaptive=True, alpha=0.3)
    n0, n1 = counts
    a0, a1 = np.linspace(-2.6, 2.3, n0), np.linspace(-3.9, 4.4, n1)
    origin, step = [a0[0], a1[0]], [(a0[-1] - a0[0]) / (n0 - 1), (a1[-1] - a1[0]) / (n1 - 1)]
    lat = est.evaluate_lattice(origin, step, counts).cpu().numpy()
    pts = np.array([g.ravel() for g in np.meshgrid(a0, a1, indexing="ij")])
    direct = est(K.to_device(pts)).cpu().numpy()
    assert direct.max() > 0 and np.all(np.isfinite(lat))
    np.testing.assert_allclose(lat, direct, rtol=2e-12, atol=1e-13 * direct.max())
    np.testing.assert_array_equal(lat, est.evaluate_lattice(origin, step, counts).cpu().numpy())


def test_kde_lattice_short_strips_and_fallback():
    """coarse lattices: the strip length drops from 32 to 16 and 8 points (R da sqrt(max s2) <= 50) and finally the
    points are written out -- every form against the point evaluation"""
    from pisa_amd import kernels as K

    rs = np.random.RandomState(19)
    n = 60000
    x = rs.randn(2, n) * np.array([[1.0], [0.8]])
    est = K.KdeEstimator(K.to_device(x), None, adaptive=True, alpha=0.5)
    for n0 in (400, 60, 30, 16, 9):        # 0.02 ... 1 sample sigma per step: many to less than one kernel width
        a0, a1 = np.linspace(-4.0, 4.0, n0), np.linspace(-3.0, 3.0, 70)
        origin, step = [a0[0], a1[0]], [(a0[-1] - a0[0]) / (n0 - 1), (a1[-1] - a1[0]) / 69]
        lat = est.evaluate_lattice(origin, step, (n0, 70)).cpu().numpy()
        pts = np.array([g.ravel() for g in np.meshgrid(a0, a1, indexing="ij")])
        direct = est(K.to_device(pts)).cpu().numpy()
        np.testing.assert_allclose(lat, direct, rtol=2e-12, atol=1e-13 * direct.max())


def test_kde_weightless_isolated_sources_do_not_poison_the_bandwidths():
    """events of weight zero far from every weighted event have pilot density 0: they are left out of the geometric
    mean of the pilot densities (device and oracle alike) instead of turning every local bandwidth into NaN"""
    from oracle import kde_oracle

    from pisa_amd import kernels as K

    rs = np.random.RandomState(23)
    n = 30000
    x = rs.randn(2, n)
    w = rs.rand(n) + 0.1
    w[rs.rand(n) < 0.3] = 0.0
    x[:, :5] = np.array([[40.0, -35.0, 60.0, 0.0, 0.0], [0.0, 0.0, 0.0, 55.0, -70.0]])   # far outliers ...
    w[:5] = 0.0                                                                           # ... without weight
    pts = np.array([g.ravel() for g in np.meshgrid(np.linspace(-3, 3, 40), np.linspace(-3, 3, 30), indexing="ij")])
    for tol in (K.KDE_DEFAULT_TOL, 0.0):
        est = K.KdeEstimator(K.to_device(x), K.to_device(w), adaptive=True, alpha=0.3, tol=tol)
        got = est(K.to_device(pts)).cpu().numpy()
        assert np.all(np.isfinite(got)) and got.max() > 0
        want = kde_oracle.gaussian_kde_eval(x, w, pts, "silverman", True, 0.3)
        np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-12 * want.max())
    lat = est.evaluate_lattice([-3.0, -3.0], [6.0 / 39, 6.0 / 29], (40, 30)).cpu().numpy()
    np.testing.assert_array_equal(lat, got)   # tol = 0: the lattice is written out


def test_kde_weightless_sources_near_the_cutoff_do_not_move_the_estimate():
    """found by scripts/dev/fuzz_kde.py (round 4): weightless events in the sparse tail of a skewed sample sit 8 ... 38
    bandwidths from their nearest weighted neighbour -- inside the exact sum (pilot tiny but positive), beyond the
    cut-off (pilot exactly 0).  While such events counted in the geometric mean of the pilot densities the local
    bandwidths of ALL events, and the map, depended on the cut-off (per cent level).  Events of weight zero are left
    out of the mean now: the estimate is the same with and without cut-off, equal to the oracle, and equal to the
    estimate of the sample without them up to the sample size in the bandwidth rule."""
    from oracle import kde_oracle

    from pisa_amd import kernels as K

    rs = np.random.RandomState(4)
    n = 6000
    x = rs.gamma(2.0, 0.8, (1, n))
    w = rs.rand(n) * 2 + 0.05
    w[rs.rand(n) < 0.3] = 0.0
    tail = np.argsort(x[0])[-12:]
    x[0, tail] = x[0].max() * np.linspace(1.05, 1.6, 12)       # a sparse far tail ...
    w[tail] = 0.0                                               # ... of weightless events
    q = np.linspace(0.0, 8.0, 400)[None, :]
    res = []
    for tol in (1e-14, 1e-12, 0.0):
        est = K.KdeEstimator(K.to_device(x), K.to_device(w), bw_method="silverman", adaptive=True, alpha=0.35, tol=tol)
        res.append(est(K.to_device(q)).cpu().numpy())
    want = kde_oracle.gaussian_kde_eval(x, w, q, "silverman", True, 0.35)
    for got, tol in zip(res, (1e-14, 1e-12, 0.0)):
        np.testing.assert_allclose(got, want, rtol=1e-10, atol=max(tol, 1e-16) * 100 * want.max())
    # the local bandwidth factors of the weighted events do not know about the weightless ones
    keep = w > 0
    s2_all = K.KdeEstimator(K.to_device(x), K.to_device(w), adaptive=True, alpha=0.35).arrays()[2].cpu().numpy()
    assert np.count_nonzero(s2_all == 1.0) >= np.count_nonzero(~keep)           # they keep the global bandwidth
    lam = np.sqrt(s2_all[s2_all != 1.0])
    assert abs(np.mean(np.log(lam))) < 1e-9                                       # geometric mean of the others' factors: 1


def test_kde_batch_reports_a_failing_job_and_stays_usable():
    """a job the estimator refuses (all weights zero: no finite moments) comes back as an error of the batch call;
    the other jobs of the batch have run, and the library's pool takes the next batch"""
    from pisa_amd import _lib
    from pisa_amd import kernels as K

    rs = np.random.RandomState(29)
    good = K.to_device(rs.randn(2, 25000))
    w_good = K.to_device(rs.rand(25000) + 0.1)
    w_bad = K.to_device(np.zeros(25000))
    origin, step, count = [-2.0, -2.0], [0.1, 0.1], (41, 41)
    with pytest.raises(_lib.PisaHipError):
        K.kde_lattice_batch([(good, w_good, None), (good, w_bad, None)], origin, step, count, n_threads=2)
    dens, sums, _ = K.kde_lattice_batch([(good, w_good, None), (good, None, None)], origin, step, count, n_threads=2)
    ref = K.KdeEstimator(good, w_good, alpha=0.3).evaluate_lattice(origin, step, count)
    np.testing.assert_array_equal(dens[0].cpu().numpy(), ref.cpu().numpy())
    assert sums[1] == 25000.0 and np.isclose(sums[0], float(w_good.sum()), rtol=1e-13)


def test_kde_batch_waits_for_a_range_of_jobs():
    """`pisa_hip_kde_lattice_wait_jobs`: a caller that wants the first jobs' maps does not wait for the rest of the queue:
    the range it names is done when the call returns (status and densities final: equal to the estimator built alone), the
    large job behind it may still run; a failing job inside the range is reported by the range's wait; the batch's own
    `wait` still closes it"""
    from pisa_amd import _lib
    from pisa_amd import kernels as K

    rs = np.random.RandomState(37)
    small = [K.to_device(rs.randn(2, 20000)) for _ in range(3)]
    big = K.to_device(rs.randn(2, 600000))
    origin, step, count = [-2.0, -2.0], [0.1, 0.1], (41, 41)
    b = K.KdeLatticeBatch(5, origin, step, count, big.device, n_threads=3)
    b.submit([(big, None, None)])
    b.submit([(x, None, None) for x in small])
    dens, sums = b.wait_jobs(1, 3)
    got = dens.cpu().numpy()
    for i, x in enumerate(small):
        ref = K.KdeEstimator(x, None, alpha=0.3).evaluate_lattice(origin, step, count)
        np.testing.assert_array_equal(got[i], ref.cpu().numpy())
    assert sums == [20000.0] * 3
    b.submit([(small[0], K.to_device(np.zeros(20000)), None)])     # refused by the estimator: no finite moments
    with pytest.raises(_lib.PisaHipError):
        b.wait_jobs(4, 1)
    dens0, _ = b.wait_jobs(0, 1)
    np.testing.assert_array_equal(dens0.cpu().numpy()[0],
                                  K.KdeEstimator(big, None, alpha=0.3).evaluate_lattice(origin, step, count).cpu().numpy())
    with pytest.raises(_lib.PisaHipError):
        b.wait()


def test_kde_pool_release_returns_the_workspaces():
    """`pisa_hip_kde_pool_release`: the pool threads' grow-only workspaces go back to the device, and the pool works again"""
    import torch

    from pisa_amd import _lib
    from pisa_amd import kernels as K

    rs = np.random.RandomState(31)
    jobs = [(K.to_device(rs.randn(2, 200000)), None, None) for _ in range(4)]
    origin, step, count = [-2.0, -2.0], [0.05, 0.05], (81, 81)
    a = K.kde_lattice_batch(jobs, origin, step, count, n_threads=4)[0].cpu().numpy()
    torch.cuda.synchronize()
    held = torch.cuda.mem_get_info()[0]
    _lib.check(_lib.lib().pisa_hip_kde_pool_release())
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] > held + (50 << 20)      # hundreds of MB come back
    b = K.kde_lattice_batch(jobs, origin, step, count, n_threads=4)[0].cpu().numpy()
    np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("dim,n", [(1, 3000), (2, 20000), (3, 4000)])
@pytest.mark.parametrize("bw", ["silverman", "scott"])
def test_fixed_bandwidth_estimator_equals_scipy(dim, n, bw):
    """The device estimator against scipy.stats.gaussian_kde -- an independent implementation of the same textbook
    estimator for unweighted samples and a fixed bandwidth (tests/test_oracle.py pins the oracle to it as well): every
    density to 1e-10 relative with the documented cut-off floor.  This is the part of the KDE core that HAS a second
    implementation in this image; weights and adaptive bandwidths follow the `kde` package's contract and stay unpinned."""
    from scipy import stats

    from pisa_amd import kernels as K

    rs = np.random.RandomState(100 * dim + len(bw))
    x = np.empty((dim, n))
    x[0] = rs.rand(n) * 2 - 1
    if dim > 1:
        x[1] = 1.5 + rs.gamma(3.0, 0.6, n) + 0.4 * x[0]
    if dim > 2:
        x[2] = rs.randn(n) * 0.3 + 0.2 * x[1]
    lo, hi = x.min(axis=1, keepdims=True), x.max(axis=1, keepdims=True)
    q = lo + (hi - lo) * (rs.rand(dim, 2000) * 1.2 - 0.1)
    est = K.KdeEstimator(K.to_device(x), None, bw_method=bw, adaptive=False, alpha=0.0, tol=1e-14)
    got = est(K.to_device(q)).cpu().numpy()
    want = stats.gaussian_kde(x, bw_method=bw)(q)
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-12 * want.max())
    assert np.count_nonzero(want > 1e-6 * want.max()) > 500
    # weighted samples, scipy given this estimator's factor (the convention the two do not share: number of points
    # against effective sample size): weighted covariance and weighted kernel sums agree
    w = rs.rand(n) * 2 + 0.1
    est_w = K.KdeEstimator(K.to_device(x), K.to_device(w), bw_method=bw, adaptive=False, alpha=0.0, tol=1e-14)
    got_w = est_w(K.to_device(q)).cpu().numpy()
    want_w = stats.gaussian_kde(x, bw_method=est_w.factor, weights=w)(q)
    np.testing.assert_allclose(got_w, want_w, rtol=1e-10, atol=1e-12 * want_w.max())


def test_kde_pinned_by_the_reference_package():
    """With the fixture of `python -m oracle.pin_kde` present (written wherever the un-vendored `kde` package is
    installed): the DEVICE estimator against the package's densities at 1e-10 relative on the reference test's exact
    set-up and three small adaptive cases (skips otherwise -- KDE core parity unpinned)."""
    from pisa_amd import kernels as K
    from tests.test_oracle import _kde_pin_fixture

    z = _kde_pin_fixture()
    for name in sorted({k.split("__")[0] for k in z.files}):
        bw, adaptive, alpha = z[name + "__settings"]
        est = K.KdeEstimator(K.to_device(z[name + "__x"]), K.to_device(z[name + "__w"]), bw_method="silverman" if bw else "scott",
                             adaptive=bool(adaptive), alpha=float(alpha), tol=0.0)
        got = est(K.to_device(z[name + "__points"])).cpu().numpy()
        np.testing.assert_allclose(got, z[name + "__density"], rtol=1e-10, atol=0, err_msg=name)
