"""Drop-in boundary tests on the GPU: the reference's `Pipeline(cfg)` /
`Stage` / `Container` protocol driving the HIP kernels, checked against the CPU
oracle.  `settings/pipeline/osc_example.cfg` is the reference's file, unmodified."""
import numpy as np
import pytest

from tests.conftest import PROB3_ATOL, PROB3_RTOL

pytestmark = pytest.mark.gpu
AC = dict(rtol=PROB3_RTOL, atol=PROB3_ATOL)

NAMES = ["nue_cc", "numu_cc", "nutau_cc", "nue_nc", "numu_nc", "nutau_nc",
         "nuebar_cc", "numubar_cc", "nutaubar_cc", "nuebar_nc", "numubar_nc", "nutaubar_nc"]


def _oracle_grid(oracle, binning, theta23_deg=42.0, dm31=2.457e-3, theta13_deg=8.5):
    """P[nubar][iE, jcz, 3, 3] on the calc grid with osc_example.cfg's nominal parameters"""
    e = binning["true_energy"].weighted_centers.m_as("GeV")
    cz = binning["true_coszen"].weighted_centers.magnitude
    prem = np.loadtxt(__import__("pisa_amd.utils.resources", fromlist=["x"]).find_resource("osc/PREM_12layer.dat"))
    lay = oracle.Layers(prem, 2.0, 20.0)
    lay.setElecFrac(0.4656, 0.4656, 0.4957)
    lay.calcLayers(cz)
    mix = oracle.mix_matrix(np.deg2rad(33.48), np.deg2rad(theta13_deg), np.deg2rad(theta23_deg), 0.0)
    dm = oracle.dm_matrix(7.5e-5, dm31)
    mat_pot = np.diag([1.0, 0, 0]).astype(complex)
    zero = np.zeros((3, 3))
    out = {}
    for nubar in (1, -1):
        P = oracle.propagate_array(dm, mix, mat_pot, -1, zero.astype(complex), zero, nubar,
                                   np.repeat(e, len(cz)), np.tile(lay.density, (len(e), 1)),
                                   np.tile(lay.distance, (len(e), 1)))
        out[nubar] = P.reshape(len(e), len(cz), 3, 3)
    return out


def test_osc_example_cfg_unmodified(oracle):
    from pisa_amd.core.pipeline import Pipeline

    pipe = Pipeline("settings/pipeline/osc_example.cfg", profile=True)
    assert pipe.service_names == ["toy_event_generator", "barr_simple", "prob3"]
    maps = pipe.get_outputs()
    assert maps.names == NAMES
    binning = pipe.output_binning
    assert binning.shape == (200, 200)
    ref = _oracle_grid(oracle, binning)
    for m in maps:
        nubar = -1 if "bar" in m.name else 1
        flav = 0 if "nue" in m.name else (1 if "numu" in m.name else 2)
        # nominal toy flux is (0, 1): the map is the P(numu -> nu_flav) oscillogram
        np.testing.assert_allclose(m.hist, ref[nubar][:, :, 1, flav], err_msg=m.name, **AC)
    # SURVEY Appendix A known answers (oracle executed on the reference itself)
    mu = maps["numu_cc"].hist
    np.testing.assert_allclose(mu[0, 0], 0.615974149730982, rtol=1e-10)
    np.testing.assert_allclose(mu[60, 10], 0.22092348875739223, rtol=1e-10)
    np.testing.assert_allclose(maps["nutaubar_cc"].hist[100, 0], 0.8420728703910567, rtol=1e-10)
    # compute memo: unchanged params -> prob3.compute_function is not re-run (stage.py:536-557)
    osc = pipe["prob3"]
    n = len(osc.calc_times)
    pipe.get_outputs()
    assert len(osc.calc_times) == n
    # change a free parameter -> recomputed, new oscillogram
    from pisa_amd.core.units import ureg

    pipe.params.theta23.value = 49.0 * ureg.degree
    maps2 = pipe.get_outputs()
    assert len(osc.calc_times) == n + 1
    ref2 = _oracle_grid(oracle, binning, theta23_deg=49.0)
    np.testing.assert_allclose(maps2["numu_cc"].hist, ref2[1][:, :, 1, 1], **AC)
    assert np.abs(maps2["numu_cc"].hist - mu).max() > 1e-3


def _oracle_event_pipeline(oracle, pipe, flux_params=(1.0, 1.0, 0.0, 0.0, 0.0), theta23_deg=42.3,
                           aeff_scale=1.0, dm31=2.457e-3, theta13_deg=8.5):
    """reference chain on the pipeline's own input columns"""
    grid = _oracle_grid(oracle, pipe["prob3"].calc_mode, theta23_deg=theta23_deg, dm31=dm31, theta13_deg=theta13_deg)
    cm = pipe["prob3"].calc_mode
    lo, hi = cm["true_energy"].domain.m_as("GeV")
    mins, maxs, nb = [np.log(lo), -1.0], [np.log(hi), 1.0], [cm["true_energy"].num_bins, cm["true_coszen"].num_bins]
    ob = pipe.output_binning
    omin = [np.log(5.0), -1.0, -1000.0]
    omax = [np.log(100.0), 1.0, 1000.0]
    onb = list(ob.shape)
    livetime = 2.5 * 365 * 86400.0
    hists, errs = {}, {}
    for c in pipe.data.containers:
        c.representation = "events"
        e, cz = c["true_energy"], c["true_coszen"]
        nubar, flav = c["nubar"], c["flav"]
        flux = oracle.barr_simple(e, cz, c["nu_flux_nominal"], c["nubar_flux_nominal"], nubar, *flux_params)
        P = grid[nubar].reshape(-1, 3, 3)
        pe = oracle.lookup_regular([np.log(e), cz], np.ascontiguousarray(P[:, 0, flav]), mins, maxs, nb)
        pmu = oracle.lookup_regular([np.log(e), cz], np.ascontiguousarray(P[:, 1, flav]), mins, maxs, nb)
        w = oracle.reweight(c["initial_weights"], flux, pe, pmu, c["weighted_aeff"], aeff_scale * livetime)
        sample = [np.log(c["reco_energy"]), c["reco_coszen"], c["pid"]]
        hists[c.name] = oracle.histogram_regular(sample, w, omin, omax, onb).reshape(ob.shape)
        errs[c.name] = np.sqrt(oracle.histogram_regular(sample, w * w, omin, omax, onb)).reshape(ob.shape)
    return hists, errs


def test_event_pipeline_fused_and_unfused(oracle):
    from pisa_amd.core.pipeline import Pipeline
    from pisa_amd.core.units import ureg

    pipe = Pipeline("settings/pipeline/example_hip.cfg")
    hist_stage = pipe["hist"]
    maps = pipe.get_outputs()
    assert hist_stage.fused_last_eval, "the deferred chain reset->osc->aeff must take the fused kernel"
    ref_h, ref_e = _oracle_event_pipeline(oracle, pipe)
    for m in maps:
        np.testing.assert_allclose(m.hist, ref_h[m.name], rtol=1e-11, atol=1e-300, err_msg=m.name)
        np.testing.assert_allclose(m.std_devs, ref_e[m.name], rtol=1e-11, atol=1e-300, err_msg=m.name)
    assert sum(m.hist.sum() for m in maps) > 0

    # free parameters of three different stages change: flux (engine.update_flux),
    # osc (new gather tables), aeff (scale)
    pipe.params.delta_index.value = 0.05 * ureg.dimensionless
    pipe.params.theta23.value = 47.0 * ureg.degree
    pipe.params.aeff_scale.value = 1.3 * ureg.dimensionless
    maps2 = pipe.get_outputs()
    assert hist_stage.fused_last_eval
    ref_h2, ref_e2 = _oracle_event_pipeline(oracle, pipe, flux_params=(1.0, 1.0, 0.05, 0.0, 0.0),
                                            theta23_deg=47.0, aeff_scale=1.3)
    for m in maps2:
        np.testing.assert_allclose(m.hist, ref_h2[m.name], rtol=1e-11, atol=1e-300, err_msg=m.name)
        np.testing.assert_allclose(m.std_devs, ref_e2[m.name], rtol=1e-11, atol=1e-300)

    # stage-by-stage (unfused) execution gives the same maps: touching the
    # weights materialises the deferred chain with the one-stage kernels
    pipe2 = Pipeline("settings/pipeline/example_hip.cfg")
    pipe2.params.delta_index.value = 0.05 * ureg.dimensionless
    pipe2.params.theta23.value = 47.0 * ureg.degree
    pipe2.params.aeff_scale.value = 1.3 * ureg.dimensionless
    pipe2["hist"]._fused = lambda: False
    maps3 = pipe2.get_outputs()
    assert not pipe2["hist"].fused_last_eval
    for a, b in zip(maps2, maps3):
        np.testing.assert_allclose(a.hist, b.hist, rtol=1e-13, atol=1e-300)
        np.testing.assert_allclose(a.std_devs, b.std_devs, rtol=1e-13, atol=1e-300)
    # event-wise weights are visible to host code exactly as the reference leaves them
    c = pipe2.data["numu_cc"]
    c.representation = "events"
    assert c["weights"].shape == c["true_energy"].shape and np.all(np.isfinite(c["weights"]))

    # metric through the Map API (GPU kernel) against the oracle
    total = sum(maps2)
    data = total.fluctuate("poisson", random_state=0)
    got = data.llh(total)
    _, want = oracle.metric("llh", data.hist, total.hist)
    np.testing.assert_allclose(got, want, rtol=1e-10)
    got = data.mod_chi2(total)
    _, want = oracle.metric("mod_chi2", data.hist, total.hist, total.variances)
    np.testing.assert_allclose(got, want, rtol=1e-10)


def test_fit_loop_recovers_injected_parameters():
    """config C4 in miniature: 2 free osc params (theta23, dm31), Asimov data at a
    shifted truth, scipy L-BFGS-B on the [0,1]-rescaled params through
    DistributionMaker + the minimiser callable (analysis.py:2493-2670)."""
    from pisa_amd.analysis.analysis import Analysis
    from pisa_amd.core.distribution_maker import DistributionMaker
    from pisa_amd.core.units import ureg

    dm = DistributionMaker("settings/pipeline/example_hip.cfg")
    for name in dm.params.free.names:
        if name not in ("theta23", "deltam31"):
            dm.params.fix(name)
    assert dm.params.free.names == ("theta23", "deltam31")
    dm.params.theta23.value = 46.5 * ureg.degree
    dm.params.deltam31.value = 2.6e-3 * ureg.eV ** 2
    data = dm.get_outputs(return_sum=True)           # Asimov "data" at the injected truth
    dm.params.reset_free()
    assert dm.params.theta23.value.m_as("deg") == 42.3
    res = Analysis().fit_hypo(data, dm, "mod_chi2")
    assert res.minimizer_metadata["success"], res.minimizer_metadata
    assert res.metric_val < 1e-3 * data[0].hist.sum()
    np.testing.assert_allclose(res.params.theta23.value.m_as("deg"), 46.5, atol=0.15)
    np.testing.assert_allclose(res.params.deltam31.value.m_as("eV**2"), 2.6e-3, rtol=5e-3)
    assert res.num_distributions_generated >= 10 and len(res.fit_history) == res.num_distributions_generated
    assert dm.pipelines[0]["hist"].fused_last_eval


def test_fit_with_reference_minimizer_settings_file():
    """the reference's own minimiser settings file (settings/minimizer/slsqp_*.json, nested
    {"value", "desc"} format) drives the same fit"""
    from pisa_amd.analysis.analysis import Analysis, load_minimizer_settings
    from pisa_amd.core.distribution_maker import DistributionMaker
    from pisa_amd.core.units import ureg

    ms = load_minimizer_settings("settings/minimizer/slsqp_ftol1e-6_eps1e-4_maxiter1000.json")
    assert ms == {"method": "SLSQP", "options": {"ftol": 1.0e-6, "eps": 1.0e-4, "maxiter": 1000}}
    dm = DistributionMaker("settings/pipeline/example_hip.cfg")
    for name in dm.params.free.names:
        if name != "theta23":
            dm.params.fix(name)
    dm.params.theta23.value = 45.2 * ureg.degree
    data = dm.get_outputs(return_sum=True)
    dm.params.reset_free()
    res = Analysis().fit_hypo(data, dm, "chi2",
                              minimizer_settings="settings/minimizer/slsqp_ftol1e-6_eps1e-4_maxiter1000.json")
    assert res.minimizer_metadata["success"], res.minimizer_metadata
    np.testing.assert_allclose(res.params.theta23.value.m_as("deg"), 45.2, atol=0.2)


def _scan(pipe, fast, data, points):
    pipe.fast_path = fast
    pipe._plan = None
    out = []
    from pisa_amd.core.units import ureg

    for t23, dm31, scale, didx in points:
        pipe.params.theta23.value = t23 * ureg.degree
        pipe.params.deltam31.value = dm31 * ureg.eV ** 2
        pipe.params.aeff_scale.value = scale * ureg.dimensionless
        pipe.params.delta_index.value = didx * ureg.dimensionless
        ms = pipe.get_outputs()
        llh = data.metric_total(expected_values=sum(ms), metric="llh")
        chi = data.metric_total(expected_values=sum(ms), metric="mod_chi2")
        out.append((llh, chi, [m.hist.copy() for m in ms], [m.std_devs.copy() for m in ms]))
    return out


def test_fast_plan_replays_the_stage_protocol_bit_for_bit():
    """`Pipeline.get_outputs()` after the first fused evaluation replays three kernel launches
    (core/fastplan.py) instead of running every stage over every container.  Same maps, errors
    and metrics, bit for bit, as the ordinary Stage protocol -- for osc and aeff parameter moves and for
    moves of a flux.barr_simple parameter with the flux per event (replayed as the engine's one-pass
    refresh of the folded flux columns)."""
    from pisa_amd.core.pipeline import Pipeline

    pipe = Pipeline("settings/pipeline/example_hip.cfg")
    data = sum(pipe.get_outputs()).fluctuate("poisson", random_state=0)
    points = [(42.3, 2.457e-3, 1.0, 0.0), (47.0, 2.6e-3, 1.0, 0.0), (47.0, 2.6e-3, 1.4, 0.0),
              (44.0, 2.3e-3, 0.9, 0.0), (44.0, 2.3e-3, 0.9, 0.05), (51.0, 2.5e-3, 0.9, 0.05),
              (51.0, 2.5e-3, 0.9, 0.05), (42.3, 2.457e-3, 1.0, 0.0)]
    slow = _scan(pipe, False, data, points)
    fast = _scan(pipe, True, data, points)
    assert pipe._plan is not None, "the plan must have been built and kept"
    assert pipe._plan._barr_ready, "the flux moves must have been replayed, not sent through the stages"
    for (l0, c0, h0, e0), (l1, c1, h1, e1) in zip(slow, fast):
        assert l0 == l1 and c0 == c1
        for a, b in zip(h0 + e0, h1 + e1):
            np.testing.assert_array_equal(a, b)
    # the metric of a device-backed total is the metric of its host copy
    from pisa_amd.core.units import ureg

    pipe.params.theta23.value = 45.5 * ureg.degree
    ms = pipe.get_outputs()
    total = sum(ms)
    assert total._lazy is not None
    on_device = data.metric_total(expected_values=total, metric="llh")
    assert total._lazy is not None, "the maps must not have travelled for the metric"
    host = sum(m for m in pipe.get_outputs())
    host.hist  # materialise
    assert host._lazy is None
    assert data.metric_total(expected_values=host, metric="llh") == on_device
    # negative data raise as in the reference (stats.py:231-240)
    bad = data * 1.0
    bad._hist[0, 0, 0] = -1.0
    with pytest.raises(ValueError):
        bad.metric_total(expected_values=sum(pipe.get_outputs()), metric="llh")
    # a replayed flux move and the containers: a reader sees the flux of the CURRENT parameters (the
    # bypassed stage recomputes on demand)
    pipe.params.delta_index.value = -0.03 * ureg.dimensionless
    pipe.get_outputs()
    ref = Pipeline("settings/pipeline/example_hip.cfg")
    ref.fast_path = False
    for name in ("theta23", "deltam31", "aeff_scale", "delta_index"):
        ref.params[name].value = pipe.params[name].value
    want = ref.get_outputs()
    a, b = pipe.data["numu_cc"], ref.data["numu_cc"]
    a.representation = b.representation = "events"
    np.testing.assert_array_equal(a["nu_flux"], b["nu_flux"])
    for m, w in zip(pipe.get_outputs(), want):
        np.testing.assert_array_equal(m.hist, w.hist)


def test_fast_plan_outputs_survive_the_next_evaluation_and_containers_stay_truthful():
    from pisa_amd.core.pipeline import Pipeline
    from pisa_amd.core.units import ureg

    pipe = Pipeline("settings/pipeline/example_hip.cfg")
    pipe.get_outputs()
    pipe.params.theta23.value = 46.0 * ureg.degree
    kept = pipe.get_outputs()            # replayed, still on the device
    assert kept[0]._lazy is not None
    pipe.params.theta23.value = 50.0 * ureg.degree
    later = pipe.get_outputs()           # the engine moves on; `kept` must be brought home first
    ref = Pipeline("settings/pipeline/example_hip.cfg")
    ref.fast_path = False
    ref.params.theta23.value = 46.0 * ureg.degree
    want46 = ref.get_outputs()
    ref.params.theta23.value = 50.0 * ureg.degree
    want50 = ref.get_outputs()
    for a, b, c, d in zip(kept, want46, later, want50):
        np.testing.assert_array_equal(a.hist, b.hist)
        np.testing.assert_array_equal(c.hist, d.hist)
    # a reader of pipeline.data sees the CURRENT parameters' arrays, not those of the last
    # evaluation that went through the stages
    c = pipe.data["numu_cc"]
    c.representation = pipe.output_binning
    np.testing.assert_array_equal(c["weights"].reshape(want50["numu_cc"].hist.shape), want50["numu_cc"].hist)
    c.representation = "events"
    c2 = ref.data["numu_cc"]
    c2.representation = "events"
    np.testing.assert_array_equal(c["weights"], c2["weights"])
    # Ye moves the Earth layers: not replayable, falls back, and gives the stages' answer
    for p in (pipe, ref):
        p.params.YeM.value = 0.48 * ureg.dimensionless
    for a, b in zip(pipe.get_outputs(), ref.get_outputs()):
        np.testing.assert_array_equal(a.hist, b.hist)


def test_minimizer_callable_x_to_metric_pins(oracle):
    """captured (x -> metric) pairs of the minimiser callable
    (pisa/analysis/analysis.py:2493-2670): [0,1]-rescaled free parameters -> parameter values
    (param.py:358-400) -> template -> metric + priors penalty (param.py:1372-1396) -> sign.
    The expected numbers come from the CPU oracle chain evaluated at the hand-rescaled values."""
    from pisa_amd.analysis.analysis import Analysis, Counter
    from pisa_amd.core.distribution_maker import DistributionMaker
    from pisa_amd.core.units import ureg

    dm = DistributionMaker("settings/pipeline/example_hip.cfg")
    pipe = dm.pipelines[0]
    names = dm.params.free.names
    assert set(names) == {"delta_index", "theta23", "deltam31", "aeff_scale"}
    ranges = {"theta23": (0.0, 90.0), "deltam31": (0.001, 0.007), "aeff_scale": (0.0, 3.0),
              "delta_index": (-0.5, 0.5)}                      # example_hip.cfg, in the cfg's units
    dm.params.theta23.value = 45.0 * ureg.degree
    dm.params.deltam31.value = 2.5e-3 * ureg.eV ** 2
    data = dm.get_outputs(return_sum=True)                     # Asimov data at this truth
    data_h = data[0].hist.copy()
    dm.reset_free()
    rs = np.random.RandomState(12)
    xs = np.column_stack([0.3 + 0.4 * rs.rand(5), 0.2 + 0.3 * rs.rand(5), 0.2 + 0.5 * rs.rand(5),
                          0.3 + 0.4 * rs.rand(5)])
    ana = Analysis()
    for metric, sign in (("mod_chi2", +1), ("llh", -1), ("chi2", +1)):
        hist, counter = [], Counter()
        for x in xs:
            got = ana._minimizer_callable(x, dm, data, metric, counter, hist)
            val = {n: ranges[n][0] + (ranges[n][1] - ranges[n][0]) * xi for n, xi in zip(names, x)}
            ref_h, ref_e = _oracle_event_pipeline(
                oracle, pipe, flux_params=(1.0, 1.0, val["delta_index"], 0.0, 0.0),
                theta23_deg=val["theta23"], aeff_scale=val["aeff_scale"], dm31=val["deltam31"])
            total = sum(ref_h[n] for n in NAMES)
            var = sum(ref_e[n] ** 2 for n in NAMES)
            _, m = oracle.metric(metric, data_h, total, var)
            prior_llh = -(val["delta_index"] - 0.0) ** 2 / (2 * 0.1 ** 2)   # delta_index = 0.0 +/- 0.1
            penalty = prior_llh if metric == "llh" else -2 * prior_llh
            np.testing.assert_allclose(got, sign * (m + penalty), rtol=1e-10, err_msg="%s %s" % (metric, x))
        assert counter.count == len(xs) and len(hist) == len(xs)
        # fit history rows: [metric_val, *free values] (analysis.py:2636-2646)
        np.testing.assert_allclose(hist[-1][1:], [ranges[n][0] + (ranges[n][1] - ranges[n][0]) * xi
                                                   for n, xi in zip(names, xs[-1])], rtol=1e-14)
    # and the loop went through the replayed evaluation for osc / aeff moves
    assert pipe._plan is not None


def test_hist_stage_binned_calc_mode_transform(oracle, tmp_path):
    """utils.hist with a BINNED calc_mode (pisa/stages/utils/hist.py:69-84, 132-160): the stage
    histograms the events once in the joint (calc grid x output) binning -> `hist_transform`, and
    every evaluation maps weights-per-calc-bin to the output bins: hist = w @ T,
    errors = sqrt(w^2 @ T).  Checked against the oracle's joint histogram and plain numpy."""
    from pisa_amd.core.pipeline import Pipeline
    from pisa_amd.core.units import ureg
    from pisa_amd.utils.resources import find_resource

    text = open(find_resource("settings/pipeline/example_hip.cfg")).read()
    text = text.replace("[osc.prob3]\ncalc_mode = calc_grid\napply_mode = events",
                        "[osc.prob3]\ncalc_mode = calc_grid_coarse\napply_mode = calc_grid_coarse")
    text = text.replace("[aeff.aeff]\napply_mode = events", "[aeff.aeff]\napply_mode = calc_grid_coarse")
    text = text.replace("[utils.hist]\ncalc_mode = events", "[utils.hist]\ncalc_mode = calc_grid_coarse")
    assert text.count("calc_grid_coarse") == 4
    path = tmp_path / "binned.cfg"
    path.write_text(text)
    pipe = Pipeline(str(path))
    cg, ob = pipe["hist"].calc_mode, pipe.output_binning
    assert cg.shape == (50, 50) and ob.shape == (10, 10, 2)
    # stage by stage: the weights per calc bin as they stand in front of the hist stage ...
    for st in pipe.stages[:-1]:
        st.run()
    w_binned = {}
    for c in pipe.data.containers:
        c.representation = cg
        w_binned[c.name] = c["weights"].copy()
        assert w_binned[c.name].shape == (2500,) and np.all(np.isfinite(w_binned[c.name]))
    # ... and the whole pipeline
    maps = pipe.get_outputs()
    assert not pipe["hist"].fused_last_eval
    mins = [np.log(1.0), -1.0, np.log(5.0), -1.0, -1000.0]
    maxs = [np.log(1000.0), 1.0, np.log(100.0), 1.0, 1000.0]
    nb = [50, 50, 10, 10, 2]
    for c in pipe.data.containers:
        c.representation = "events"
        sample = [np.log(c["true_energy"]), c["true_coszen"], np.log(c["reco_energy"]), c["reco_coszen"], c["pid"]]
        T = oracle.histogram_regular(sample, None, mins, maxs, nb).reshape(2500, 200)
        c.representation = cg
        np.testing.assert_array_equal(c["hist_transform"], T)
        w = w_binned[c.name]
        m = maps[c.name]
        np.testing.assert_allclose(m.hist.ravel(), w @ T, rtol=1e-12, atol=1e-300, err_msg=c.name)
        np.testing.assert_allclose(m.std_devs.ravel(), np.sqrt(np.square(w) @ T), rtol=1e-12, atol=1e-300)
        assert m.hist.sum() > 0
    # parameters still move the maps; same run twice gives the same bits
    a = {m.name: m.hist.copy() for m in maps}
    pipe.params.theta23.value = 49.0 * ureg.degree
    b = pipe.get_outputs()
    assert np.abs(b["numu_cc"].hist - a["numu_cc"]).max() > 1e-3 * a["numu_cc"].max()
    pipe.params.theta23.value = 42.3 * ureg.degree
    for m in pipe.get_outputs():
        np.testing.assert_array_equal(m.hist, a[m.name])


def test_octant_fit_finds_the_other_octant():
    """`Analysis.fit_octants` (analysis.py:974-1088, manipulate_params.py:44-123): theta23 is fitted with
    its range confined to either octant and the better fit wins -- a local minimiser started at 42.3
    degrees does not cross the octant degeneracy to an injected 49.5 degrees by itself.  The maker keeps
    its own parameter object, with its original range, at the best-fit value; the result carries the
    losing octant's fit as well."""
    from pisa_amd.analysis.analysis import Analysis
    from pisa_amd.core.distribution_maker import DistributionMaker
    from pisa_amd.core.units import ureg

    dm = DistributionMaker("settings/pipeline/example_hip.cfg")
    for name in dm.params.free.names:
        if name not in ("theta23", "deltam31"):
            dm.params.fix(name)
    theta23 = dm.params.theta23
    rng = [r.m_as("deg") for r in theta23.range]
    theta23.value = 49.5 * ureg.degree
    dm.params.deltam31.value = 2.55e-3 * ureg.eV ** 2
    data = dm.get_outputs(return_sum=True)
    data[0].hist
    ana = Analysis()
    res = ana.fit_octants(data, dm, "mod_chi2", angle="theta23", inflection_point=45.0 * ureg.degree)
    np.testing.assert_allclose(res.params.theta23.value.m_as("deg"), 49.5, atol=0.2)
    np.testing.assert_allclose(res.params.deltam31.value.m_as("eV**2"), 2.55e-3, rtol=5e-3)
    assert res.metric_val < 1e-3 * data[0].hist.sum()
    other = res.alternate_fit
    assert other.params.theta23.value.m_as("deg") <= 45.0 + 1e-9 and other.metric_val > res.metric_val
    # the maker: same Param object, original range, best-fit values
    assert dm.params.theta23 is theta23 and dm.pipelines[0].params.theta23 is theta23
    assert [r.m_as("deg") for r in theta23.range] == rng
    np.testing.assert_allclose(theta23.value.m_as("deg"), res.params.theta23.value.m_as("deg"))
    assert [r.m_as("deg") for r in res.params.theta23.range] == rng
    # a fixed angle: plain fit
    dm.params.fix("theta23")
    plain = ana.fit_octants(data, dm, "mod_chi2")
    assert not hasattr(plain, "alternate_fit")


def test_fit_returns_at_once_when_the_start_matches_the_data():
    """analysis.py:1746-1786: pseudo-data generated at the nominal values -> the template at the starting
    point equals the data -> no minimisation"""
    from pisa_amd.analysis.analysis import Analysis
    from pisa_amd.core.distribution_maker import DistributionMaker

    dm = DistributionMaker("settings/pipeline/example_hip.cfg")
    data = dm.get_outputs(return_sum=True)
    data[0].hist
    res = Analysis().fit_hypo(data, dm, "mod_chi2")
    assert res.num_distributions_generated == 0 and res.minimizer_metadata["nit"] == 0
    assert res.minimizer_metadata["success"] and abs(res.metric_val) < 1e-20
    assert res.params.theta23.value == dm.params.theta23.nominal_value


@pytest.mark.gpu
def test_flux_stage_argument_block_follows_replaced_inputs():
    """`flux.barr_simple` evaluates all containers in one launch with a cached argument block (device
    pointers of its inputs and of the `nu_flux` arrays it owns).  The block is reused while the
    containers hold those very arrays -- `nu_flux` is then rewritten in place -- and rebuilt when an
    input was replaced; either way every container's `nu_flux` equals the single-container kernel on
    the current inputs, bit for bit."""
    import torch

    from pisa_amd import kernels as K
    from pisa_amd.core.pipeline import Pipeline
    from pisa_amd.core.units import ureg

    pipe = Pipeline("settings/pipeline/osc_example.cfg")
    stage = pipe["barr_simple"]
    names = ("nue_numu_ratio", "nu_nubar_ratio", "delta_index", "Barr_uphor_ratio", "Barr_nu_nubar_ratio")

    def check():
        vals = [float(stage.params[n].value.m_as("dimensionless")) for n in names]
        for c in pipe.data.containers:
            c.representation = stage.calc_mode
            want = K.barr_simple(c.device("true_energy"), c.device("true_coszen"), c.device("nu_flux_nominal"),
                                 c.device("nubar_flux_nominal"), c["nubar"], *vals)
            assert bool(torch.isfinite(want).all())
            assert torch.equal(c.device("nu_flux"), want), c.name
            assert np.array_equal(c["nu_flux"], want.cpu().numpy())      # host mirror follows

    pipe.get_outputs()
    check()
    first = [c.current_data["nu_flux"] for c in pipe.data.containers]
    pipe.params.delta_index.value = 0.05 * ureg.dimensionless
    pipe.get_outputs()
    check()
    for c, arr in zip(pipe.data.containers, first):
        c.representation = stage.calc_mode
        assert c.current_data["nu_flux"] is arr                          # rewritten in place
    # an input replaced by a new array: the block is rebuilt at the next evaluation of the stage
    for c in pipe.data.containers:
        c.representation = stage.calc_mode
        c["nu_flux_nominal"] = c["nu_flux_nominal"] * 1.5 + 0.25
        c["nubar_flux_nominal"] = c["nubar_flux_nominal"] * 1.25 + 0.125
    pipe.params.delta_index.value = -0.03 * ureg.dimensionless
    pipe.get_outputs()
    check()


def test_device_backed_outputs_deepcopy_and_pickle_as_host_maps():
    """`deepcopy(maker.get_outputs(return_sum=True))` is a standard pattern with the reference.  From
    the second evaluation on the outputs are device backed (they reference the engine: HBM tensors,
    ctypes argument blocks); copies and pickles must be plain host maps with the same numbers, and
    must not clone the engine"""
    import copy
    import pickle

    from pisa_amd.core.map import Map, MapSet
    from pisa_amd.core.pipeline import Pipeline
    from pisa_amd.core.units import ureg

    pipe = Pipeline("settings/pipeline/example_hip.cfg")
    pipe.get_outputs()
    pipe.params.theta23.value = 46.0 * ureg.degree
    ms = pipe.get_outputs()
    assert ms[0]._lazy is not None
    total = sum(pipe.get_outputs())
    assert total._lazy is not None
    tc = copy.deepcopy(total)
    assert type(tc) is Map and tc._lazy is None
    ms = pipe.get_outputs()
    mc = copy.deepcopy(ms)
    assert type(mc) is MapSet and all(m._lazy is None for m in mc)
    mp = pickle.loads(pickle.dumps(pipe.get_outputs()))
    tp = pickle.loads(pickle.dumps(sum(pipe.get_outputs())))
    ref = Pipeline("settings/pipeline/example_hip.cfg")
    ref.fast_path = False
    ref.params.theta23.value = 46.0 * ureg.degree
    want = ref.get_outputs()
    for a, b, c in zip(mc, mp, want):
        assert a.name == b.name == c.name
        np.testing.assert_array_equal(a.hist, c.hist)
        np.testing.assert_array_equal(b.hist, c.hist)
        np.testing.assert_array_equal(a.std_devs, c.std_devs)
    np.testing.assert_array_equal(tc.hist, sum(want).hist)
    np.testing.assert_array_equal(tp.hist, sum(want).hist)
    # the copy is independent of the engine: the next evaluation does not touch it
    before = tc.hist.copy()
    pipe.params.theta23.value = 51.0 * ureg.degree
    pipe.get_outputs()[0].hist
    np.testing.assert_array_equal(tc.hist, before)


def test_a_replay_that_raises_is_not_replayed_again(monkeypatch):
    """`FastPlan.run` marks parameter changes as seen before it has applied them; if one of its steps
    raises, the plan is dropped and the next evaluation goes through the Stage protocol with fresh
    tables (it must not replay with the stale ones)"""
    from pisa_amd.core.pipeline import Pipeline
    from pisa_amd.core.units import ureg

    pipe = Pipeline("settings/pipeline/example_hip.cfg")
    pipe.get_outputs()
    pipe.params.theta23.value = 44.0 * ureg.degree
    pipe.get_outputs()
    assert pipe._plan is not None
    osc = pipe["prob3"]
    real = osc._matrices

    def boom():
        raise RuntimeError("matrices failed")

    monkeypatch.setattr(osc, "_matrices", boom)
    pipe.params.theta23.value = 49.0 * ureg.degree
    with pytest.raises(RuntimeError):
        pipe.get_outputs()
    assert pipe._plan is None
    monkeypatch.setattr(osc, "_matrices", real)
    got = pipe.get_outputs()           # same parameter values: must be theta23 = 49 deg maps
    ref = Pipeline("settings/pipeline/example_hip.cfg")
    ref.fast_path = False
    ref.params.theta23.value = 49.0 * ureg.degree
    for a, b in zip(got, ref.get_outputs()):
        np.testing.assert_array_equal(a.hist, b.hist)


@pytest.mark.parametrize("settings", [None, "settings/minimizer/slsqp_ftol1e-6_eps1e-4_maxiter1000.json"],
                         ids=["l-bfgs-b", "slsqp"])
def test_fit_with_the_stencil_in_one_sweep_is_the_same_fit(settings):
    """config C4: 2 free parameters (theta23, deltam31).  With `batched_gradient` every iterate's
    finite-difference stencil (n + 1 independent points) goes through `DistributionMaker.metric_many`
    -> `FastPlan.metric_many` -> `HotPathEngine.eval_many` (one sweep of the events); the minimiser sees
    the same function values and the same gradient quotients as when it takes the differences itself,
    so history, iterates and result are identical."""
    from pisa_amd.analysis.analysis import Analysis
    from pisa_amd.core.distribution_maker import DistributionMaker
    from pisa_amd.core.units import ureg

    def make():
        dm = DistributionMaker("settings/pipeline/example_hip.cfg")
        for name in dm.params.free.names:
            if name not in ("theta23", "deltam31"):
                dm.params.fix(name)
        return dm

    dm = make()
    dm.params.theta23.value = 47.5 * ureg.degree
    dm.params.deltam31.value = 2.55e-3 * ureg.eV ** 2
    data = dm.get_outputs(return_sum=True).fluctuate("poisson", random_state=3)
    results = []
    for batched in (False, True):
        dm = make()
        res = Analysis().fit_hypo(data, dm, "llh", minimizer_settings=settings, batched_gradient=batched)
        results.append((res, dm))
    (a, dm_a), (b, dm_b) = results
    assert a.minimizer_metadata["nit"] == b.minimizer_metadata["nit"] and a.minimizer_metadata["nit"] >= 2
    assert a.num_distributions_generated == b.num_distributions_generated
    assert a.fit_history == b.fit_history
    assert a.metric_val == b.metric_val
    for p, q in zip(a.params.free, b.params.free):
        assert p.value == q.value
    # the batched fit did take the sweep, the other one did not
    assert getattr(dm_b.pipelines[0]["hist"]._engine, "last_many", None) is not None
    assert getattr(dm_a.pipelines[0]["hist"]._engine, "last_many", None) is None
    # and the maker is left consistent: outputs at the best fit equal a fresh evaluation there
    fresh = make()
    for p in b.params.free:
        fresh.params[p.name].value = p.value
    np.testing.assert_array_equal(dm_b.get_outputs(return_sum=True)[0].hist, fresh.get_outputs(return_sum=True)[0].hist)


def test_metric_many_matches_point_by_point_and_falls_back():
    """`DistributionMaker.metric_many`: osc + aeff parameters moving -> one sweep; a flux parameter moving
    as well -> point by point; identical numbers either way, priors included"""
    from pisa_amd.core.distribution_maker import DistributionMaker

    dm = DistributionMaker("settings/pipeline/example_hip.cfg")
    keep = ("theta23", "deltam31", "aeff_scale", "delta_index")
    for name in dm.params.free.names:
        if name not in keep:
            dm.params.fix(name)
    data = dm.get_outputs(return_sum=True).fluctuate("poisson", random_state=1)
    dm.get_outputs(return_sum=True)
    assert dm.pipelines[0]._plan is not None
    names = dm.params.free.names
    rs = np.random.RandomState(4)
    x0 = np.array(dm.params.free._rescaled_values)
    pts = [np.clip(x0 + 0.05 * (rs.rand(len(x0)) - 0.5), 0, 1) for _ in range(4)]
    i_flux = names.index("delta_index")
    osc_only = [p.copy() for p in pts]
    for p in osc_only:
        p[i_flux] = x0[i_flux]

    def serial(points):
        out = []
        for x in points:
            dm._set_rescaled_free_params(x)
            hypo = dm.get_outputs(return_sum=True)
            out.append(data.metric_total(expected_values=hypo, metric="mod_chi2") + dm.params.priors_penalty(metric="mod_chi2"))
        return out

    eng = dm.pipelines[0]["hist"]._engine
    want = serial(osc_only)
    assert getattr(eng, "last_many", None) is None
    assert dm.metric_many(osc_only, data, "mod_chi2") == want
    assert eng.last_many is not None
    eng.last_many = None
    want = serial(pts)
    assert dm.metric_many(pts, data, "mod_chi2") == want      # delta_index moves: the flux stage, no sweep
    assert eng.last_many is None
    assert serial(osc_only[:2]) == dm.metric_many(osc_only[:2], data, "mod_chi2")
    # the containers' aeff scales are kept from sweep to sweep: an aeff parameter moved BETWEEN two sweeps (and fixed
    # during them) must still reach the next one
    i_aeff = names.index("aeff_scale")
    same_aeff = [p.copy() for p in osc_only]
    for p in same_aeff:
        p[i_aeff] = x0[i_aeff]
    first = dm.metric_many(same_aeff, data, "mod_chi2")
    assert first == serial(same_aeff)
    for p in same_aeff:
        p[i_aeff] = min(1.0, x0[i_aeff] + 0.07)
    second = dm.metric_many(same_aeff, data, "mod_chi2")
    assert second == serial(same_aeff) and second != first


def test_binned_pipeline_runs_its_weight_chains_in_one_launch():
    """osc_example.cfg applies every stage on 200 x 200 maps: the loader's reset and prob3's reweighting are
    recorded per container and `get_mapset` runs all twelve chains in ONE launch (`pisa_hip_weight_chain_multi`);
    the maps carry the bits of the one-step calls (copy -> pisa_hip_apply_osc_weights), a single container read
    on its own materialises its chain alone with the same bits, and a second evaluation after a parameter
    change does not alias the first one's maps."""
    from pisa_amd import kernels as K
    from pisa_amd.core.pipeline import Pipeline
    from pisa_amd.core.units import ureg
    from pisa_amd.stages import deferred

    pipe = Pipeline("settings/pipeline/osc_example.cfg")
    maps = pipe.get_outputs()
    first = {m.name: m.hist.copy() for m in maps}
    binning = pipe.output_binning
    for c in pipe.data.containers:
        assert not c.pending.get(deferred.KEY)
        c.representation = binning
        w = c.device("initial_weights").clone()
        K.apply_osc_weights(c.device("nu_flux"), c.device_view("prob_e"), c.device_view("prob_mu"), w)
        np.testing.assert_array_equal(first[c.name].ravel(), w.cpu().numpy())
    # one container read on its own, before anybody asks for the map set: its chain alone, same bits
    pipe.params.theta23.value = 47.0 * ureg.degree
    pipe.run()
    c = pipe.data["numu_cc"]
    c.representation = binning
    assert deferred.chain_open(c)
    alone = c["weights"].copy()
    assert not c.pending.get(deferred.KEY)
    maps2 = pipe.get_outputs()
    np.testing.assert_array_equal(maps2["numu_cc"].hist.ravel(), alone)
    assert np.abs(maps2["numu_cc"].hist - first["numu_cc"]).max() > 1e-3
    np.testing.assert_array_equal(maps["nue_cc"].hist, first["nue_cc"])      # the first evaluation's maps are intact


def test_stage_protocol_outputs_are_lazy_rows_of_the_device_table():
    """With the plan switched off every evaluation runs the Stage protocol in full; utils.hist then publishes
    ROWS of the device map table (`BlockRow`) instead of copying maps: `get_outputs()` hands out device-backed
    Maps (sum and metric on the device), reading a container gives the very numbers, in-place edits +
    `mark_changed` behave as on any array, Maps somebody keeps survive the next evaluation, and rows nobody read
    are void afterwards (a clear error, not stale numbers)."""
    import sys

    from tests.conftest import ROOT

    sys.path.insert(0, ROOT)
    import bench
    from pisa_amd.core.container import BlockRow
    from pisa_amd.core.pipeline import Pipeline
    from pisa_amd.core.units import ureg

    pipe = Pipeline(bench._pipeline_cfg(120000))
    pipe.fast_path = False
    maps = pipe.get_outputs()
    assert all(m._lazy is not None for m in maps.maps)            # nothing has travelled yet
    total = sum(maps)
    data = total.fluctuate("poisson", random_state=0)
    llh_dev = data.metric_total(expected_values=total, metric="llh")
    host = {m.name: (m.hist.copy(), m.std_devs.copy()) for m in maps}         # fetches the block
    ref_total = sum(h for h, _ in host.values())
    np.testing.assert_array_equal(total.hist, ref_total)
    from pisa_amd.core.map import Map

    plain = Map("t", ref_total, total.binning)
    # the tail kernel's value == the generic metric kernel's on the host total (same sums in the same order)
    assert llh_dev == data.metric_total(expected_values=plain, metric="llh")
    # the containers hold rows of the same table
    binning = pipe.output_binning
    c = pipe.data["numu_cc"]
    c.representation = binning
    assert type(c.current_data["weights"]) is BlockRow
    np.testing.assert_array_equal(c["weights"].reshape(binning.shape), host["numu_cc"][0])
    np.testing.assert_array_equal(c["errors"].reshape(binning.shape), host["numu_cc"][1])
    w = c["weights"]
    w *= 2.0
    c.mark_changed("weights")
    np.testing.assert_array_equal(c.device("weights").cpu().numpy().reshape(binning.shape), 2.0 * host["numu_cc"][0])
    # next evaluation: the kept Maps keep their numbers, the containers get new rows
    keep = maps["nue_cc"]
    pipe.params.theta23.value = 48.0 * ureg.degree
    row_before = pipe.data["nue_cc"].current_data["weights"]
    maps2 = pipe.get_outputs()
    np.testing.assert_array_equal(keep.hist, host["nue_cc"][0])
    assert np.abs(maps2["numu_cc"].hist - host["numu_cc"][0]).max() > 0
    c2 = pipe.data["nue_cc"]
    c2.representation = binning
    assert c2.current_data["weights"] is not row_before
    # a row of the evaluation before last that nobody read: void
    pipe.params.theta23.value = 49.0 * ureg.degree
    stale = c2.current_data["weights"]
    assert stale.pristine
    pipe.get_outputs()
    with pytest.raises(RuntimeError):
        stale.get_host()


def test_metric_many_without_errors_uses_zero_variance_for_mod_chi2():
    """round-3 advisor finding: with `output_key = weights` the maps carry no errors and `Map.metric` gives
    mod_chi2 a zero variance; the one-sweep tail would hand it the histogram's sumw2.  `FastPlan.metric_many`
    therefore refuses mod_chi2 there (point by point, the serial values), and still takes llh in one sweep;
    an unknown metric name goes point by point as well (and raises there, as the reference does)."""
    import sys

    from tests.conftest import ROOT

    sys.path.insert(0, ROOT)
    import bench
    from pisa_amd.core.distribution_maker import DistributionMaker

    cfg = bench._pipeline_cfg(120000)
    cfg["pipeline"]["output_key"] = "weights"
    dm = DistributionMaker([cfg])
    for name in dm.params.free.names:
        if name not in ("theta23", "deltam31"):
            dm.params.fix(name)
    data = dm.get_outputs(return_sum=True).fluctuate("poisson", random_state=1)
    dm.get_outputs(return_sum=True)
    plan = dm.pipelines[0]._plan
    assert plan is not None and not plan.with_errors
    rs = np.random.RandomState(4)
    x0 = np.array(dm.params.free._rescaled_values)
    pts = [np.clip(x0 + 0.05 * (rs.rand(len(x0)) - 0.5), 0, 1) for _ in range(3)]

    def serial(points, metric):
        out = []
        for x in points:
            dm._set_rescaled_free_params(x)
            out.append(data.metric_total(expected_values=dm.get_outputs(return_sum=True), metric=metric)
                       + dm.params.priors_penalty(metric=metric))
        return out

    eng = dm.pipelines[0]["hist"]._engine
    eng.last_many = None
    want = serial(pts, "mod_chi2")
    assert dm.metric_many(pts, data, "mod_chi2") == want
    assert eng.last_many is None                      # no sweep: the guard sent it point by point
    # zero variance really is what the serial path uses: the sweep's value (sigma^2 = sumw2) would differ
    hypo = dm.get_outputs(return_sum=True)
    hypo = hypo.maps[0] if hasattr(hypo, "maps") else hypo
    assert not np.any(hypo.variances)
    assert dm.metric_many(pts, data, "llh") == serial(pts, "llh")
    assert eng.last_many is not None                  # llh: one sweep
    with pytest.raises(ValueError):
        dm.metric_many(pts, data, "no_such_metric")


def test_var_binning_pipeline_one_mapset_per_selection():
    """A pipeline whose output binning is a `VarBinning` (pisa/core/pipeline.py:389-451, 686-763; the reference's
    own checks at :920-958): one MapSet per selection, each the weighted histogram of that selection's events in that
    selection's binning with errors sqrt(sum w^2) -- compared with numpy's histogram of the host copies --, a
    binned apply_mode refused, overlapping cuts refused, an empty selection accepted."""
    from pisa_amd.core.binning import MultiDimBinning, VarBinning
    from pisa_amd.core.map import MapSet
    from pisa_amd.core.pipeline import Pipeline
    from pisa_amd.core.units import ureg

    p = Pipeline("settings/pipeline/varbin_example_hip.cfg")
    vb = p.output_binning
    assert isinstance(vb, VarBinning) and vb.nselections == 2
    out = p.get_outputs()
    assert len(out) == 2 and all(isinstance(ms, MapSet) for ms in out)

    def expected(container, binning, keep):
        # the translation's histogram is fast_histogram's: the range is half open, an event ON the last edge (the
        # synthetic reco_coszen is clipped to 1) is outside (translation.py:148-166), where numpy would count it
        keep = keep.copy()
        for d in binning:
            keep &= container[d.name] < d.edge_magnitudes[-1]
        sample = [container[d.name][keep] for d in binning]
        edges = [d.edge_magnitudes for d in binning]
        w = container["weights"][keep]
        return np.histogramdd(sample, bins=edges, weights=w)[0], np.sqrt(np.histogramdd(sample, bins=edges, weights=w * w)[0])

    p.data.representation = "events"
    edges = vb.selections.edge_magnitudes
    total = 0.0
    for i, (ms, binning) in enumerate(zip(out, vb.binnings)):
        assert [m.name for m in ms] == [c.name for c in p.data] and all(m.binning == binning for m in ms)
        for m, c in zip(ms, p.data):
            keep = (c["pid"] >= edges[i]) & (c["pid"] < edges[i + 1])
            assert 0 < keep.sum() < c.size
            h, e = expected(c, binning, keep)
            np.testing.assert_allclose(m.nominal_values, h, rtol=1e-12, atol=1e-300)
            np.testing.assert_allclose(m.std_devs, e, rtol=1e-12, atol=1e-300)
            total += m.nominal_values.sum()
    # the two selections partition the events inside the reco binning
    whole = sum(c["weights"][(c["reco_energy"] >= 5) & (c["reco_energy"] < 100) & (c["reco_coszen"] < 1)].sum() for c in p.data)
    np.testing.assert_allclose(total, whole, rtol=1e-10)

    # new parameters move the maps (the selected rows are kept, the weights are gathered again)
    before = out[1][3].nominal_values.copy()
    p.params.theta23.value = 38.0 * ureg.deg
    again = p.get_outputs()
    assert np.abs(again[1][3].nominal_values - before).max() > 0
    c, keep = p.data.containers[3], None
    p.data.representation = "events"
    keep = (c["pid"] >= edges[1]) & (c["pid"] < edges[2])
    np.testing.assert_allclose(again[1][3].nominal_values, expected(c, vb.binnings[1], keep)[0], rtol=1e-12, atol=1e-300)

    # selections by cut expressions, given from outside; a single output key: no errors
    by_cuts = VarBinning(binnings=vb.binnings, selections=["(true_coszen > 0) & (pid > 0)", "true_coszen <= 0"])
    maps = p.get_outputs(output_binning=by_cuts, output_key="weights")
    c = p.data.containers[0]
    p.data.representation = "events"
    np.testing.assert_allclose(maps[0][0].nominal_values,
                               expected(c, vb.binnings[0], (c["true_coszen"] > 0) & (c["pid"] > 0))[0], rtol=1e-12, atol=1e-300)
    assert not np.any(maps[0][0].std_devs)

    # a stage that applies to a binning cannot feed a variable binning
    osc = p["prob3"]
    assert isinstance(osc.calc_mode, MultiDimBinning)
    osc.apply_mode = osc.calc_mode
    with pytest.raises(ValueError, match="apply_mode='events'"):
        p.get_outputs()
    osc.apply_mode = "events"
    p.get_outputs()
    # overlapping selections are refused when they are set, empty ones are not
    with pytest.raises(ValueError, match="not mutually exclusive"):
        p.output_binning = VarBinning(binnings=vb.binnings, selections=["pid > 0"] * 2)
    assert p.output_binning is vb
    with pytest.raises(ValueError, match="not mutually exclusive"):
        p.get_outputs(output_binning=VarBinning(binnings=vb.binnings, selections=["pid > 0"] * 2))
    p.output_binning = VarBinning(binnings=vb.binnings, selections=["pid > 0", "pid > np.inf"])
    out = p.get_outputs()
    assert out[0][0].nominal_values.sum() > 0 and all(m.nominal_values.sum() == 0 for m in out[1])


@pytest.mark.parametrize("cfg", ["example.cfg", "fast_example.cfg"])
def test_reference_example_cfgs_unmodified(oracle, cfg):
    """The reference's own `settings/pipeline/example.cfg` / `fast_example.cfg`, unmodified, on the reference's own toy
    events file (read by this package's HDF5 reader, `utils/hdf.py`; the `data.simple_data_loader` service): two
    selector dimensions (`nh`, `earth`), cuts, the legacy "oppo" flux columns, 12 x 100 events -- every output map
    against the oracle's restatement of the reference chain on the same columns."""
    from pisa_amd.core.pipeline import Pipeline
    from pisa_amd.core.units import ureg

    pipe = Pipeline("settings/pipeline/" + cfg)
    assert pipe.service_names[0] == "simple_data_loader" and sorted(pipe.param_selections) == ["earth", "nh"]
    maps = pipe.get_outputs()
    pipe.data.representation = "events"
    assert len(maps) == 12 and {c.size for c in pipe.data} == {100}
    c = pipe.data["numubar_nc"]
    assert c["nu_flux_nominal"].shape == (100, 2) and -1 <= c["true_coszen"].min() and c["true_energy"].max() <= 80
    # the "oppo" columns of the file are the fluxes of the other sign (events_pi.py:74-85)
    assert not np.array_equal(c["nu_flux_nominal"], c["nubar_flux_nominal"])
    ref_h, ref_e = _oracle_event_pipeline(oracle, pipe)
    for m in maps:
        np.testing.assert_allclose(m.hist, ref_h[m.name], rtol=1e-11, atol=1e-300, err_msg=m.name)
        np.testing.assert_allclose(m.std_devs, ref_e[m.name], rtol=1e-11, atol=1e-300, err_msg=m.name)
    assert sum(m.hist.sum() for m in maps) > 0
    # parameters of three stages move; the other Earth composition is selected and changes the maps
    pipe.params.delta_index.value = 0.05 * ureg.dimensionless
    pipe.params.theta23.value = 47.0 * ureg.degree
    pipe.params.aeff_scale.value = 1.3 * ureg.dimensionless
    maps2 = pipe.get_outputs()
    ref_h2, ref_e2 = _oracle_event_pipeline(oracle, pipe, flux_params=(1.0, 1.0, 0.05, 0.0, 0.0), theta23_deg=47.0,
                                            aeff_scale=1.3)
    for m in maps2:
        np.testing.assert_allclose(m.hist, ref_h2[m.name], rtol=1e-11, atol=1e-300, err_msg=m.name)
        np.testing.assert_allclose(m.std_devs, ref_e2[m.name], rtol=1e-11, atol=1e-300)
    before = sum(maps2).hist.copy()
    pipe.select_params("lead")
    assert pipe.params.YeI.value == 0.398
    assert np.abs(sum(pipe.get_outputs()).hist - before).max() > 0


def test_reference_varbin_example_cfg_unmodified():
    """`settings/pipeline/varbin_example.cfg` as it is: a MapSet per pid selection of the toy events"""
    from pisa_amd.core.pipeline import Pipeline

    p = Pipeline("settings/pipeline/varbin_example.cfg")
    out = p.get_outputs()
    assert len(out) == 2 and [ms[0].hist.shape for ms in out] == [(10, 10), (10, 20)]
    p.data.representation = "events"
    kept = sum(float(c["weights"][(c["reco_energy"] >= 5) & (c["reco_energy"] < 100) & (c["reco_coszen"] < 1) & (c["pid"] < 1000)].sum())
               for c in p.data)
    np.testing.assert_allclose(sum(m.hist.sum() for ms in out for m in ms), kept, rtol=1e-10)
    assert all(np.all(m.std_devs[m.hist > 0] > 0) for ms in out for m in ms)


def test_pipeline_with_correlated_priors():
    """`Pipeline.add_covariance` (pisa/core/pipeline.py:485-536): two parameters of `example.cfg` get correlated
    priors; the pipeline then runs on the derived values, evaluation after evaluation (plan replays included), exactly
    as a plain pipeline set to the same values does."""
    from pisa_amd.core.param import DerivedParam
    from pisa_amd.core.pipeline import Pipeline

    pipe, plain = Pipeline("settings/pipeline/example.cfg"), Pipeline("settings/pipeline/example.cfg")
    cov = {"aeff_scale": {"aeff_scale": 0.04, "nu_nc_norm": 0.01}, "nu_nc_norm": {"aeff_scale": 0.01, "nu_nc_norm": 0.04}}
    assert pipe.add_covariance(cov)
    with pytest.raises(ValueError):
        pipe.add_covariance(cov)
    aeff = pipe["aeff"]
    assert isinstance(aeff.params.aeff_scale, DerivedParam) and isinstance(pipe.params.nu_nc_norm, DerivedParam)
    assert "aeff_scale_rotated" in aeff.params.names and "nu_nc_norm_rotated" in pipe.params.free.names
    assert "aeff_scale" not in pipe.params.free.names
    evals, evecs = np.linalg.eig(np.array([[0.04, 0.01], [0.01, 0.04]]))
    means = np.array([1.5, 1.0])            # aeff_scale: uniform prior on [0, 3]; nu_nc_norm: 1.0 +/- 0.2
    rs = np.random.RandomState(2)
    for k in range(6):
        v = np.zeros(2) if k == 0 else rs.uniform(-0.15, 0.15, 2)
        pipe.params.aeff_scale_rotated.value = v[0]
        pipe.params.nu_nc_norm_rotated.value = v[1]
        x = v @ np.linalg.inv(evecs) + means
        np.testing.assert_allclose([pipe.params.aeff_scale.value.m, pipe.params.nu_nc_norm.value.m], x, atol=1e-12)
        plain.params.aeff_scale.value = x[0]
        plain.params.nu_nc_norm.value = x[1]
        got, want = pipe.get_outputs(), plain.get_outputs()
        for g, w in zip(got, want):
            np.testing.assert_allclose(g.hist, w.hist, rtol=1e-12, atol=1e-300, err_msg="%s at step %d" % (g.name, k))
        np.testing.assert_allclose(pipe.params.priors_penalty("llh") - plain.params.priors_penalty("llh")
                                   + plain.params.nu_nc_norm.prior_penalty("llh"),
                                   -0.5 * (x - means) @ np.linalg.inv(np.array([[0.04, 0.01], [0.01, 0.04]])) @ (x - means),
                                   atol=1e-10)


def test_selection_switch_alone_is_seen_by_the_plan(oracle):
    """found by scripts/dev/fuzz_pipeline.py (round 4): `select_params` exchanges parameter OBJECTS without setting a
    value; the evaluation plan looked at the structural counter only after the value counter had moved, so a switch of the
    mass ordering ALONE was replayed with the other ordering's oscillation tables.  Every evaluation after a switch --
    with nothing else touched -- must be the new selection's."""
    from pisa_amd.core.pipeline import Pipeline
    from pisa_amd.core.units import ureg

    pipe = Pipeline("settings/pipeline/example_hip.cfg")
    pipe.get_outputs()
    pipe.get_outputs()
    assert pipe._plan is not None                                  # replaying
    nh = sum(pipe.get_outputs()).hist.copy()
    ref_nh, _ = _oracle_event_pipeline(oracle, pipe)
    pipe.select_params("ih")                                       # ... and nothing else
    t23, dm31 = pipe.params.theta23.value.m_as("deg"), pipe.params.deltam31.value.m_as("eV**2")
    t13 = pipe.params.theta13.value.m_as("deg")
    assert dm31 < 0 and t13 != 8.5                                   # the other ordering's own values
    ih_maps = pipe.get_outputs()
    ref_ih, _ = _oracle_event_pipeline(oracle, pipe, theta23_deg=t23, dm31=dm31, theta13_deg=t13)
    for m in ih_maps:
        np.testing.assert_allclose(m.hist, ref_ih[m.name], rtol=1e-11, atol=1e-300, err_msg=m.name)
    assert np.abs(sum(ih_maps).hist - nh).max() > 1e-6 * nh.max()
    pipe.get_outputs()
    pipe.select_params("nh")
    for m in pipe.get_outputs():
        np.testing.assert_allclose(m.hist, ref_nh[m.name], rtol=1e-11, atol=1e-300, err_msg=m.name)
    # the same with values set on the deselected objects in between
    pipe.select_params("ih")
    pipe.params.theta23.value = 51.0 * ureg.deg
    pipe.get_outputs()
    pipe.select_params("nh")
    for m in pipe.get_outputs():
        np.testing.assert_allclose(m.hist, ref_nh[m.name], rtol=1e-11, atol=1e-300, err_msg=m.name)


@pytest.mark.parametrize("metric", ["mcllh_eff", "correct_chi2", "conv_llh"])
def test_fit_with_a_metric_beyond_the_fused_tail(metric):
    """a metric the one-kernel tail does not carry (stats.py:384-438, 697-730, 558-596): the minimiser callable goes
    through the maps and `pisa_hip_metric`; the value at the fit's end is the restatement's on those maps, the fit
    improves on its start, and the parameters move towards the injected truth"""
    from oracle import stages_oracle as so
    from pisa_amd.analysis.analysis import Analysis
    from pisa_amd.core.distribution_maker import DistributionMaker
    from pisa_amd.core.units import ureg

    dm = DistributionMaker("settings/pipeline/example_hip.cfg")
    for name in dm.params.free.names:
        if name not in ("theta23", "deltam31"):
            dm.params.fix(name)
    dm.params.theta23.value = 46.5 * ureg.degree
    dm.params.deltam31.value = 2.6e-3 * ureg.eV ** 2
    data = dm.get_outputs(return_sum=True)
    dm.params.reset_free()
    start = data[0].metric_total(dm.get_outputs(return_sum=True)[0], metric)
    res = Analysis().fit_hypo(data, dm, metric)
    templ = dm.get_outputs(return_sum=True)[0]
    want = np.nansum(so.metric_wide(metric, data[0].hist, templ.hist, templ.std_devs))
    prior = dm.params.priors_penalty(metric=metric)
    np.testing.assert_allclose(res.metric_val, want + prior, rtol=1e-9, atol=1e-9)
    better = res.metric_val > start if metric.endswith("llh") or metric.startswith("mcllh") else res.metric_val < start
    assert better, (start, res.metric_val)
    nominal = dm.params.deltam31.nominal_value.m_as("eV**2")
    assert abs(res.params.deltam31.value.m_as("eV**2") - 2.6e-3) < abs(nominal - 2.6e-3)


def test_pipeline_cfg_with_the_services_around_the_path(oracle, tmp_path):
    """a pipeline TEXT that goes through services of round 4: reco.resolutions (once at setup), osc.two_nu_osc in place
    of prob3, utils.kfold and utils.bootstrap in front of the histogram; the parser hands every kwarg over in the form
    the stages take (ints, bools, parameters), and the event weights and maps equal the chain restated with numpy on
    the columns the loader left"""
    import re

    from oracle import stages_oracle as so
    from pisa_amd.core.pipeline import Pipeline
    from pisa_amd.utils.resources import find_resource

    text = open(find_resource("settings/pipeline/example_hip.cfg")).read()
    text = text.replace("order = data.synthetic_events, flux.barr_simple, osc.prob3, aeff.aeff, utils.hist",
                        "order = data.synthetic_events, reco.resolutions, flux.barr_simple, osc.two_nu_osc, aeff.aeff, utils.kfold,"
                        " utils.bootstrap, utils.hist")
    new_osc = ("[osc.two_nu_osc]\napply_mode = events\nparam.theta23 = 42. * units.degree\nparam.theta23.fixed = False\n"
               "param.theta23.range = [0, 90] * units.degree\nparam.theta23.prior = uniform\n"
               "param.deltam31 = 2.5e-3 * units.eV**2\nparam.deltam31.fixed = True\n\n"
               "[reco.resolutions]\ncalc_mode = events\nrelative_pid = True\nparam.energy_improvement = 0.25\n"
               "param.energy_improvement.fixed = True\nparam.coszen_improvement = 0.5\nparam.coszen_improvement.fixed = True\n"
               "param.pid_improvement = 0.1\nparam.pid_improvement.fixed = True\n\n"
               "[utils.kfold]\ncalc_mode = events\napply_mode = events\nn_splits = 4\nselect_split = 2\nrenormalize = True\n\n"
               "[utils.bootstrap]\ncalc_mode = events\napply_mode = events\nseed = 5\n\n")
    text, n = re.subn(r"\[osc\.prob3\].*?(?=\[aeff\.aeff\])", new_osc, text, flags=re.S)
    assert n == 1
    text = text.replace("param.n_events = 1.2e5", "param.n_events = 2.4e4")
    path = tmp_path / "services.cfg"
    path.write_text(text)
    pipe = Pipeline(str(path))
    assert [s.service_name for s in pipe.stages] == ["synthetic_events", "resolutions", "barr_simple", "two_nu_osc", "aeff", "kfold",
                                                     "bootstrap", "hist"]
    assert pipe["kfold"].n_splits == 4 and pipe["kfold"].renormalize is True and pipe["bootstrap"].seed == 5
    assert pipe["resolutions"].relative_pid is True
    maps = pipe.get_outputs()
    assert not pipe["hist"].fused_last_eval                     # stage by stage: not the replayable shape
    ob = pipe.output_binning
    rng = np.random.default_rng(5)
    scale = pipe.params.aeff_scale.m_as("dimensionless") * pipe.params.livetime.m_as("sec")
    from sklearn.model_selection import KFold

    for c in pipe.data.containers:
        c.representation = "events"
        n = c.size
        e, cz = np.array(c["true_energy"]), np.array(c["true_coszen"])
        flav = 0 if "nue" in c.name else (1 if "numu" in c.name else 2)
        w = so.two_nu_weights(np.array(c["nu_flux"]), np.deg2rad(42.0), 2.5e-3, e, cz, flav, np.array(c["initial_weights"]))
        w = w * (np.array(c["weighted_aeff"]) * scale)
        fold = np.zeros(n)
        fold[list(KFold(n_splits=4).split(np.empty(n)))[2][1]] = 4.0
        w = w * fold
        w = w * np.bincount(rng.integers(n, size=n), minlength=n)
        np.testing.assert_allclose(c["weights"], w, rtol=1e-10, atol=1e-300)
        got = maps[c.name]
        c.representation = ob
        np.testing.assert_allclose(got.hist.ravel(), c["weights"], rtol=0, atol=0)
        assert got.hist.sum() > 0 and np.isfinite(got.hist).all()
    # the improved resolutions were applied once at setup: reco_energy sits a quarter of the way towards the truth
    plain = tmp_path / "plain.cfg"
    plain.write_text(open(find_resource("settings/pipeline/example_hip.cfg")).read().replace("param.n_events = 1.2e5", "param.n_events = 2.4e4"))
    twin = Pipeline(str(plain))
    twin.get_outputs()
    for c, t in zip(pipe.data.containers, twin.data.containers):
        c.representation = t.representation = "events"
        assert c.name == t.name and c.size == t.size
        np.testing.assert_array_equal(c["reco_energy"], so.shift_toward(np.array(t["reco_energy"]), np.array(t["true_energy"]), 0.25))
        np.testing.assert_array_equal(c["reco_coszen"], so.shift_toward(np.array(t["reco_coszen"]), np.array(t["true_coszen"]), 0.5, (-1, 1)))
        np.testing.assert_array_equal(c["pid"], so.shift_toward(np.array(t["pid"]), 1.0 if c.name in ("numu_cc", "numubar_cc") else 0.0, 0.1))


def test_pipeline_and_maker_tables_lookup_and_hash():
    """pipeline.py:138-146, 199-247, 676-680; distribution_maker.py:211-217: the table of stages, stages by stage
    name / number / attribute, a hash that follows the parameter values"""
    from pisa_amd.core.distribution_maker import DistributionMaker
    from pisa_amd.core.units import ureg

    dm = DistributionMaker("settings/pipeline/example_hip.cfg")
    pipe = dm.pipelines[0]
    table = pipe.tabulate()
    lines = table.splitlines()
    assert len(lines) >= 1 + len(pipe.stages) and "stage number" in lines[0] and "# free params" in lines[0]
    assert all(s.__class__.__name__ in table for s in pipe.stages) and repr(pipe).count("\n") >= len(pipe.stages)
    assert pipe.index("osc") == 2 and pipe.index(4) == 4 and pipe.osc is pipe["osc"] is pipe.stages[2]
    with pytest.raises(ValueError):
        pipe.index("reco")
    with pytest.raises(AttributeError):
        pipe.no_such_stage  # pylint: disable=pointless-statement
    assert "<table>" in pipe._repr_html_() and "neutrinos" in dm.tabulate() and "<table>" in dm._repr_html_()
    h0, m0 = pipe.hash, dm.hash
    assert h0 == pipe.hash
    dm.params.theta23.value = 44.0 * ureg.degree
    assert pipe.hash != h0 and dm.hash != m0
    dm.params.reset_free()
    assert pipe.hash == h0 and dm.hash == m0
