"""The reference's service smoke test, ported (pisa_tests/test_services.py:136-448): every `class X(Stage)` under
pisa_amd/stages is instantiated through its module's `init_test(**param_kwargs)`, given two fabricated 10-event
containers (`add_test_inputs`: linspace(0.1, 1, 10) columns, random flux pairs, aux data nubar / flav = 1; an empty
ContainerSet for the data services), `calc_mode` chosen as the reference chooses it, then `setup()` + `run()`.
As there, the assertion is "does not raise" -- plus, here, that every kernel status stays clean and the outputs are
finite.  The services whose input file is not part of the public data release (the neutrino MC of csv_loader,
`setup.py` / README of the reference) are skipped with that reason."""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STAGES = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pisa_amd", "stages")
AUX_DATA_KEYS = ["nubar", "flav"]


def _services():
    out = []
    for stage in sorted(os.listdir(STAGES)):
        d = os.path.join(STAGES, stage)
        if not os.path.isdir(d) or stage.startswith("_"):
            continue
        for f in sorted(os.listdir(d)):
            if f.endswith(".py") and not f.startswith("_"):
                with open(os.path.join(d, f)) as fh:
                    names = [ln.split()[1].split("(")[0] for ln in fh if ln.startswith("class ") and "(Stage)" in ln]
                if names:
                    assert len(names) == 1, (stage, f, names)       # one service per module, as in the reference
                    out.append("%s.%s" % (stage, f[:-3]))
    return out


SERVICES = _services()


def test_every_service_is_found_and_has_an_init_test():
    assert len(SERVICES) >= 15 and "osc.prob3" in SERVICES and "utils.hist" in SERVICES and "utils.kde" in SERVICES
    for s in SERVICES:
        assert hasattr(importlib.import_module("pisa_amd.stages." + s), "init_test"), s


def _add_test_inputs(service, empty):
    """pisa_tests/test_services.py:136-166"""
    from pisa_amd import FTYPE
    from pisa_amd.core.container import Container, ContainerSet
    from pisa_amd.stages.utils.kde import service_test_binning

    if empty:
        service.data = ContainerSet("data")
        return
    rs = np.random.RandomState(0)
    c1, c2 = Container("test1_cc"), Container("test2_nc")
    keys = set(list(service.expected_container_keys) + ["reco_energy", "reco_coszen", "pid", "weights"])
    for k in sorted(keys):
        if k in AUX_DATA_KEYS:
            c1.set_aux_data(k, 1)
            c2.set_aux_data(k, 1)
        elif k in ("nu_flux", "nu_flux_nominal", "nubar_flux_nominal"):
            c1[k] = rs.random_sample((10, 2)).astype(FTYPE)
            c2[k] = rs.random_sample((10, 2)).astype(FTYPE)
        elif k.endswith("mask"):
            c1[k] = np.ones(10, dtype=np.int64)
            c2[k] = np.zeros(10, dtype=np.int64)
        else:
            c1[k] = np.linspace(0.1, 1, 10, dtype=FTYPE)
            c2[k] = np.linspace(0.1, 1, 10, dtype=FTYPE)
    service.data = ContainerSet("data", [c1, c2])
    b = service_test_binning()
    service.data["output_binning"] = b
    service.data["regularized_output_binning"] = b


@pytest.mark.parametrize("stage_dot_service", SERVICES)
def test_service_sets_up_and_runs(stage_dot_service):
    from pisa_amd.core.stage import Stage
    from pisa_amd.stages.utils.kde import service_test_binning

    module = importlib.import_module("pisa_amd.stages." + stage_dot_service)
    try:
        service = module.init_test(prior=None, range=None, is_fixed=True)
    except (IOError, OSError, ValueError) as err:
        if "neutrino_mc" in str(err) or "Could not find resource" in str(err):
            pytest.skip("input file not part of the public data release: %s" % err)
        raise
    assert isinstance(service, Stage)
    if service.data is None:
        _add_test_inputs(service, empty=stage_dot_service.split(".")[0] == "data")
    if service.calc_mode is None and None not in service.supported_reps["calc_mode"]:
        try:
            service.calc_mode = "events"
        except ValueError:
            service.calc_mode = service_test_binning()
    if service.apply_mode is None and None not in service.supported_reps["apply_mode"]:
        try:
            service.apply_mode = "events"
        except ValueError:
            service.apply_mode = service_test_binning()
    service.setup()
    service.run()
    # beyond the reference's "does not raise": what the service left in the containers is finite
    for c in service.data.containers:
        for rep in c.representations:
            c.representation = rep
            if "weights" in c.keys:
                w = np.asarray(c["weights"])
                assert w.size and np.all(np.isfinite(w)), (stage_dot_service, c.name)


def test_detectors_shared_and_per_detector_parameters():
    """`Detectors` (pisa/core/detectors.py:36-381; the flow of its own test, :384-433, on two copies of the event-mode
    example): pipelines grouped by detector name, shared parameters ONE parameter of the fit, the others per
    detector under `<name>_<detector>`, selections switched everywhere, rescaled values dealt in the order of
    `params.free`."""
    import numpy as np

    from pisa_amd.core.detectors import Detectors
    from pisa_amd.core.pipeline import Pipeline
    from pisa_amd.core.units import ureg

    pipes = [Pipeline("settings/pipeline/example_hip.cfg") for _ in range(3)]
    pipes[0].detector_name = pipes[1].detector_name = "detector1"
    pipes[2].detector_name = "detector2"
    with pytest.raises(NameError):
        Detectors([pipes[0], Pipeline("settings/pipeline/example_hip.cfg")])       # one pipeline without a detector
    with pytest.raises(NameError):
        Detectors(pipes, shared_params=["no_such_param"])
    model = Detectors(pipes, shared_params=["theta23", "deltam31", "delta_index"])
    assert model.det_names == ["detector1", "detector2"]
    assert [len(d.pipelines) for d in model] == [2, 1] and [d.detector_name for d in model.distribution_makers] == model.det_names
    names = list(model.params.names)
    assert names[:3] == ["theta23", "deltam31", "delta_index"]                      # the shared ones first
    assert "aeff_scale" in names and "aeff_scale_detector2" in names and "theta23_detector2" not in names
    assert model.param_selections == ["nh"]

    nominal = model.get_outputs(return_sum=True)
    assert len(nominal) == 2 and nominal[0][0].hist.shape == nominal[1][0].hist.shape
    np.testing.assert_allclose(nominal[0][0].hist, 2 * nominal[1][0].hist, rtol=1e-12)   # two pipelines against one

    model.params.delta_index.value = 0.05                   # shared
    model.params.aeff_scale.value = 2.0                     # detector1 only
    model.params.aeff_scale_detector2.value = 0.5           # detector2 only
    out = model.get_outputs(return_sum=True)
    d1, d2 = model.distribution_makers
    assert d1.params.delta_index.value == 0.05 and d2.params.delta_index.value == 0.05
    assert d1.params.aeff_scale.value == 2.0 and d2.params.aeff_scale.value == 0.5
    # aeff_scale multiplies every weight: detector1 = 2 pipelines x 2.0, detector2 = 1 pipeline x 0.5
    np.testing.assert_allclose(out[0][0].hist, 8 * out[1][0].hist, rtol=1e-12)
    assert np.abs(out[1][0].hist / 0.5 - nominal[1][0].hist).max() > 0              # and delta_index moved the maps
    per_pipeline = model.get_outputs()
    assert [len(x) for x in per_pipeline] == [2, 1]

    # selections
    t23 = dict(nh=model.params.theta23.value)
    model.select_params("ih")
    t23["ih"] = model.params.theta23.value
    assert t23["ih"] != t23["nh"] and all(d.params.theta23.value == t23["ih"] for d in model)
    model.select_params("nh")
    assert model.params.theta23.value == t23["nh"] and model.param_selections == ["nh"]

    # free parameters: values / rescaled values in the order of `params.free`
    free = list(model.params.free.names)
    assert free[:3] == ["theta23", "deltam31", "delta_index"] and "aeff_scale_detector2" in free
    assert model.shared_param_ind_list == [[(list(d.params.free.names).index(n), model.shared_params.index(n))
                                            for n in d.params.free.names if n in model.shared_params] for d in model]
    r = np.linspace(0.2, 0.8, len(free))
    model._set_rescaled_free_params(r)
    for name, rv in zip(free, r):
        np.testing.assert_allclose(model.params[name]._rescaled_value, rv, rtol=1e-12)
    own = dict(zip(free, r))
    for d in model:
        for prm in d.params.free:
            want = own.get("%s_%s" % (prm.name, d.detector_name), own[prm.name])
            np.testing.assert_allclose(prm._rescaled_value, want, rtol=1e-12)
    vals = [p.value for p in model.params.free]
    vals[free.index("aeff_scale_detector2")] = 1.25 * ureg.dimensionless
    model.set_free_params(vals)
    assert model.distribution_makers[1].params.aeff_scale.value == 1.25 and model.distribution_makers[0].params.aeff_scale.value != 1.25
    model.reset_free()
    assert model.params.aeff_scale_detector2.value == model.params.aeff_scale_detector2.nominal_value
    model.randomize_free_params(random_state=3)
    assert model.params.theta23.value == d1.params.theta23.value == d2.params.theta23.value
    counts = model.num_events_per_bin
    assert len(counts) == 2 and counts[0].sum() == 2 * counts[1].sum() > 0 and len(model.empty_bin_indices) == 2


def test_fits_over_detectors_and_over_a_variable_binning():
    """`Analysis.fit_hypo` where the template is a list: one MapSet per detector (`Detectors`, a metric each,
    analysis.py:2588-2599) or per selection of a `VarBinning` (:2600-2608).  Pseudo-data made at injected values,
    fit started at the nominal ones: the injected values come back and the metric is the sum of the parts."""
    import numpy as np

    from pisa_amd.analysis.analysis import Analysis
    from pisa_amd.core.detectors import Detectors
    from pisa_amd.core.distribution_maker import DistributionMaker
    from pisa_amd.core.pipeline import Pipeline
    from pisa_amd.core.units import ureg

    def only_free(maker, names):
        for p in maker.params:
            p.is_fixed = p.name not in names

    ana = Analysis()
    with pytest.raises(ValueError):
        ana._sign(["llh", "chi2"])
    assert ana._sign(["llh", "poisson_llh"]) == -1 and ana._sign(["chi2", "mod_chi2"]) == 1

    # two detectors sharing theta23, each with its own aeff_scale
    pipes = [Pipeline("settings/pipeline/example_hip.cfg") for _ in range(2)]
    pipes[0].detector_name, pipes[1].detector_name = "near", "far"
    model = Detectors(pipes, shared_params=["theta23"])
    only_free(model, ("theta23", "aeff_scale", "aeff_scale_far"))
    for d in model:
        only_free(d, ("theta23", "aeff_scale"))
    model.init_params()
    assert list(model.params.free.names) == ["theta23", "aeff_scale", "aeff_scale_far"]
    model.params.theta23.value = 46.5 * ureg.deg
    model.params.aeff_scale.value = 1.1
    model.params.aeff_scale_far.value = 0.9
    data = [ms for ms in model.get_outputs(return_sum=True)]
    data = [type(ms)([ms[0]._new(ms[0].hist.copy(), None, name="total")]) for ms in data]
    assert abs(data[0][0].hist.sum() / data[1][0].hist.sum() - 1.1 / 0.9) < 1e-9
    model.reset_free()
    start = ana._total_metric(data, model.get_outputs(return_sum=True), model, ["chi2", "chi2"])
    parts = [d.metric_total(expected_values=h, metric="chi2") for d, h in zip(data, model.get_outputs(return_sum=True))]
    np.testing.assert_allclose(start, sum(parts) + model.params.priors_penalty("chi2"), rtol=1e-12)
    fit = ana.fit_hypo(data, model, ["chi2", "chi2"], reset_free=True)
    assert fit.metric_val < 1e-4 * start           # the default tolerances of the minimiser (ftol 2e-5)
    np.testing.assert_allclose(model.params.theta23.value.m_as("deg"), 46.5, atol=0.2)
    np.testing.assert_allclose(model.params.aeff_scale.value.m, 1.1, atol=2e-3)
    np.testing.assert_allclose(model.params.aeff_scale_far.value.m, 0.9, atol=2e-3)
    assert model.distribution_makers[1].params.aeff_scale.value.m == model.params.aeff_scale_far.value.m

    # one maker whose pipeline has a variable binning: the template is a list of MapSets
    maker = DistributionMaker(["settings/pipeline/varbin_example_hip.cfg"])
    only_free(maker, ("theta23", "aeff_scale"))
    maker.params.theta23.value = 47.0 * ureg.deg
    maker.params.aeff_scale.value = 1.2
    template = maker.get_outputs(return_sum=True)
    assert isinstance(template, list) and len(template) == 2 and template[0].names == ["total"]
    pseudo = [type(ms)([ms[0]._new(ms[0].hist.copy(), None, name="total")]) for ms in template]
    maker.reset_free()
    start = ana._total_metric(pseudo, maker.get_outputs(return_sum=True), maker, "chi2")
    fit = ana.fit_hypo(pseudo, maker, "chi2", reset_free=True)
    assert fit.metric_val < 1e-4 * start
    np.testing.assert_allclose(maker.params.theta23.value.m_as("deg"), 47.0, atol=0.2)
    np.testing.assert_allclose(maker.params.aeff_scale.value.m, 1.2, atol=2e-3)
    # starting ON the data: no fit
    maker.params.theta23.value = 47.0 * ureg.deg
    maker.params.aeff_scale.value = 1.2
    again = ana.fit_hypo(pseudo, maker, "chi2", reset_free=False)
    assert again.minimizer_metadata["nit"] == 0 and again.num_distributions_generated == 0


def test_nested_fit_strategies():
    """`Analysis.fit_recursively` (pisa/analysis/analysis.py:854-1560): octants around a local scipy fit find an
    injected second-octant theta23 and keep the mirrored fit as the alternate; ranges / grid_scan / best_of / staged /
    condition run the same inner fit their way and end at the same point, constrained ends on its bound; the maker is
    left at the best fit with its original ranges and nominal values."""
    import numpy as np

    from pisa_amd.analysis.analysis import Analysis
    from pisa_amd.core.distribution_maker import DistributionMaker
    from pisa_amd.core.units import ureg

    maker = DistributionMaker(["settings/pipeline/example_hip.cfg"])
    for p in maker.params:
        p.is_fixed = p.name not in ("theta23", "aeff_scale")
    nominal = {p.name: p.nominal_value for p in maker.params.free}
    t23_range = [q for q in maker.params.theta23.range]
    maker.params.theta23.value = 49.0 * ureg.deg
    maker.params.aeff_scale.value = 1.08
    truth = maker.get_outputs(return_sum=True)
    data = type(truth)([truth[0]._new(truth[0].hist.copy(), None, name="total")])
    maker.reset_free()
    ana = Analysis()
    local = dict(method="scipy", method_kwargs=dict(method=dict(value="L-BFGS-B"),
                                                    options=dict(value=dict(ftol=1e-9, gtol=1e-8, eps=1e-6, maxiter=200))),
                 local_fit_kwargs=None)

    def at_truth(fit, tol=0.05):
        np.testing.assert_allclose(maker.params.theta23.value.m_as("deg"), 49.0, atol=tol)
        np.testing.assert_allclose(maker.params.aeff_scale.value.m, 1.08, atol=1e-3)
        np.testing.assert_allclose(fit.params.theta23.value.m_as("deg"), 49.0, atol=tol)
        assert fit.metric_val < 1e-3
        assert maker.params.theta23.range[0] == t23_range[0] and maker.params.theta23.range[1] == t23_range[1]
        assert all(maker.params[n].nominal_value == v for n, v in nominal.items())

    # the local fit alone (on this sample it crosses the octant boundary by itself; histories only when asked for)
    alone = ana.fit_recursively(data, maker, "chi2", None, **local)
    at_truth(alone)
    assert alone.fit_history is None
    maker.reset_free()
    octants = dict(method="octants", method_kwargs=dict(angle="theta23", inflection_point=45 * ureg.deg),
                   local_fit_kwargs=local)
    fit = ana.fit_recursively(data, maker, "chi2", None, store_fit_history=True, **octants)
    at_truth(fit)
    assert fit.alternate_fit.params.theta23.value.m_as("deg") <= 45.0 and fit.alternate_fit.metric_val > fit.metric_val   # held at the octant boundary
    assert fit.fit_history and fit.params.theta23.range[1] == t23_range[1]
    # the same through two ranges of the angle
    maker.reset_free()
    ranges = dict(method="fit_ranges", method_kwargs=dict(param_name="theta23", ranges=[[31, 45] * ureg.deg, [45, 59] * ureg.deg]),
                  local_fit_kwargs=local)
    at_truth(ana.fit_recursively(data, maker, "chi2", None, **ranges))
    # a grid of starting points; then with the grid parameter held and a refined fit from the best point
    maker.reset_free()
    grid = dict(method="grid_scan", method_kwargs=dict(grid=dict(theta23=[40, 50] * ureg.deg)), local_fit_kwargs=local)
    fit = ana.fit_recursively(data, maker, "chi2", None, **grid)
    at_truth(fit)
    assert fit.grid_metric_vals.shape == (2,) and fit.grid_metric_vals[1] < fit.grid_metric_vals[0]
    maker.reset_free()
    held = dict(method="grid_scan", method_kwargs=dict(grid=dict(theta23=[41, 48, 50.5] * ureg.deg), fix_grid_params=True,
                                                       refined_fit=local), local_fit_kwargs=local)
    at_truth(ana.fit_recursively(data, maker, "chi2", None, **held))
    assert list(maker.params.free.names) == ["theta23", "aeff_scale"]
    # best_of: the plain local fit against the octant search; staged: coarse, then fine from where it ended
    maker.reset_free()
    at_truth(ana.fit_recursively(data, maker, "chi2", None, "best_of", None, [local, octants]))
    maker.reset_free()
    coarse = dict(local, method_kwargs=dict(method=dict(value="L-BFGS-B"), options=dict(value=dict(ftol=1e-3, eps=1e-4, maxiter=5))))
    at_truth(ana.fit_recursively(data, maker, "chi2", None, "staged", None, [dict(octants, local_fit_kwargs=coarse), local]))
    # condition: a callable, or text that evaluates to one
    maker.reset_free()
    cond = dict(condition_func="lambda maker: 'theta23' in maker.params.free.names")
    at_truth(ana.fit_recursively(data, maker, "chi2", None, "condition", cond, [octants, local]))
    # constrained: theta23 held below 47 deg in rescaled space -> the fit ends on the bound, worse than the free fit
    maker.reset_free()
    r47 = (47.0 - t23_range[0].m_as("deg")) / (t23_range[1].m_as("deg") - t23_range[0].m_as("deg"))
    bound = dict(method="constrained",
                 method_kwargs=dict(ineq_func=lambda params: r47 - params.theta23._rescaled_value,
                                    necessary_free_params=["theta23"], starting_values=dict(theta23=46.0 * ureg.deg)),
                 local_fit_kwargs=local)
    fit = ana.fit_recursively(data, maker, "chi2", None, **bound)
    assert 46.5 < fit.params.theta23.value.m_as("deg") <= 47.0 + 0.02 and fit.metric_val > 1e-3
    # refusals
    with pytest.raises(ImportError):
        ana.fit_recursively(data, maker, "chi2", None, "iminuit", {}, None)
    with pytest.raises(ValueError):
        ana.fit_recursively(data, maker, "chi2", None, "simulated_annealing", {}, None)
    with pytest.raises(AssertionError):
        ana.fit_recursively(data, maker, ["chi2", "chi2"], None, **local)


@pytest.mark.parametrize("name", ["differential_evolution", "basinhopping", "dual_annealing", "shgo"])
def test_global_scipy_methods(name):
    """`fit_recursively("scipy", {"global_method": ...})` (analysis.py:1594-1680, 1811-1893): the four global optimisers
    over the rescaled free parameters, small budgets; each reports a point it evaluated (the best of its population for
    differential evolution, no worse than the start for those that start there) and what it spent; with a fixed seed the
    run repeats itself"""
    from pisa_amd.analysis.analysis import Analysis
    from pisa_amd.core.distribution_maker import DistributionMaker
    from pisa_amd.core.units import ureg

    dm = DistributionMaker("settings/pipeline/example_hip.cfg")
    for n in dm.params.free.names:
        if n not in ("theta23", "deltam31"):
            dm.params.fix(n)
    dm.params.theta23.value = 47.0 * ureg.degree
    dm.params.deltam31.value = 2.55e-3 * ureg.eV ** 2
    data = dm.get_outputs(return_sum=True)
    dm.params.reset_free()
    start = data[0].metric_total(dm.get_outputs(return_sum=True)[0], "mod_chi2")
    local = {"method": {"value": "L-BFGS-B"}, "options": {"value": {"ftol": 1e-6, "gtol": 1e-5, "eps": 1e-4, "maxiter": 30}}}
    options = {"differential_evolution": dict(seed=3, maxiter=4, popsize=6, tol=0.1, polish=False),
               "basinhopping": dict(seed=3, niter=2, stepsize=0.2),
               "dual_annealing": dict(seed=3, maxiter=15, no_local_search=True),
               "shgo": dict(n=16, iters=1, sampling_method="sobol")}[name]

    def run():
        dm.params.reset_free()
        return Analysis().fit_recursively(data, dm, "mod_chi2", None, "scipy", {"global_method": name, "options": dict(options)},
                                          local if name in ("basinhopping", "shgo") else None, store_fit_history=True)

    res = run()
    assert res.minimizer_metadata["global_method"] == name and res.num_distributions_generated >= 10
    assert len(res.fit_history) == res.num_distributions_generated
    values = np.array([h[0] for h in res.fit_history])
    assert np.min(np.abs(values - res.metric_val)) <= 1e-12 * max(1.0, abs(res.metric_val))     # a point that was evaluated
    if name == "differential_evolution":            # (starts from its own population, not from the nominal point)
        assert res.metric_val == values.min()
    else:
        assert res.metric_val <= start * (1 + 1e-12)
    # the parameters were left at the optimum: the template there gives the reported value
    np.testing.assert_allclose(data[0].metric_total(dm.get_outputs(return_sum=True)[0], "mod_chi2")
                               + dm.params.priors_penalty(metric="mod_chi2"), res.metric_val, rtol=1e-10)
    if name != "shgo":
        again = run()
        assert again.metric_val == res.metric_val and again.num_distributions_generated == res.num_distributions_generated
    with pytest.raises(ValueError):
        Analysis().fit_recursively(data, dm, "mod_chi2", None, "scipy", {"global_method": "simulated_magic", "options": {}}, None)


def test_constrained_local_fits_as_in_the_reference_unit_test():
    """analysis.py:3019-3160 (`test_constrained_minimization`): constraints among the minimiser options, given as
    functions or strings of the ParamSet -- SLSQP with two inequalities and an equality, COBYLA with inequalities,
    trust-constr with an inequality; the best fit respects them"""
    from pisa_amd.analysis.analysis import Analysis
    from pisa_amd.core.distribution_maker import DistributionMaker
    from pisa_amd.core.units import ureg

    dm = DistributionMaker("settings/pipeline/example_hip.cfg")
    data = dm.get_outputs(return_sum=True)
    assert set(dm.params.free.names) >= {"theta23", "deltam31", "aeff_scale", "delta_index"}
    tol = 1e-5

    def fit(method, constraints, **options):
        dm.params.reset_free()
        dm.params.theta23.value = 39.0 * ureg.degree
        settings = {"method": {"value": method, "desc": ""}, "options": {"value": dict(options, constraints=constraints), "desc": {}}}
        return Analysis().fit_recursively(data_dist=data, hypo_maker=dm, metric="chi2", external_priors_penalty=None,
                                          store_fit_history=True, method="scipy", method_kwargs=settings)

    min_delta_index, max_aeff_scale, t23 = 5e-3, 0.986, 44.2
    bf = fit("slsqp", [{"type": "ineq", "fun": lambda params: params.delta_index.m_as("dimensionless") - min_delta_index},
                       {"type": "ineq", "fun": 'lambda p: -p.aeff_scale.m_as("dimensionless") + %s' % max_aeff_scale},
                       {"type": "eq", "fun": lambda params: params.theta23.m_as("degree") - t23}], ftol=1e-7, eps=1e-6, maxiter=100)
    assert bf.minimizer_metadata["success"], bf.minimizer_metadata
    np.testing.assert_allclose(bf.params.theta23.m_as("degree"), t23, rtol=1e-8)
    assert bf.params.delta_index.m_as("dimensionless") >= min_delta_index - tol
    assert bf.params.aeff_scale.m_as("dimensionless") <= max_aeff_scale + tol
    assert isinstance(bf.fit_history[0][0], float) and bf.metric_val > 0
    min_t23, min_aeff_scale = 46.0, 1.02
    bf = fit("cobyla", [{"type": "ineq", "fun": lambda params: params.theta23.m_as("degree") - min_t23},
                        {"type": "ineq", "fun": lambda params: params.aeff_scale.m_as("dimensionless") - min_aeff_scale}],
             rhobeg=0.05, tol=1e-6, maxiter=300)
    assert bf.minimizer_metadata["success"], bf.minimizer_metadata
    assert bf.params.theta23.m_as("degree") >= min_t23 - 1e-3 and bf.params.aeff_scale.m_as("dimensionless") >= min_aeff_scale - 1e-4
    bf = fit("trust-constr", [{"type": "ineq", "fun": lambda params: params.aeff_scale.m_as("dimensionless") - min_aeff_scale}],
             maxiter=60, gtol=1e-4, xtol=1e-6, finite_diff_rel_step=1e-5)
    assert bf.params.aeff_scale.m_as("dimensionless") >= min_aeff_scale - 1e-4
    with pytest.raises(TypeError):
        fit("slsqp", [{"type": "ineq", "fun": "3.0"}])


def test_detailed_metric_info_of_a_fit_result():
    """analysis.py:373-459: the metric map by map, the priors' penalties, the per-bin values as maps; other metrics
    alongside the fit's"""
    from oracle import stages_oracle as so
    from pisa_amd.analysis.analysis import Analysis
    from pisa_amd.core.distribution_maker import DistributionMaker
    from pisa_amd.core.units import ureg

    dm = DistributionMaker("settings/pipeline/example_hip.cfg")
    for n in dm.params.free.names:
        if n not in ("theta23", "delta_index"):
            dm.params.fix(n)
    dm.params.theta23.value = 45.5 * ureg.degree
    data = dm.get_outputs(return_sum=True)
    dm.params.reset_free()
    res = Analysis().fit_hypo(data, dm, "mod_chi2")
    info = res.add_detailed_metric_info(data, dm, other_metrics=["correct_chi2", "llh"], include_maps_binned=True)
    assert list(info) == ["detector_name", "correct_chi2", "llh", "mod_chi2"] or list(info) == ["correct_chi2", "llh", "mod_chi2"]
    templ = res.hypo_asimov_dist[0]
    for m in ("mod_chi2", "correct_chi2", "llh"):
        d = info[m]
        assert list(d["maps"]) == ["total"] and d["maps_binned"][0].hist.shape == templ.hist.shape
        np.testing.assert_allclose(np.nansum(d["maps_binned"][0].hist), d["maps"]["total"], rtol=1e-12)
        assert d["priors"] == res.params.priors_penalties(metric="mod_chi2") and len(d["priors"]) == len(res.params)
    np.testing.assert_allclose(info["correct_chi2"]["maps_binned"][0].hist.ravel(),
                               so.metric_wide("correct_chi2", data[0].hist, templ.hist, templ.std_devs), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(info["mod_chi2"]["maps"]["total"] + sum(info["mod_chi2"]["priors"]), res.metric_val, rtol=1e-10)
