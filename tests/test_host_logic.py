"""CPU tests of the host-side mirror of the reference interface (no GPU):
units, binning numerics, cfg grammar, params / priors / hashing."""
import numpy as np
import pytest

from pisa_amd.core.binning import MultiDimBinning, OneDimBinning
from pisa_amd.core.config_parser import parse_pipeline_config, parse_quantity
from pisa_amd.core.param import Param, ParamSelector, ParamSet, Prior
from pisa_amd.core.units import DimensionalityError, Quantity, ureg


def test_units_match_pint_conventions():
    q = 42.0 * ureg.degree
    assert q.m_as("rad") == np.deg2rad(42.0)
    assert (2.5 * ureg.common_year).m_as("sec") == 2.5 * 365 * 86400
    assert (7.5e-5 * ureg.eV ** 2).m_as("eV**2") == 7.5e-5
    r = [0.001, 0.007] * ureg.eV ** 2
    np.testing.assert_array_equal(r.magnitude, [0.001, 0.007])
    assert (np.array([1.0, 80.0]) * ureg.GeV).m_as("GeV")[1] == 80.0
    assert Quantity(0.3).units == ureg.dimensionless
    assert not (q.units == ureg.dimensionless)
    with pytest.raises(DimensionalityError):
        q.m_as("km")
    assert ureg.parse_expression("33.48 deg").m_as("deg") == 33.48
    assert (1 * ureg.GeV).m_as("eV") == 1e9


def test_binning_numerics_follow_reference():
    # edges: np.logspace / np.linspace (binning.py:416-428); centres: geometric mean (:901-911)
    e = OneDimBinning("true_energy", num_bins=200, is_log=True, domain=[1.0, 1000] * ureg.GeV)
    edges = np.logspace(0, 3, 201)
    np.testing.assert_array_equal(e.edge_magnitudes, edges)
    np.testing.assert_array_equal(e.weighted_centers.m, np.sqrt(edges[:-1] * edges[1:]))
    assert e.weighted_centers.m[0] == 1.017419366180605
    cz = OneDimBinning("true_coszen", num_bins=200, is_lin=True, domain=[-1, 1])
    assert cz.weighted_centers.m[0] == -0.995 and not cz.is_irregular and not e.is_irregular
    b = MultiDimBinning([e, cz])
    assert b.shape == (200, 200) and b.size == 40000 and b.names == ["true_energy", "true_coszen"]
    grid = b.meshgrid("weighted_centers")
    assert grid[0].ravel()[201] == e.weighted_centers.m[1]  # flat = iE*n_cz + jcz
    assert hash(b) == hash(MultiDimBinning([e, cz])) and b == MultiDimBinning([e, cz])
    assert hash(b) != hash(MultiDimBinning([cz, e]))
    # dragon's 8-digit log edges are NOT log-uniform at rtol 1e-12 -> irregular
    dragon = OneDimBinning("reco_energy", is_log=True, bin_edges=[5.62341325, 7.49894209, 10.0,
                           13.33521432, 17.7827941, 23.71373706, 31.6227766, 42.16965034, 56.23413252])
    assert dragon.is_irregular
    pid = OneDimBinning("pid", bin_edges=[-1000.0, 0.0, 1000.0])
    assert not pid.is_irregular and pid.is_lin
    inf_pid = OneDimBinning("pid", bin_edges=[-np.inf, 0.55, np.inf])
    assert inf_pid.is_irregular
    over = cz.oversample(10)
    assert over.num_bins == 2000 and over.edge_magnitudes[10] == cz.edge_magnitudes[1]
    vol = MultiDimBinning([cz, pid]).bin_volumes()
    assert vol.shape == (200, 2) and np.isclose(vol[0, 0], 0.01 * 1000)


def test_cfg_grammar_osc_example():
    cfg = parse_pipeline_config("settings/pipeline/osc_example.cfg")
    assert list(cfg)[1:] == [("data", "toy_event_generator"), ("flux", "barr_simple"), ("osc", "prob3")]
    pl = cfg["pipeline"]
    assert pl["name"] == "neutrinos" and pl["output_key"] == "weights"
    assert pl["output_binning"].shape == (200, 200)
    osc = cfg[("osc", "prob3")]
    assert osc["calc_mode"] == pl["output_binning"]
    p = osc["params"].params
    # param_selections = nh picks the nh variants; ${osc:...} interpolation + 'units.' parsing
    assert p.theta23.value.m_as("deg") == 42.0 and not p.theta23.is_fixed
    assert p.theta23.range[1].m_as("deg") == 90.0 and p.theta23.prior.kind == "uniform"
    assert p.deltam31.value.m_as("eV**2") == 2.457e-3
    assert p.theta13.value.m_as("deg") == 8.5 and p.theta13.prior.kind == "gaussian"
    assert p.theta13.prior.stddev.m_as("deg") == 0.205
    assert p.theta12.value.m_as("deg") == 33.48 and p.YeM.value.m == 0.4957
    assert p.detector_depth.value.m_as("km") == 2.0 and p.earth_model.value == "osc/PREM_12layer.dat"
    osc["params"].select_params(["ih"])
    assert osc["params"].params.deltam31.value.m_as("eV**2") == -2.374e-3
    flux = cfg[("flux", "barr_simple")]["params"].params
    # 'nominal + [-5, +5] * sigma'
    np.testing.assert_allclose([r.m for r in flux.delta_index.range], [-0.5, 0.5])
    assert flux.delta_index.prior.kind == "gaussian" and not flux.delta_index.is_fixed
    toy = cfg[("data", "toy_event_generator")]
    assert toy["output_names"][0] == "nue_cc" and len(toy["output_names"]) == 12
    assert toy["params"].params.random.value is False
    q = parse_quantity("1.2 +/- 0.7 * units.meter")
    assert (q.nominal_value, q.std_dev) == (1.2, 0.7) and q.units == ureg.m


def test_param_rescaling_hash_and_priors():
    p = Param("theta23", 42.0 * ureg.deg, prior=Prior("uniform"), range=[0.0, 90.0] * ureg.deg,
              is_fixed=False)
    assert np.isclose(p._rescaled_value, 42.0 / 90.0)
    p._rescaled_value = 0.5
    assert p.value.m_as("deg") == 45.0
    with pytest.raises(ValueError):
        p._rescaled_value = 1.5
    with pytest.raises(ValueError):
        p.value = 100.0 * ureg.deg
    g = Param("x", 1.0, prior=Prior("gaussian", mean=Quantity(1.0), stddev=Quantity(0.5)),
              range=[0.0, 2.0], is_fixed=False)
    ps = ParamSet([p, g, Param("fixed", 3.0)])
    h0 = ps.values_hash
    g.value = Quantity(1.5)
    assert ps.values_hash != h0
    assert np.isclose(ps.priors_penalty("llh"), -0.5)       # -(x-m)^2/(2 s^2), prior.py:249-253
    assert np.isclose(ps.priors_penalty("mod_chi2"), 1.0)   # chi2 = -2 llh
    g.value = Quantity(1.0 + 1e-14)  # below 12 significant figures: same hash as 1.0
    g2 = Quantity(1.0)
    hh = ps.values_hash
    g.value = g2
    assert ps.values_hash == hh
    assert ps.free.names == ("theta23", "x")
    ps.randomize_free(random_state=0)
    rs = np.random.RandomState(0).rand(2)
    assert np.isclose(p.value.m_as("deg"), 90 * rs[0]) and np.isclose(g.value.m, 2 * rs[1])
    ps.reset_free()
    assert p.value.m_as("deg") == 42.0
    sel = ParamSelector(regular_params=[Param("a", 1.0)],
                        selector_param_sets={"nh": [Param("dm", 2.0)], "ih": [Param("dm", -2.0)]},
                        selections=["nh"])
    assert sel.params.dm.value.m == 2.0
    sel.select_params(["ih"])
    assert sel.params.dm.value.m == -2.0 and sel.params.a.value.m == 1.0


def test_stage_param_checks_and_memo():
    from pisa_amd.core.stage import Stage

    class demo(Stage):  # pylint: disable=invalid-name
        def __init__(self, **kw):
            super().__init__(expected_params=("a",), expected_container_keys=(), **kw)
            self.n = 0

        def compute_function(self):
            self.n += 1

    with pytest.raises(ValueError):
        demo(params=ParamSet([Param("b", 1.0)]))
    from pisa_amd.core.container import ContainerSet

    s = demo(params=ParamSet([Param("a", 1.0, is_fixed=False, range=[0, 2])]), calc_mode="events")
    s.data = ContainerSet("x")
    s.setup()
    s.compute(); s.compute()
    assert s.n == 1
    s.params.a.value = Quantity(1.5)
    s.compute()
    assert s.n == 2
    with pytest.raises(ValueError):
        demo(params=ParamSet([Param("a", 1.0)]), calc_mode="nonsense")


def test_data_release_hyperplanes_and_csv_loader_host_side(tmp_path, monkeypatch):
    """SURVEY 8(f) rank 3-4 host logic: hyperplane CSVs of the data release
    (hypersurface.py:2065-2173, linear terms on raw parameter values) and the
    PDG/type selection of data.csv_loader (csv_loader.py:116-146)."""
    import os
    import subprocess
    import sys

    import pandas as pd

    from pisa_amd.core.config_parser import parse_pipeline_config
    from pisa_amd.stages.data.csv_loader import csv_loader
    from pisa_amd.utils import hypersurface as hs
    from pisa_amd.utils.resources import find_resource

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([sys.executable, os.path.join(root, "scripts", "make_synthetic_3y_mc.py"),
                           str(tmp_path), "5000", "1"])
    monkeypatch.setenv("PISA_RESOURCES", str(tmp_path))
    cfg = parse_pipeline_config("settings/pipeline/IceCube_3y_neutrinos.cfg")
    binning = cfg[("utils", "hist")]["apply_mode"]
    assert binning.names == ["reco_energy", "reco_coszen", "pid"] and binning.shape == (8, 8, 2)
    surfaces = hs.load_hypersurfaces("events/IceCube_3y_oscillations/hyperplanes_*.csv.bz2", binning)
    assert list(surfaces) == ["nue_cc+nuebar_cc", "numu_cc+numubar_cc", "nutau_cc+nutaubar_cc", "nu_nc+nubar_nc"]
    h = surfaces["numu_cc+numubar_cc"]
    assert h.param_names == ["ice_absorption", "ice_scattering", "opt_eff_headon", "opt_eff_lateral",
                             "opt_eff_overall"]
    vals = dict(ice_absorption=0.5, ice_scattering=-2.0, opt_eff_headon=0.1, opt_eff_lateral=25.0,
                opt_eff_overall=1.0)
    t = pd.read_csv(find_resource("events/IceCube_3y_oscillations/hyperplanes_numu_cc.csv.bz2"))
    want = t["offset"].values.copy()
    for n in h.param_names:
        want += t[n].values * vals[n]
    np.testing.assert_array_equal(h.evaluate(vals).ravel(), want)
    # at the nominal detector parameters the fitted planes are close to 1
    nominal = dict(ice_absorption=0.0, ice_scattering=0.0, opt_eff_headon=0.0, opt_eff_lateral=25.0,
                   opt_eff_overall=1.0)
    assert abs(np.median(h.evaluate(nominal)) - 1.0) < 0.05

    kw = cfg[("data", "csv_loader")]
    stage = csv_loader(events_file=kw["events_file"], data_dict=kw["data_dict"],
                       output_names=kw["output_names"], calc_mode="events", apply_mode="events")
    from pisa_amd.core.container import ContainerSet

    stage.data = ContainerSet("csv_loader_test")  # what Pipeline._init_stages hands to the first stage
    stage.setup()
    mc = pd.read_csv(os.path.join(str(tmp_path), "events/IceCube_3y_oscillations/neutrino_mc.csv.bz2"))
    assert sum(c.size for c in stage.data) == len(mc)
    c = stage.data["numubar_nc"]
    sel = (mc["pdg"] == -14) & (mc["type"] == 0)
    assert c["nubar"] == -1 and c["flav"] == 1
    np.testing.assert_array_equal(c["weighted_aeff"], mc["weight"].values[sel])
    np.testing.assert_array_equal(c["initial_weights"], np.ones(sel.sum()))


def test_minimizer_settings_formats():
    from pisa_amd.analysis.analysis import load_minimizer_settings

    ref = load_minimizer_settings("settings/minimizer/l-bfgs-b_ftol2e-5_gtol1e-5_eps1e-4_maxiter200.json")
    assert ref["method"] == "L-BFGS-B" and ref["options"]["maxiter"] == 200 and ref["options"]["eps"] == 1e-4
    assert load_minimizer_settings({"method": "SLSQP"}) == {"method": "SLSQP"}
    nested = {"method": {"value": "TNC", "desc": "x"}, "options": {"value": {"maxiter": 3}, "desc": {}}}
    assert load_minimizer_settings(nested) == {"method": "TNC", "options": {"maxiter": 3}}


def _toy_pipeline(name, scale_fixed=False, extra=None):
    """a Pipeline of parameter-only stages (no kernels): what DistributionMaker needs"""
    from collections import OrderedDict

    from pisa_amd.core.pipeline import Pipeline
    from pisa_amd.core.stage import Stage

    class svc(Stage):  # pylint: disable=invalid-name
        def __init__(self, **kw):
            super().__init__(expected_params=[p.name for p in kw["params"]], **kw)

    prm = [Param(name="aeff_scale", value=1.0, prior=None, range=[0.0, 2.0], is_fixed=scale_fixed),
           Param(name="livetime", value=2.5 * ureg.common_year, prior=None, range=None, is_fixed=True)]
    prm += list(extra or [])
    pl = Pipeline.__new__(Pipeline)
    pl.name, pl.detector_name, pl._profile = name, None, False
    pl._stages = [svc(params=ParamSet(prm))]
    pl._config = OrderedDict()
    return pl


def test_distribution_maker_sets_shared_free_params_in_every_pipeline():
    """a free parameter shared by name across pipelines must move in ALL of
    them (distribution_maker.py:183-196, 420-436, 462-476)."""
    from pisa_amd.core.distribution_maker import DistributionMaker

    a = _toy_pipeline("a", extra=[Param(name="only_a", value=3.0, prior=None, range=[0.0, 6.0], is_fixed=False)])
    b = _toy_pipeline("b")
    dm = DistributionMaker([a, b])
    assert dm.params.free.names == ("aeff_scale", "only_a")
    # one Param object per name after construction
    assert a.params.aeff_scale is b.params.aeff_scale
    dm._set_rescaled_free_params([0.25, 0.5])
    assert [p.params.aeff_scale.value.m for p in dm] == [0.5, 0.5]
    assert a.params.only_a.value.m == 3.0
    dm.set_free_params([1.5 * ureg.dimensionless, 1.0 * ureg.dimensionless])
    assert [p.params.aeff_scale.value.m for p in dm] == [1.5, 1.5] and a.params.only_a.value.m == 1.0
    dm.reset_free()
    assert [p.params.aeff_scale.value.m for p in dm] == [1.0, 1.0] and a.params.only_a.value.m == 3.0
    dm.randomize_free_params(random_state=0)
    assert a.params.aeff_scale.value.m == b.params.aeff_scale.value.m != 1.0
    dm.set_nominal_by_current_values()
    v = a.params.aeff_scale.value.m
    dm.set_free_params([0.1 * ureg.dimensionless, 0.2 * ureg.dimensionless])
    dm.reset_all()
    assert b.params.aeff_scale.value.m == v
    # even if the objects were distinct (pipelines built separately and modified later), every
    # pipeline is addressed by name
    b._stages[0]._param_selector.update(Param(name="aeff_scale", value=v, prior=None, range=[0.0, 2.0],
                                              is_fixed=False))
    assert a.params.aeff_scale is not b.params.aeff_scale
    dm._set_rescaled_free_params([0.75, 0.5])
    assert [p.params.aeff_scale.value.m for p in dm] == [1.5, 1.5]


def test_distribution_maker_refuses_param_free_in_one_pipeline_fixed_in_another():
    from pisa_amd.core.distribution_maker import DistributionMaker

    dm = DistributionMaker.__new__(DistributionMaker)
    dm._pipelines, dm.label, dm._profile = [_toy_pipeline("a", scale_fixed=True), _toy_pipeline("b")], None, False
    with pytest.raises(AttributeError):
        dm._set_rescaled_free_params([0.3])
    with pytest.raises(AttributeError):
        dm.set_free_params([0.3 * ureg.dimensionless])


def test_spline_and_linterp_priors():
    """prior.py:262-318: `spline` priors are scipy splev of (knots, coeffs, deg) with ext=2
    (error outside the knots); read from the JSON resource the cfg names, entry
    '<param>_<selector>' (config_parser.py:541-553).  example.cfg's theta23 uses one."""
    import json

    from scipy.interpolate import splev

    from pisa_amd.utils.resources import find_resource

    data = json.load(open(find_resource("priors/nufitv20shiftedtheta23splines.json")))["theta23_nh"]
    pr = Prior(kind="spline", knots=np.array(data["knots"]) * ureg.parse_units(data["units"]),
               coeffs=data["coeffs"], deg=data["deg"])
    for deg in (38.0, 42.3, 47.5, 51.0):
        want = splev(np.deg2rad(deg), (np.array(data["knots"]), np.array(data["coeffs"]), data["deg"]), ext=2)
        assert pr.llh(deg * ureg.degree) == want
        assert pr.chi2(deg * ureg.degree) == -2 * want
    with pytest.raises(ValueError):
        pr.llh(5.0 * ureg.degree)
    p = Param(name="theta23", value=42.3 * ureg.degree, prior=pr, range=[31, 59] * ureg.degree, is_fixed=False)
    assert p.prior_penalty("llh") == pr.llh(42.3 * ureg.degree)
    li = Prior(kind="linterp", param_vals=[0.0, 1.0, 3.0] * ureg.eV, llh_vals=[0.0, -1.0, -9.0])
    assert li.llh(2.0 * ureg.eV) == -5.0
    with pytest.raises(ValueError):
        li.llh(4.0 * ureg.eV)


def test_spline_prior_from_cfg(tmp_path):
    """the reference's own cfg lines for a spline prior (settings/osc/nufitv20.cfg:13-14 through
    ${osc:...} interpolation, as pisa_examples' example.cfg does)"""
    from scipy.interpolate import splev
    import json

    from pisa_amd.utils.resources import find_resource

    text = open(find_resource("settings/pipeline/example_hip.cfg")).read()
    text = text.replace("param.nh.theta23.prior = uniform",
                        "param.nh.theta23.prior = ${osc:theta23_nh.prior}\n"
                        "param.nh.theta23.prior.data = ${osc:theta23_nh.prior.data}")
    path = tmp_path / "spline.cfg"
    path.write_text(text)
    cfg = parse_pipeline_config(str(path))
    p = cfg[("osc", "prob3")]["params"].params.theta23
    assert p.prior.kind == "spline" and p.prior.units == ureg.degree
    d = json.load(open(find_resource("priors/nufitv20shiftedtheta23splines.json")))["theta23_nh"]
    knots_deg = (np.array(d["knots"]) * ureg.parse_units(d["units"])).m_as("deg")
    np.testing.assert_array_equal(p.prior.knots.magnitude, knots_deg)
    assert p.prior_penalty("chi2") == -2 * splev(42.3, (knots_deg, np.array(d["coeffs"]), d["deg"]), ext=2)


def test_hypersurface_forms_state_roundtrip_and_uncertainty(tmp_path):
    """pisa/utils/hypersurface/hypersurface.py:81-205 (functional forms), 356-475 (evaluate with
    the fit covariance), 1182-1283 (state round trip), 1285-1322 (fluctuate)."""
    import json

    from pisa_amd.utils.hypersurface import (HYPERSURFACE_PARAM_FUNCTIONS, Hypersurface, HypersurfaceParam,
                                             load_hypersurfaces)

    assert list(HYPERSURFACE_PARAM_FUNCTIONS) == ["linear", "quadratic", "exponential", "exponential_scaled",
                                                  "logarithmic"]
    rs = np.random.RandomState(3)
    b = MultiDimBinning([OneDimBinning("reco_energy", num_bins=4, is_log=True, domain=[5.0, 80.0]),
                         OneDimBinning("reco_coszen", num_bins=3, is_lin=True, domain=[-1, 1])])
    shape = b.shape
    params = [HypersurfaceParam("dom_eff", "linear", rs.randn(*shape, 1) * 0.1, nominal_value=1.0),
              HypersurfaceParam("hole_ice", "quadratic", rs.randn(*shape, 2) * 0.05, nominal_value=25.0),
              HypersurfaceParam("abs", "exponential", rs.randn(*shape, 1) * 0.1, nominal_value=0.0),
              HypersurfaceParam("scat", "exponential_scaled", rs.randn(*shape, 2) * 0.1, nominal_value=1.0),
              HypersurfaceParam("bulk", "logarithmic", np.abs(rs.randn(*shape, 1)) * 0.1, nominal_value=0.0)]
    n_c = 1 + sum(p.num_fit_coeffts for p in params)
    a = rs.randn(*shape, n_c, n_c) * 0.01
    cov = np.einsum("...ij,...kj->...ik", a, a)
    icpt = 1.0 + rs.randn(*shape) * 0.02
    hsf = Hypersurface(b, params, icpt, log=False, fit_cov_mat=cov)
    assert hsf.num_fit_coeffts == n_c == 8 and hsf.fit_coeffts.shape == shape + (8,)
    vals = dict(dom_eff=1.07, hole_ice=22.0, abs=0.3, scat=0.9, bulk=0.5)
    got, unc = hsf.evaluate(vals, return_uncertainty=True)
    # by hand, bin by bin
    c = {p.name: p.fit_coeffts for p in params}
    d = {n: vals[n] - p.nominal_value for n, p in zip(vals, params)}
    want = (icpt + c["dom_eff"][..., 0] * d["dom_eff"]
            + c["hole_ice"][..., 0] * d["hole_ice"] + c["hole_ice"][..., 1] * d["hole_ice"] ** 2
            + (np.exp(c["abs"][..., 0] * d["abs"]) - 1)
            + (c["scat"][..., 0] + 1) * (np.exp(c["scat"][..., 1] * d["scat"]) - 1)
            + np.log(1 + c["bulk"][..., 0] * d["bulk"]))
    np.testing.assert_allclose(got, want, rtol=1e-14)
    grad = np.stack([np.ones(shape), np.full(shape, d["dom_eff"]), np.full(shape, d["hole_ice"]),
                     np.full(shape, d["hole_ice"] ** 2), d["abs"] * np.exp(c["abs"][..., 0] * d["abs"]),
                     np.exp(c["scat"][..., 1] * d["scat"]) - 1,
                     (c["scat"][..., 0] + 1) * d["scat"] * np.exp(c["scat"][..., 1] * d["scat"]),
                     d["bulk"] / (1 + c["bulk"][..., 0] * d["bulk"])], axis=-1)
    want_unc = np.sqrt(np.einsum("...i,...ij,...j", grad, cov, grad))
    np.testing.assert_allclose(unc, want_unc, rtol=1e-12)
    # at the nominal point only the intercept is left
    np.testing.assert_allclose(hsf.evaluate({p.name: p.nominal_value for p in params}), icpt, rtol=1e-15)
    # log mode exponentiates (and scales the gradient)
    lg = Hypersurface(b, [HypersurfaceParam("dom_eff", "linear", c["dom_eff"], nominal_value=1.0)],
                      icpt - 1.0, log=True, fit_cov_mat=cov[..., :2, :2])
    f, u = lg.evaluate(dict(dom_eff=1.07), return_uncertainty=True)
    np.testing.assert_allclose(f, np.exp(icpt - 1.0 + c["dom_eff"][..., 0] * 0.07), rtol=1e-14)
    g2 = f[..., None] * np.stack([np.ones(shape), np.full(shape, 0.07)], axis=-1)
    np.testing.assert_allclose(u, np.sqrt(np.einsum("...i,...ij,...j", g2, cov[..., :2, :2], g2)), rtol=1e-12)
    # fit-file round trip through load_hypersurfaces
    path = tmp_path / "fits.json"
    from pisa_amd.utils import jsons

    path.write_text(jsons.dumps({"nue_cc+nuebar_cc": hsf.serializable_state, "nu_nc+nubar_nc": lg.serializable_state}))
    loaded = load_hypersurfaces(str(path), expected_binning=b)
    assert list(loaded) == ["nue_cc+nuebar_cc", "nu_nc+nubar_nc"] and loaded["nu_nc+nubar_nc"].log
    g2_, u2_ = loaded["nue_cc+nuebar_cc"].evaluate(vals, return_uncertainty=True)
    np.testing.assert_array_equal(g2_, got)
    np.testing.assert_array_equal(u2_, unc)
    # fluctuate: reproducible for a seed, different from the fit, same shape
    f1 = hsf.fluctuate(np.random.RandomState(5)).evaluate(vals)
    f2 = hsf.fluctuate(np.random.RandomState(5)).evaluate(vals)
    np.testing.assert_array_equal(f1, f2)
    assert np.abs(f1 - got).max() > 0
    with pytest.raises(ValueError):
        Hypersurface(b, [params[0]], icpt).evaluate(dict(dom_eff=1.0), return_uncertainty=True)


def test_param_views_are_cached_and_do_not_count_as_structural_changes():
    """`Pipeline.params` / `DistributionMaker.params` are merged VIEWS of the stages' own sets: asking
    for them must not move `ParamSet.struct_clock` (evaluation plans and the other pipelines' views
    key on it), they are rebuilt only after a real structural change, and a set's name index follows
    extend / replace / update."""
    from pisa_amd.core.distribution_maker import DistributionMaker
    from pisa_amd.core.param import ParamSet

    a, b = _toy_pipeline("a"), _toy_pipeline("b")
    dm = DistributionMaker([a, b])
    clock = ParamSet.struct_clock
    views = (a.params, b.params, dm.params)
    for _ in range(3):                                   # (round 2: every access rebuilt AND bumped)
        assert (a.params, b.params, dm.params) == views
        assert all(x is y for x, y in zip((a.params, b.params, dm.params), views))
    assert ParamSet.struct_clock == clock
    # a real change: a Param object replaced in a stage's own set
    new = Param(name="aeff_scale", value=1.2, prior=None, range=[0.0, 2.0], is_fixed=False)
    b._stages[0]._param_selector.update(new)
    assert ParamSet.struct_clock > clock
    assert b.params is not views[1] and b.params.aeff_scale is new
    assert dm.params is not views[2]
    # the name index of an owned set
    ps = ParamSet(Param(name="x", value=1.0, prior=None, range=None, is_fixed=True))
    ps.extend(Param(name="y", value=2.0, prior=None, range=None, is_fixed=True))
    assert ps.names == ("x", "y") and ps.y.value.m == 2.0 and "z" not in ps
    y2 = Param(name="y", value=5.0, prior=None, range=None, is_fixed=True)
    ps.replace(y2)
    assert ps["y"] is y2 and ps.index("y") == 1
    ps.update([Param(name="z", value=7.0, prior=None, range=None, is_fixed=True), y2])
    assert ps.names == ("x", "y", "z") and ps.z.value.m == 7.0 and ps["y"] is y2
    with pytest.raises(ValueError):
        ps.extend(Param(name="x", value=0.0, prior=None, range=None, is_fixed=True))


def test_interpolated_hypersurfaces(tmp_path):
    """hyper_interpolator.py:48-265, 920-1039: hypersurfaces fitted on a rectilinear grid of oscillation
    parameters (file layout of `fit_hypersurfaces`, quantities as `[magnitude, [[unit, exponent]]]`),
    coefficients and covariances interpolated piecewise-linearly, `scales_log` axes in log10, requests
    clipped to the grid, empty bins (NaN) -> intercept 1 / slopes 0, non-PSD covariances repaired."""
    import json
    from collections import OrderedDict

    from scipy.interpolate import RegularGridInterpolator

    from pisa_amd.utils.hypersurface import (Hypersurface, HypersurfaceParam, frobenius_nearest_psd, is_psd,
                                             load_interpolated_hypersurfaces)

    rs = np.random.RandomState(8)
    b = MultiDimBinning([OneDimBinning("reco_energy", num_bins=3, is_log=True, domain=[5.0, 80.0]),
                         OneDimBinning("reco_coszen", num_bins=2, is_lin=True, domain=[-1, 1])])
    shape = b.shape
    dm_vals, th_vals = [1.0e-3, 2.0e-3, 4.0e-3, 8.0e-3], [35.0, 45.0, 55.0]
    fits, coeff, covs = [], {}, {}
    for i, dm in enumerate(dm_vals):
        for j, th in enumerate(th_vals):
            maps = OrderedDict()
            for name in ("nue_cc+nuebar_cc", "nu_nc+nubar_nc"):
                lin = rs.randn(*shape, 1) * 0.1
                quad = rs.randn(*shape, 2) * 0.05
                icpt = 1.0 + rs.randn(*shape) * 0.02
                if name.startswith("nu_nc") and (i, j) == (1, 1):
                    icpt[0, 0] = np.nan                      # an empty bin at one grid point
                a = rs.randn(*shape, 4, 4) * 0.02
                cov = np.einsum("...ij,...kj->...ik", a, a)
                h = Hypersurface(b, [HypersurfaceParam("dom_eff", "linear", lin, nominal_value=1.0),
                                     HypersurfaceParam("hole_ice", "quadratic", quad, nominal_value=25.0)],
                                 icpt, fit_cov_mat=cov)
                maps[name] = h.serializable_state
                coeff[name, i, j], covs[name, i, j] = h.fit_coeffts, cov
            fits.append({"param_values": {"deltam31": [dm, [["electron_volt", 2.0]]], "theta23": [th, [["degree", 1.0]]]},
                         "hs_fit": maps})
    spec = OrderedDict([("deltam31", {"values": [[v, [["electron_volt", 2.0]]] for v in dm_vals], "scales_log": True}),
                        ("theta23", {"values": [[v, [["degree", 1.0]]] for v in th_vals], "scales_log": False})])
    path = tmp_path / "interp.json"
    from pisa_amd.utils import jsons

    path.write_text(jsons.dumps({"interpolation_param_spec": spec, "hs_fits": fits}))
    loaded = load_interpolated_hypersurfaces(str(path), expected_binning=b)
    assert list(loaded) == ["nue_cc+nuebar_cc", "nu_nc+nubar_nc"]
    hi = loaded["nue_cc+nuebar_cc"]
    assert hi.interpolation_param_names == ["deltam31", "theta23"] and hi.param_names == ["dom_eff", "hole_ice"]
    # at a grid point: the stored fit
    at = hi.get_hypersurface(deltam31=2.0e-3 * ureg.eV ** 2, theta23=55.0 * ureg.degree)
    np.testing.assert_allclose(at.fit_coeffts, coeff["nue_cc+nuebar_cc", 1, 2], rtol=1e-13)
    # in between: linear in (log10 deltam31, theta23), also through other units of the request
    cz = np.stack([np.stack([coeff["nue_cc+nuebar_cc", i, j] for j in range(3)]) for i in range(4)])
    ref = RegularGridInterpolator([np.log10(dm_vals), th_vals], cz)
    mid = hi.get_hypersurface(deltam31=2.9e-3 * ureg.eV ** 2, theta23=(np.deg2rad(41.0)) * ureg.rad)
    np.testing.assert_allclose(mid.fit_coeffts, ref([np.log10(2.9e-3), 41.0])[0], rtol=1e-12)
    vals = dict(dom_eff=1.05, hole_ice=23.0)
    want = (mid.fit_coeffts[..., 0] + mid.fit_coeffts[..., 1] * 0.05 + mid.fit_coeffts[..., 2] * -2.0
            + mid.fit_coeffts[..., 3] * 4.0)
    got, unc = mid.evaluate(vals, return_uncertainty=True)
    np.testing.assert_allclose(got, want, rtol=1e-13)
    assert np.all(unc > 0) and all(is_psd(m) for m in mid.fit_cov_mat.reshape(-1, 4, 4))
    # outside the grid: clipped to its bounds
    out = hi.get_hypersurface(deltam31=1.0 * ureg.eV ** 2, theta23=10.0 * ureg.degree)
    np.testing.assert_allclose(out.fit_coeffts, coeff["nue_cc+nuebar_cc", 3, 0], rtol=1e-13)
    # the empty bin: a NaN coefficient anywhere in the interpolation cell -> that coefficient becomes its
    # default (intercept 1, slopes 0), element by element as in the reference (:252-257)
    nc = loaded["nu_nc+nubar_nc"].get_hypersurface(deltam31=2.5e-3 * ureg.eV ** 2, theta23=46.0 * ureg.degree)
    assert nc.fit_coeffts[0, 0, 0] == 1.0 and np.all(np.isfinite(nc.fit_coeffts))
    assert np.all(np.isfinite(nc.evaluate(vals)))
    with pytest.raises(AssertionError):
        hi.get_hypersurface(deltam31=2e-3 * ureg.eV ** 2)
    # the covariance repair
    m = np.array([[1.0, 2.0], [2.0, 1.0]])
    assert not is_psd(m)
    fixed = frobenius_nearest_psd(m)
    assert is_psd(fixed) and np.allclose(fixed, fixed.T) and np.abs(fixed - m).max() < 1.1


def test_legacy_hyperplane_fit_files(tmp_path):
    """hypersurface.py:1967-2062: fit files of pre-hypersurface PISA versions (`sys_list`, `map_names`, one
    [binning..., 1 + n_sys] array per map, in either of the two layouts); linear, evaluated with the raw
    parameter values because such files carry no nominal values"""
    import json

    from pisa_amd.utils.hypersurface import load_hypersurfaces

    rs = np.random.RandomState(1)
    b = MultiDimBinning([OneDimBinning("reco_energy", num_bins=4, is_log=True, domain=[5.0, 80.0]),
                         OneDimBinning("reco_coszen", num_bins=3, is_lin=True, domain=[-1, 1])])
    arrs = {m: np.concatenate([1.0 + rs.randn(4, 3, 1) * 0.02, rs.randn(4, 3, 2) * 0.1], axis=-1)
            for m in ("nue_cc", "numu_cc")}
    vals = dict(dom_eff=1.1, hole_ice=0.4)
    for layout in (0, 1):
        data = {"sys_list": ["dom_eff", "hole_ice"], "map_names": list(arrs)}
        if layout == 0:
            data.update({m: a.tolist() for m, a in arrs.items()})
        else:
            data["hyperplanes"] = {m: {"fit_params": a.tolist()} for m, a in arrs.items()}
        path = tmp_path / ("legacy%d.json" % layout)
        path.write_text(json.dumps(data))
        loaded = load_hypersurfaces(str(path), expected_binning=b)
        assert list(loaded) == ["nue_cc", "numu_cc"]
        for m, a in arrs.items():
            assert loaded[m].using_legacy_data and loaded[m].param_names == ["dom_eff", "hole_ice"]
            np.testing.assert_allclose(loaded[m].evaluate(vals), a[..., 0] + a[..., 1] * 1.1 + a[..., 2] * 0.4,
                                       rtol=1e-15)
    with pytest.raises(AssertionError):
        load_hypersurfaces(str(path), expected_binning=MultiDimBinning(
            [OneDimBinning("reco_energy", num_bins=5, is_log=True, domain=[5.0, 80.0]),
             OneDimBinning("reco_coszen", num_bins=3, is_lin=True, domain=[-1, 1])]))


def test_device_side_in_place_rewrite_is_announced_like_a_host_side_one():
    """a kernel that rewrites a variable's DEVICE array in place (the one-launch flux stage) calls
    `mark_dev_changed`: the host mirror is what went stale, every other representation is invalidated and
    the per-key version / per-container write counters move -- the counterpart of the reference's
    `container[key][...] = ...; container.mark_changed(key)` (container.py:638-649) for host arrays"""
    import torch

    from pisa_amd.core.container import Container

    c = Container("nue_cc")
    t = torch.arange(6, dtype=torch.float64).reshape(3, 2)     # stands in for a device tensor
    c["nu_flux"] = t
    assert np.array_equal(c["nu_flux"], np.arange(6.0).reshape(3, 2))   # host mirror fetched
    v0, w0 = c.version("nu_flux"), c.writes
    arr = c.current_data["nu_flux"]
    arr.host = arr.host.copy()       # (a CPU tensor and its numpy view share memory: make it a real mirror)
    t.mul_(2.0)                                                 # "kernel" rewrites the array in place
    assert np.array_equal(c["nu_flux"], np.arange(6.0).reshape(3, 2))   # stale mirror until announced
    c.mark_dev_changed("nu_flux")
    assert arr.dev_valid and not arr.host_valid
    assert np.array_equal(c["nu_flux"], 2.0 * np.arange(6.0).reshape(3, 2))
    assert c.device("nu_flux") is t
    assert c.version("nu_flux") == v0 + 1 and c.writes == w0 + 1
    # host-side edit, for symmetry: the device copy is what goes stale
    arr.host = arr.host.copy()
    c["nu_flux"][0, 0] = 7.0
    c.mark_changed("nu_flux")
    assert arr.host_valid and not arr.dev_valid
    assert c.version("nu_flux") == v0 + 2


def test_forward_stencil_is_scipys_two_point_scheme():
    """`Analysis._forward_stencil` (the points of a gradient that go through one sweep) reproduces
    scipy's own '2-point' finite differences inside bounds -- points, steps and quotients -- including
    steps flipped or shrunk at the bounds and steps too small to move x"""
    from scipy.optimize._numdiff import approx_derivative

    from pisa_amd.analysis.analysis import Analysis

    rs = np.random.RandomState(0)
    a = rs.rand(4, 4)
    a = a @ a.T
    calls = []

    def f(x):
        calls.append(np.array(x))
        return float(x @ a @ x + np.sin(x).sum())

    lb, ub = np.zeros(4), np.ones(4)
    for x0 in (rs.rand(4), np.array([1.0, 0.0, 0.99995, 0.5]), np.array([0.99999999, 1e-9, 0.3, 1.0]),
               np.array([0.5, 0.99996, 0.00004, 1.0 - 1e-12])):
        for eps in (1e-4, 1e-8, 1e-20):
            calls.clear()
            g = approx_derivative(f, x0, method="2-point", abs_step=eps, bounds=(lb, ub))
            asked = [c.copy() for c in calls]
            pts, dx = Analysis._forward_stencil(x0, eps, lb, ub)
            assert len(asked) == len(pts) and all(np.array_equal(p, q) for p, q in zip(asked, pts))
            vals = np.array([f(p) for p in pts])
            np.testing.assert_array_equal((vals[1:] - vals[0]) / dx, g)


def test_kde_map_postprocessing_of_a_stack_equals_map_by_map():
    """`kde_hist._finish_hist_many` (reflection at coszen = -1 / +1 without concatenations, bin volumes, block sums of the
    oversampling, vectorised over the estimators of an evaluation) gives, bit for bit, what `_finish_hist` -- the
    reference's sequence (kde_hist.py:168-217) -- gives map by map: both reflections, one, none, coszen on either axis"""
    from pisa_amd.core.binning import MultiDimBinning, OneDimBinning
    from pisa_amd.utils import kde_hist as kh

    for names in (("energy", "coszen"), ("coszen", "energy")):
        for dom in ([-1, 1], [-1, 0.5], [-0.5, 1], [-0.8, 0.8]):
            dims = dict(energy=OneDimBinning("energy", domain=[0.5, 3.5], num_bins=6, is_lin=True),
                        coszen=OneDimBinning("coszen", domain=dom, num_bins=8, is_lin=True))
            b = MultiDimBinning([dims[n] for n in names])
            for refl, over in ((0.25, 5), (0.5, 1), (0.25, 1)):
                g = kh._evaluation_grid(b, over, "coszen", refl)
                d = np.random.RandomState(0).rand(7, *g["megashape"])
                many = kh._finish_hist_many(g, d, over)
                for k in range(7):
                    one, _ = kh._finish_hist(g, d[k], None, over)
                    np.testing.assert_array_equal(one, many[k])


def test_param_set_deepcopy_is_independent():
    """`HypoFitResult` snapshots the parameter set after every fit (analysis.py:356-372): the copy has Param and Prior
    objects, range lists and array magnitudes of its own; moving or fixing a copied parameter leaves the original (and the
    process-wide change counters' consumers) alone."""
    import copy

    from pisa_amd.core.param import Param, ParamSet, Prior
    from pisa_amd.core.units import ureg

    ps = ParamSet([Param(name="p%d" % i, value=(1.0 + i) * ureg.degree,
                         prior=Prior(kind="gaussian", mean=1.0 * ureg.degree, stddev=0.5 * ureg.degree),
                         range=[0 * ureg.degree, 90 * ureg.degree], is_fixed=i % 2 == 0) for i in range(6)])
    ps.extend(Param(name="arr", value=np.arange(3.0) * ureg.m, is_fixed=True))
    ps.extend(Param(name="flag", value=True, is_fixed=True))
    c = copy.deepcopy(ps)
    assert c.names == ps.names and [p.is_fixed for p in c] == [p.is_fixed for p in ps]
    assert all(a is not b and a.value == b.value for a, b in zip(c, ps) if a.name != "arr")
    c.p1.value = 5 * ureg.degree
    c.p1.range = [1 * ureg.degree, 10 * ureg.degree]
    c.fix("p1")
    assert ps.p1.value.m_as("degree") == 2.0 and not ps.p1.is_fixed and ps.p1.range[1].m_as("degree") == 90
    assert c.p1.prior is not ps.p1.prior and c.p1.prior.mean == ps.p1.prior.mean
    assert c.p1.prior_penalty("llh") != ps.p1.prior_penalty("llh")
    c["arr"].value.magnitude[0] = 9.0
    assert ps["arr"].value.magnitude[0] == 0.0
    assert c.flag.value is True
    # the free / fixed views of the copy are the copy's objects
    assert all(p is c[p.name] for p in c.free)


def test_priors_penalty_keeps_terms_and_follows_every_change():
    """`ParamSet.priors_penalty` keeps its terms in an array and re-evaluates those of moved parameters only: the
    same number as the plain sum of `prior_penalty` after value changes, a prior exchanged, a parameter without prior,
    another metric, and for a set built later from the same Param objects."""
    from pisa_amd.core.param import Param, ParamSet, Prior
    from pisa_amd.core.units import ureg

    rs = np.random.RandomState(3)
    ps = ParamSet([Param(name="p%d" % i, value=(1.0 + i) * ureg.degree,
                         prior=Prior(kind="gaussian", mean=1.3 * ureg.degree, stddev=0.5 * ureg.degree) if i % 3 else None,
                         range=[0 * ureg.degree, 90 * ureg.degree], is_fixed=False) for i in range(12)])
    plain = lambda s_, m: np.sum([p.prior_penalty(m) for p in s_])
    for step in range(40):
        m = "llh" if step % 5 else "mod_chi2"
        k = int(rs.randint(12))
        ps["p%d" % k].value = float(rs.uniform(0.5, 80.0)) * ureg.degree
        if step == 17:
            ps.p4.prior = Prior(kind="gaussian", mean=20 * ureg.degree, stddev=3 * ureg.degree)
        if step == 23:
            ps.p5.prior = None
        assert ps.priors_penalty(m) == plain(ps, m)
    view = ParamSet([ps.p1, ps.p4, ps.p7])
    ps.p4.value = 33 * ureg.degree
    assert view.priors_penalty("llh") == plain(view, "llh") and ps.priors_penalty("llh") == plain(ps, "llh")


def test_quantity_truth_value_is_its_magnitudes():
    """`if param.value:` (toy_event_generator.py:80 on `random`) -- pint's rule: the magnitude decides"""
    from pisa_amd.core.units import ureg

    assert bool(1 * ureg.dimensionless) and bool(2.5 * ureg.GeV)
    assert not bool(0 * ureg.dimensionless) and not bool(0.0 * ureg.m)


def test_container_set_reference_unit_test():
    """pisa/core/container.py:1141-1189 `test_container_set`: duplicate containers are refused, shared keys with and
    without representation independence, auxiliary data count as keys in every representation"""
    from pisa_amd import FTYPE
    from pisa_amd.core.container import Container, ContainerSet

    c1, c2 = Container("test1"), Container("test2")
    data = ContainerSet("data", [c1, c2])
    with pytest.raises(ValueError):
        data.add_container(c1)
    n = 10
    c1["true_energy"] = np.linspace(1, 80, n, dtype=FTYPE)
    c2["reco_coszen"] = np.linspace(-1, 1, 2 * n, dtype=FTYPE)
    for rep_indep in (True, False):
        assert len(data.get_shared_keys(rep_indep=rep_indep)) == 0
    c1["reco_coszen"] = c2["reco_coszen"][:n]
    for rep in (None, "events"):
        data.representation = rep
        for rep_indep in (True, False):
            shared = data.get_shared_keys(rep_indep=rep_indep)
            if rep_indep or rep is None:
                assert tuple(shared) == ("reco_coszen",)
            else:
                for c in data.containers:
                    assert len(c.keys) == 0
                assert len(shared) == 0
    key = "AmIEvil"
    c1.set_aux_data(key=key, val=False)
    c2.set_aux_data(key=key, val=True)
    indep, dep = data.get_shared_keys(rep_indep=True), data.get_shared_keys(rep_indep=False)
    assert key in indep and key in dep
    assert len(indep) == 2 and len(dep) == 1


def test_find_index_is_numpys_bin_rule():
    """`translation.find_index` (pisa/core/translation.py:504-553) by the reference's own recipe (:821-942): the bin
    `np.histogramdd` counts a value in; -1 below the range or for NaN, `num_bins` above; infinite outer edges"""
    from pisa_amd.core.translation import find_index

    eps = np.finfo(float).eps
    for basic in ([-1, -0.5, -0.1, 0, 0.1, 0.5, 1, 2, 3, 4], [], [0.1], [-0.1, 0.1]):
        for lo, hi in ((None, None), (-np.inf, None), (None, np.inf), (-np.inf, np.inf)):
            edges = ([] if lo is None else [lo]) + list(basic) + ([] if hi is None else [hi])
            if len(edges) < 2:
                continue
            edges = np.array(edges, dtype=float)
            n = len(edges) - 1
            inside = [(a + b) / 2 if np.isfinite(a) and np.isfinite(b) else a + 10.5 if np.isfinite(a) else
                      b - 10.5 if np.isfinite(b) else 10.5 for a, b in zip(edges[:-1], edges[1:])]
            with np.errstate(invalid="ignore"):
                vals = np.concatenate([[-np.inf, np.inf, np.nan], edges, inside, (1 - eps) * edges, (1 + eps) * edges])
            want = []
            for v in vals:
                hit = np.nonzero(np.histogramdd([v], np.atleast_2d(edges))[0])[0]
                want.append(-1 if (np.isnan(v) or v < edges[0]) else n if v > edges[-1] else int(hit[0]))
            assert [find_index(v, edges) for v in vals] == want
            assert find_index(vals, edges).tolist() == want


def test_hdf5_reader_and_events_pi():
    """`utils/hdf.py` on the reference's toy events file (an HDF5 file written by h5py with PISA's `to_hdf`: old-style
    groups, chunked + byte-shuffled float64 arrays): structure, sizes, value ranges, selective reading; `EventsPi`:
    flavour / interaction groups, renamed and stacked variables, the "oppo" flux fix, cuts, reproducible sub-samples."""
    from pisa_amd.core.events_pi import EventsPi
    from pisa_amd.utils.hdf import from_hdf

    path = ("events/events__vlvnt__toy_1_to_80GeV_spidx1.0_cz-1_to_1_1e2evts_set0__unjoined__with_fluxes_"
            "honda-2015-spl-solmin-aa.hdf5")
    data = from_hdf(path)
    assert list(data) == ["nue", "nue_bar", "numu", "numu_bar", "nutau", "nutau_bar"]
    assert all(list(g) == ["cc", "nc"] for g in data.values())
    cols = data["numu_bar"]["nc"]
    assert sorted(cols) == ["neutrino_nue_flux", "neutrino_numu_flux", "neutrino_oppo_nue_flux", "neutrino_oppo_numu_flux",
                            "pid", "reco_coszen", "reco_energy", "true_coszen", "true_energy", "weighted_aeff"]
    for g in data.values():
        for sub in g.values():
            assert all(a.shape == (100,) and a.dtype == np.float64 and np.all(np.isfinite(a)) for a in sub.values())
            assert 1.0 <= sub["true_energy"].min() and sub["true_energy"].max() <= 80.0
            assert -1.0 <= sub["true_coszen"].min() and sub["true_coszen"].max() <= 1.0
            assert set(np.unique(sub["pid"])) <= {-1.0, 1.0} and np.all(sub["weighted_aeff"] >= 0)
            assert sub["neutrino_numu_flux"].mean() > sub["neutrino_nue_flux"].mean() > 0     # atmospheric: more numu than nue
    assert len(set(float(g[s]["true_energy"].sum()) for g in data.values() for s in g)) == 12   # twelve different samples
    few = from_hdf(path, choose=["pid", "true_energy"])
    assert sorted(few["nue"]["cc"]) == ["pid", "true_energy"] and np.array_equal(few["nue"]["cc"]["pid"], data["nue"]["cc"]["pid"])
    node = from_hdf(path, return_node="/nutau_bar/cc")
    assert np.array_equal(node["reco_energy"], data["nutau_bar"]["cc"]["reco_energy"])
    with pytest.raises(KeyError):
        from_hdf(path, return_node="/nutau_bar/dis")

    mapping = {"true_energy": "true_energy", "pid": "pid", "nu_flux_nominal": ["nominal_nue_flux", "nominal_numu_flux"],
               "nubar_flux_nominal": ["nominal_nuebar_flux", "nominal_numubar_flux"]}
    ev = EventsPi(name="Events")
    ev.load_events_file(path, variable_mapping=mapping)
    assert sorted(ev) == sorted("%s%s_%s" % (f, b, i) for f in ("nue", "numu", "nutau") for b in ("", "bar") for i in ("cc", "nc"))
    nu, nubar = ev["numu_cc"], ev["numubar_cc"]
    assert nu["nu_flux_nominal"].shape == (100, 2) and list(nu) == list(mapping)
    assert np.array_equal(nu["nu_flux_nominal"][:, 0], data["numu"]["cc"]["neutrino_nue_flux"])
    assert np.array_equal(nu["nubar_flux_nominal"][:, 1], data["numu"]["cc"]["neutrino_oppo_numu_flux"])
    assert np.array_equal(nubar["nu_flux_nominal"][:, 0], data["numu_bar"]["cc"]["neutrino_oppo_nue_flux"])   # the other sign's
    assert np.array_equal(nubar["nubar_flux_nominal"][:, 1], data["numu_bar"]["cc"]["neutrino_numu_flux"])
    cut = ev.apply_cut("(true_energy <= 70) & (np.abs(pid) > 0)")
    assert all(np.all(g["true_energy"] <= 70) for g in cut.values()) and 0 < len(cut["nue_nc"]["pid"]) < 100
    assert cut.metadata["cuts"] == ["(true_energy <= 70) & (np.abs(pid) > 0)"] and ev.metadata["cuts"] == []
    assert cut.apply_cut("(true_energy <= 70) & (np.abs(pid) > 0)") is cut
    with pytest.raises(KeyError):
        EventsPi().load_events_file(path, variable_mapping={"x": "no_such_variable"})
    # three statistically independent quarters of the sample, the same for the same seed
    parts = []
    for k in range(3):
        sub = EventsPi(fraction_events_to_keep=0.25, events_subsample_index=k)
        sub.load_events_file(path, variable_mapping=mapping, seed=7)
        parts.append(sub["nue_cc"]["true_energy"])
        assert parts[-1].shape == (25,) and sub["nue_cc"]["nu_flux_nominal"].shape == (25, 2)
    assert len(set(np.concatenate(parts))) == 75
    again = EventsPi(fraction_events_to_keep=0.25, events_subsample_index=1)
    again.load_events_file(path, variable_mapping=mapping, seed=7)
    assert np.array_equal(again["nue_cc"]["true_energy"], parts[1])
    with pytest.raises(AssertionError):
        EventsPi(fraction_events_to_keep=0.25, events_subsample_index=4)


def test_sqlite_loader_reads_the_reference_test_database(tmp_path):
    """data.sqlite_loader on the ten-event database the reference's init_test writes (sqlite_loader.py:152-177): rows
    selected by PDG code and interaction type, truth and reconstruction joined by event number, weighted_aeff"""
    from pisa_amd.core.container import ContainerSet
    from pisa_amd.stages.data.sqlite_loader import sqlite_loader, write_test_database

    path = str(tmp_path / "events.db")
    truth, reco = write_test_database(path)
    st = sqlite_loader(database=path, output_names=["numu_cc", "numubar_cc"], data=ContainerSet("data"), calc_mode="events",
                       apply_mode="events")
    assert st.get_pid_and_interaction_type("nutaubar_nc") == (-16, 2, -1, 2)
    assert st.get_pid_and_interaction_type("nue_cc") == (12, 1, 1, 0)
    st.setup()
    st.run()
    c, empty = st.data.containers
    t, r = np.array(truth), np.array(reco)
    assert c.size == 10 and empty.size == 0 and c["nubar"] == 1 and c["flav"] == 1
    assert np.array_equal(c["true_energy"], t[:, 0]) and np.array_equal(c["true_coszen"], np.cos(t[:, 1]))
    assert np.array_equal(c["reco_energy"], r[:, 0]) and np.array_equal(c["reco_coszen"], np.cos(r[:, 1]))
    assert np.array_equal(c["pid"], r[:, 2]) and np.array_equal(c["weights"], np.ones(10))
    assert np.array_equal(c["weighted_aeff"], 1e-4 * t[:, 2] / 1 / t[:, 3] / 10)


def test_partition_accounting_adds_up():
    """`engine._partition_accounting` (round 6: the host side shared by the torch and the native form of the partitioned
    resident order): for random partition populations the depositing blocks are the events topped up to whole blocks of 256,
    the idle blocks dealt to the partitions add up to what the container has, nothing is used twice, and a container
    without enough idle events to top its partitions up is refused (None)."""
    from pisa_amd.engine import _partition_accounting

    rs = np.random.RandomState(6)
    seen_none = seen_aligned = 0
    for _ in range(300):
        n_part = int(rs.randint(2, 9))
        n_dep = [int(v) for v in rs.randint(0, 40000, size=n_part) * (rs.rand(n_part) < 0.8)]
        n_idle = int(rs.randint(0, 90000))
        n = sum(n_dep) + n_idle
        n_wg = int(rs.randint(1, 40)) if rs.rand() < 0.7 else None
        acc = _partition_accounting(n, n_dep, n_idle, 256, n_wg)
        top_need = sum((-k) % 256 for k in n_dep)
        if n_idle < top_need or n // 256 == 0:
            assert acc is None
            seen_none += 1
            continue
        top, dep_blocks, share = acc
        assert all(0 <= t < 256 and (k + t) % 256 == 0 and b == (k + t) // 256 for k, t, b in zip(n_dep, top, dep_blocks))
        assert all(s >= 0 for s in share)
        used_idle = sum(top) + 256 * sum(share)
        assert used_idle <= n_idle
        if sum(n_dep) > 0:      # (a container in which nothing deposits keeps its idle events in the tail: any order serves)
            assert n_idle - used_idle < 512                # what is left is the tail behind the last partition
            assert sum(dep_blocks) + sum(share) in (n // 256, n // 256 - 1)
        seen_aligned += n_wg is not None
    assert seen_none > 0 and seen_aligned > 50
