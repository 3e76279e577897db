"""The restatement of the services around the path (`oracle/stages_oracle.py`) against the reference's own functions
executed in the build container (`oracle/gen_golden.py side` -> tests/golden/side_stages_ref.npz), the host-side
pieces of those services, and the arrays of the reference's `test_lookup_indices` (bin_indexing.py:164-226)."""
import os

import numpy as np
import pytest

from oracle import stages_oracle as so

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "side_stages_ref.npz"))


def test_two_nu_osc_restatement():
    for ic, (t23, dm31) in enumerate(G["two_params"]):
        for flav, tag in ((0, "nue"), (1, "numu"), (2, "nutau")):
            got = so.two_nu_weights(G["two_flux"], t23, dm31, G["two_e"], G["two_cz"], flav, G["two_w0"])
            np.testing.assert_allclose(got, G["two_%d_%s" % (ic, tag)], rtol=1e-13, atol=1e-15)


def test_power_law_and_genie_restatements():
    for ic, (delta, norm) in enumerate(G["astro_params"]):
        got = so.power_law(G["astro_e"], 100.0e3, delta, norm, G["astro_nominal"])
        np.testing.assert_allclose(got, G["astro_%d" % ic], rtol=1e-15, atol=0)
    np.testing.assert_allclose(so.power_law(G["astro_e"], 100.0e3, -2.5, 0.787e-18), G["astro_nominal"], rtol=1e-15)
    for ic, ps in enumerate(G["genie_params"]):
        for k in (1, 2, 3):
            got = so.poly_scale(ps[:k], G["genie_lin"][:k], G["genie_quad"][:k], G["genie_w0"])
            assert np.array_equal(got, G["genie_%d_%d" % (ic, k)])
    assert (G["genie_3_3"] == 0).any() and (G["genie_3_3"] > 0).any()       # the clamp at 0 is exercised


def test_lookup_indices_restatement():
    edges = [G["idx_edges0"], G["idx_edges1"], G["idx_edges2"]]
    for nd in (1, 2, 3):
        got = so.lookup_indices(list(G["idx_cols"][:nd]), edges[:nd])
        assert np.array_equal(got, G["idx_%dd" % nd])
    e = [np.linspace(0, 7, 8), np.linspace(0, 4, 5), np.linspace(0, 2, 3)]
    cols = [G["idx_test_x"], G["idx_test_y"], G["idx_test_z"]]
    for nd in (1, 2, 3):
        assert np.array_equal(so.lookup_indices(cols[:nd], e[:nd]), G["idx_test_%dd" % nd])


def test_kfold_folds_are_scikit_learns():
    from sklearn.model_selection import KFold

    from pisa_amd.stages.utils.kfold import _fold

    for n in (10, 11, 13):
        for k in (2, 3, 5):
            for sel in range(k + 2):
                for shuffle, seed in ((False, None), (True, 3)):
                    for i, (_, test) in enumerate(KFold(n_splits=k, shuffle=shuffle, random_state=seed).split(np.empty(n))):
                        if i == sel:
                            break
                    assert np.array_equal(test, _fold(n, k, sel, shuffle, seed)), (n, k, sel, shuffle)
    with pytest.raises(ValueError):
        _fold(3, 5, 0, False, None)


def test_bootstrap_insertion_and_angle_as_dimensionless():
    from collections import OrderedDict

    from pisa_amd.core.units import DimensionalityError, ureg
    from pisa_amd.stages.utils.bootstrap import insert_bootstrap_after_data_loader

    cfg = OrderedDict([(("data", "simple_data_loader"), {"a": 1}), (("flux", "barr_simple"), {}), (("utils", "hist"), {})])
    out = insert_bootstrap_after_data_loader(cfg, seed=4)
    assert list(out) == [("data", "simple_data_loader"), ("utils", "bootstrap"), ("flux", "barr_simple"), ("utils", "hist")]
    assert out[("utils", "bootstrap")] == {"apply_mode": "events", "calc_mode": "events", "seed": 4} and list(cfg) != list(out)
    # pint's radian is dimensionless: two_nu_osc.py:68 reads theta23 this way
    assert (45 * ureg.degree).m_as("dimensionless") == np.deg2rad(45.0)
    with pytest.raises(DimensionalityError):
        (1 * ureg.m).m_as("dimensionless")


def test_decoherence_restatement():
    for ic, (t12, t13, t23, dm21, dm31, g21, g31, g32) in enumerate(G["dec_params"]):
        u2 = so.tau_row_sq(*(np.arcsin(np.sin(t)) for t in (t12, t13, t23)))
        coef = [u2[1] * u2[0], u2[2] * u2[0], u2[2] * u2[1]]
        dis = so.decoherence_disappearance(coef, [g21, g31, g32], [dm21, dm31, dm31 - dm21], G["dec_e"], G["dec_l"])
        table = so.decoherence_table(dis)
        assert np.array_equal(table[:, 0, :], G["dec_%d_nue" % ic])
        np.testing.assert_allclose(table[:, 1, :], G["dec_%d_numu" % ic], rtol=1e-15, atol=1e-16)
    assert G["dec_0_numu"][:, 1].min() < 0.6 and (G["dec_0_numu"][:, 0] == 0).all()


def test_linear_interpolant_of_atm_muons_is_numpy_interp():
    """scipy's interp1d(kind='linear') on 1-D float data evaluates numpy.interp: the device kernel restates the latter"""
    from scipy.interpolate import interp1d

    from pisa_amd.stages.background.atm_muons import init_test

    xk, yk = init_test(prior=None, range=None, is_fixed=True)._make_prim_unc_spline()
    assert xk[0] == 0.0 and xk[-1] == 1.0 and len(xk) == 22 and not (yk == 0).any()
    x = np.concatenate([np.random.RandomState(0).rand(1000), xk])
    assert np.array_equal(interp1d(xk, yk, kind="linear")(x), np.interp(x, xk, yk))
    with pytest.raises(ValueError):
        interp1d(xk, yk, kind="linear")(1.5)


def test_wide_metric_restatements():
    """oracle/stages_oracle.metric_wide against the reference's stats functions (oracle/gen_golden.py stats_wide)"""
    W = np.load(os.path.join(os.path.dirname(__file__), "golden", "stats_wide_ref.npz"))
    for kind in ("mcllh_mean", "mcllh_eff", "correct_chi2", "signed_sqrt_mod_chi2", "conv_llh"):
        got = so.metric_wide(kind, W["actual"], W["expected"], W["sigma"])
        np.testing.assert_allclose(got, W[kind], rtol=1e-13, atol=1e-13, equal_nan=True, err_msg=kind)
    # the Poisson limit of the mixture where sigma = 0
    from scipy.special import gammaln

    k, lam = W["actual"][6:12], W["expected"][6:12]
    np.testing.assert_allclose(W["mcllh_eff"][6:12], k * np.log(lam) - lam - gammaln(k + 1), rtol=1e-14)


def test_simple_param_stage_reproduces_the_reference_draws():
    """reco.simple_param is event preparation on the host (numpy's RandomState(0) stream, the reference's order of
    draws): the whole stage against the reference's three functions executed in the build container, container
    after container from one generator; then the `perfect_reco` switch"""
    from pisa_amd.core.container import Container, ContainerSet
    from pisa_amd.core.param import Param, ParamSet
    from pisa_amd.stages.reco.simple_param import dict_lookup_wildcard, simple_param

    names = ("numu_cc", "nutau_cc", "nue_nc", "muons", "numubar_cc")

    def make(perfect):
        cs = []
        for name in names:
            c = Container(name)
            c["true_energy"], c["true_coszen"] = G["reco_e"].copy(), G["reco_cz"].copy()
            cs.append(c)
        values = [("perfect_reco", perfect),
                  ("reco_energy_params", "{'nu*_cc': [10., 0.3, -0.2], '*_nc': [10., 0.5, 0.1], 'muons': [5., 0.6, 0.]}"),
                  ("reco_coszen_params", "{'nu*_cc': [10., 0.4, -0.5], '*_nc': [10., 0.6, -0.3], 'muons': [5., 0.1, 0.]}"),
                  ("pid_track_params", "{'numu*_cc': [0.9, 0.3, 12.], 'nue*': [0.3, 0.1, 30.], 'nutau*': [0.3, 0.1, 30.], 'muons': [1., 1., 0.]}"),
                  ("track_pid", 1.0), ("cascade_pid", 0.0)]
        st = simple_param(data=ContainerSet("data", cs, representation="events"), calc_mode="events",
                          params=ParamSet([Param(name=n, value=v, prior=None, range=None, is_fixed=True) for n, v in values]))
        st.setup()
        return cs

    for c in make(False):
        for key in ("energy", "coszen"):
            assert np.array_equal(c["reco_" + key], G["reco_%s_%s" % (c.name, key)]), (c.name, key)
        assert np.array_equal(c["pid"], G["reco_%s_pid" % c.name])
        assert (np.array(c["reco_energy"]) >= 0).all()          # (coszen is reflected ONCE: a large error may still leave it outside)
    assert set(np.unique(G["reco_numu_cc_pid"])) == {0.0, 1.0} and G["reco_muons_pid"].mean() > 0.45
    for c in make(True):
        assert np.array_equal(c["reco_energy"], G["reco_e"]) and np.array_equal(c["reco_coszen"], G["reco_cz"])
        assert np.array_equal(c["pid"], np.full(300, 1.0 if c.name in ("numu_cc", "numubar_cc", "muons") else 0.0))
    with pytest.raises(AssertionError):
        dict_lookup_wildcard({"nu*": 1, "numu*": 2}, "numu_cc")
