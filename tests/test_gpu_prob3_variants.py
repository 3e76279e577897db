"""`osc.prob3` constructed with every physics option of the reference's service
(pisa/stages/osc/prob3.py:167-176, 476-578: `nsi_type`, `neutrino_decay`, `lri_type`,
`include_nlo`, `reparam_mix_matrix`) through the Pipeline/cfg boundary, maps compared with the
oracle fed with the REFERENCE's own matrices for the same parameter values
(`tests/golden/params_ref.npz`).  Also: the fused path's staleness rules and fall-backs."""
import numpy as np
import pytest

from tests.conftest import PROB3_ATOL, PROB3_RTOL, load_golden

pytestmark = pytest.mark.gpu
AC = dict(rtol=PROB3_RTOL, atol=PROB3_ATOL)


def _cfg_with(tmp_path, extra_lines, name="variant.cfg", base="settings/pipeline/osc_example.cfg", replace=None):
    """the reference's cfg text + extra keys appended to its last section ([osc.prob3])"""
    from pisa_amd.utils.resources import find_resource

    text = open(find_resource(base)).read()
    for old, new in (replace or {}).items():
        assert old in text
        text = text.replace(old, new)
    path = tmp_path / name
    path.write_text(text + "\n" + "\n".join(extra_lines) + "\n")
    return str(path)


def _oracle_maps(oracle, binning, mat_pot, decay_flag=-1, mat_decay=None, lri_pot=None, mix=None):
    from pisa_amd.utils.resources import find_resource

    e = binning["true_energy"].weighted_centers.m_as("GeV")
    cz = binning["true_coszen"].weighted_centers.magnitude
    lay = oracle.Layers(np.loadtxt(find_resource("osc/PREM_12layer.dat")), 2.0, 20.0)
    lay.setElecFrac(0.4656, 0.4656, 0.4957)
    lay.calcLayers(cz)
    if mix is None:
        mix = oracle.mix_matrix(np.deg2rad(33.48), np.deg2rad(8.5), np.deg2rad(42.0), 0.0)
    dm = oracle.dm_matrix(7.5e-5, 2.457e-3)
    mat_decay = np.zeros((3, 3), complex) if mat_decay is None else mat_decay
    lri_pot = np.zeros((3, 3)) if lri_pot is None else lri_pot
    out = {}
    for nubar in (1, -1):
        P = oracle.propagate_array(dm, mix, np.asarray(mat_pot, complex), decay_flag, mat_decay, lri_pot,
                                   nubar, np.repeat(e, len(cz)), np.tile(lay.density, (len(e), 1)),
                                   np.tile(lay.distance, (len(e), 1)))
        out[nubar] = P.reshape(len(e), len(cz), 3, 3)
    return out


def _check(maps, ref):
    for m in maps:
        nubar = -1 if "bar" in m.name else 1
        flav = 0 if "nue" in m.name else (1 if "numu" in m.name else 2)
        np.testing.assert_allclose(m.hist, ref[nubar][:, :, 1, flav], err_msg=m.name, **AC)


STD = np.diag([1.0, 0.0, 0.0]).astype(complex)


def test_prob3_standard_nsi(oracle, tmp_path):
    from pisa_amd.core.pipeline import Pipeline

    g = load_golden("params_ref.npz")
    v = g["stdnsi::inputs"][0]
    names = ["eps_ee", "eps_emu_magn", "eps_emu_phase", "eps_etau_magn", "eps_etau_phase", "eps_mumu",
             "eps_mutau_magn", "eps_mutau_phase", "eps_tautau"]
    lines = ["nsi_type = standard"]
    for n, x in zip(names, v):
        lines.append("param.%s = %r%s" % (n, float(x), " * units.rad" if n.endswith("phase") else ""))
        lines.append("param.%s.fixed = True" % n)
    pipe = Pipeline(_cfg_with(tmp_path, lines))
    maps = pipe.get_outputs()
    np.testing.assert_array_equal(pipe["prob3"].gen_mat_pot_matrix_complex, STD + g["stdnsi0::eps"])
    _check(maps, _oracle_maps(oracle, pipe.output_binning, STD + g["stdnsi0::eps"]))
    # NSI moves the oscillogram: not a vacuous comparison
    ref_std = _oracle_maps(oracle, pipe.output_binning, STD)
    assert np.abs(maps["numu_cc"].hist - ref_std[1][:, :, 1, 1]).max() > 1e-2


def test_prob3_vacuum_like_nsi_with_nlo(oracle, tmp_path):
    from pisa_amd.core.pipeline import Pipeline

    g = load_golden("params_ref.npz")
    v = g["vacnsi::inputs"][1]
    names = ["eps_scale", "eps_prime", "phi12", "phi13", "phi23", "alpha1", "alpha2", "deltansi"]
    lines = ["nsi_type = vacuum-like", "include_nlo = True"]
    for n, x in zip(names, v):
        lines.append("param.%s = %r%s" % (n, float(x), "" if n.startswith("eps") else " * units.rad"))
        lines.append("param.%s.fixed = True" % n)
    pipe = Pipeline(_cfg_with(tmp_path, lines))
    maps = pipe.get_outputs()
    want = np.diag([1.02, 0.0, 0.0]).astype(complex) + g["vacnsi1::eps"]     # prob3.py:539-557
    np.testing.assert_array_equal(pipe["prob3"].gen_mat_pot_matrix_complex, want)
    _check(maps, _oracle_maps(oracle, pipe.output_binning, want))


def test_prob3_neutrino_decay(oracle, tmp_path):
    from pisa_amd.core.pipeline import Pipeline

    g = load_golden("params_ref.npz")
    lines = ["neutrino_decay = True", "param.decay_alpha3 = %r * units.eV**2" % float(g["decay::alpha3"]),
             "param.decay_alpha3.fixed = True"]
    pipe = Pipeline(_cfg_with(tmp_path, lines))
    maps = pipe.get_outputs()
    np.testing.assert_array_equal(pipe["prob3"].decay_matrix, g["decay::matrix"])
    ref = _oracle_maps(oracle, pipe.output_binning, STD, decay_flag=1, mat_decay=g["decay::matrix"])
    _check(maps, ref)
    # with decay the rows no longer sum to one
    tot = sum(maps[n].hist for n in ("nue_cc", "numu_cc", "nutau_cc"))
    assert tot.min() < 1 - 1e-6


@pytest.mark.parametrize("sym", ["emu", "etau", "mutau"])
def test_prob3_lri(oracle, tmp_path, sym):
    from pisa_amd.core.pipeline import Pipeline

    g = load_golden("params_ref.npz")
    lines = ["lri_type = %s-symmetry" % sym, "param.v_lri = %r * units.eV" % float(g["lri::v"]),
             "param.v_lri.fixed = True"]
    pipe = Pipeline(_cfg_with(tmp_path, lines))
    maps = pipe.get_outputs()
    np.testing.assert_array_equal(pipe["prob3"].lri_pot, g["lri::" + sym])
    _check(maps, _oracle_maps(oracle, pipe.output_binning, STD, lri_pot=g["lri::" + sym]))
    ref_std = _oracle_maps(oracle, pipe.output_binning, STD)
    assert np.abs(maps["numu_cc"].hist - ref_std[1][:, :, 1, 1]).max() > 1e-3


def test_prob3_reparam_mix_matrix_and_nlo(oracle, tmp_path):
    from pisa_amd.core.pipeline import Pipeline

    pipe = Pipeline(_cfg_with(tmp_path, ["reparam_mix_matrix = True", "include_nlo = True"]))
    maps = pipe.get_outputs()
    mix = oracle.mix_matrix(np.deg2rad(33.48), np.deg2rad(8.5), np.deg2rad(42.0), 0.0, reparam=True)
    _check(maps, _oracle_maps(oracle, pipe.output_binning, np.diag([1.02, 0, 0]).astype(complex), mix=mix))


def test_fused_path_sees_in_place_flux_edit(oracle):
    """the reference's idiom for an in-place edit: `container['nu_flux'][:] = ...;
    container.mark_changed('nu_flux')` (container.py:638-649).  The array OBJECT stays the same; the
    fused engine's folded (w0*aeff*flux) column must follow."""
    from pisa_amd.core.pipeline import Pipeline

    pipe = Pipeline("settings/pipeline/example_hip.cfg")
    maps = pipe.get_outputs()
    assert pipe["hist"].fused_last_eval
    before = {m.name: m.hist.copy() for m in maps}
    for c in pipe.data.containers:
        c.representation = "events"
        flux = c["nu_flux"]
        obj = c.current_data["nu_flux"]
        flux[:, 1] *= 2.0
        flux[:, 0] *= 3.0
        c.mark_changed("nu_flux")
        assert c.current_data["nu_flux"] is obj
    maps2 = pipe.get_outputs()
    assert pipe["hist"].fused_last_eval
    # the same edit through the unfused path
    pipe2 = Pipeline("settings/pipeline/example_hip.cfg")
    pipe2["hist"]._fused = lambda: False
    pipe2.get_outputs()
    for c in pipe2.data.containers:
        c.representation = "events"
        flux = c["nu_flux"]
        flux[:, 1] *= 2.0
        flux[:, 0] *= 3.0
        c.mark_changed("nu_flux")
    maps3 = pipe2.get_outputs()
    for a, b in zip(maps2, maps3):
        np.testing.assert_allclose(a.hist, b.hist, rtol=1e-13, atol=1e-300, err_msg=a.name)
        assert np.abs(a.hist - before[a.name]).max() > 1e-3 * np.abs(before[a.name]).max()
    # and a rewritten static column (weighted_aeff) rebuilds the engine
    for p in (pipe, pipe2):
        for c in p.data.containers:
            c.representation = "events"
            c["weighted_aeff"] = c["weighted_aeff"] * 0.5
    m4, m5 = pipe.get_outputs(), pipe2.get_outputs()
    assert pipe["hist"].fused_last_eval
    for a, b, c0 in zip(m4, m5, maps2):
        np.testing.assert_allclose(a.hist, b.hist, rtol=1e-13, atol=1e-300)
        np.testing.assert_allclose(a.hist, 0.5 * c0.hist, rtol=1e-13, atol=1e-300)


def test_event_weights_stay_readable_after_fused_eval():
    """histogramming does not invalidate the event-wise weights (utils/hist.py:213): after a fused
    evaluation a read of `weights` in the events representation gives the per-event products, not
    a bin lookup of the map."""
    from pisa_amd.core.pipeline import Pipeline

    pipe = Pipeline("settings/pipeline/example_hip.cfg")
    maps = pipe.get_outputs()
    assert pipe["hist"].fused_last_eval
    pipe2 = Pipeline("settings/pipeline/example_hip.cfg")
    pipe2["hist"]._fused = lambda: False
    pipe2.get_outputs()
    for name in ("numu_cc", "nuebar_nc"):
        c, c2 = pipe.data[name], pipe2.data[name]
        c.representation = c2.representation = "events"
        w, w2 = c["weights"], c2["weights"]
        assert w.shape == c["true_energy"].shape
        np.testing.assert_allclose(w, w2, rtol=1e-15, atol=0)
        # the map is still valid and unchanged after the event-wise read
        c.representation = pipe.output_binning
        np.testing.assert_array_equal(c["weights"].reshape(maps[name].hist.shape), maps[name].hist)
    # the next evaluation is fused again
    pipe.get_outputs()
    assert pipe["hist"].fused_last_eval


def test_linear_energy_calc_grid_takes_unfused_path(oracle, tmp_path):
    """a calc grid that is not (log E, lin cz) must not go through the engine's GridSpec
    (ADVICE r1): the fused path declines and the lookups go through `regularized()`."""
    from pisa_amd.core.pipeline import Pipeline
    from pisa_amd.utils.resources import find_resource

    text = open(find_resource("settings/pipeline/example_hip.cfg")).read()
    text = text.replace("#include settings/binning/example.cfg as binning\n", "")
    btext = open(find_resource("settings/binning/example.cfg")).read()
    import re

    btext_lin = re.sub(r"calc_grid\.true_energy = \{[^\n]*\}",
                       "calc_grid.true_energy = {'num_bins':40, 'is_lin':True, 'domain':[1.,81] * units.GeV, 'tex': r'E'}",
                       btext)
    assert btext_lin != btext
    path = tmp_path / "lin.cfg"
    path.write_text(text + "\n[binning]\n" + btext_lin)
    pipe = Pipeline(str(path))
    maps = pipe.get_outputs()
    assert not pipe["hist"].fused_last_eval
    # oracle chain with a LINEAR energy lookup
    cm = pipe["prob3"].calc_mode
    e = cm["true_energy"].weighted_centers.m_as("GeV")
    cz = cm["true_coszen"].weighted_centers.magnitude
    np.testing.assert_allclose(e[:2], [2.0, 4.0])
    lay = oracle.Layers(np.loadtxt(find_resource("osc/PREM_12layer.dat")), 2.0, 20.0)
    lay.setElecFrac(0.4656, 0.4656, 0.4957)
    lay.calcLayers(cz)
    mix = oracle.mix_matrix(np.deg2rad(33.48), np.deg2rad(8.5), np.deg2rad(42.3), 0.0)
    dm = oracle.dm_matrix(7.5e-5, 2.457e-3)
    zero = np.zeros((3, 3))
    ob = pipe.output_binning
    omin, omax, onb = [np.log(5.0), -1.0, -1000.0], [np.log(100.0), 1.0, 1000.0], list(ob.shape)
    for c in pipe.data.containers:
        c.representation = "events"
        nubar, flav = c["nubar"], c["flav"]
        P = oracle.propagate_array(dm, mix, STD, -1, zero.astype(complex), zero, nubar, np.repeat(e, len(cz)),
                                   np.tile(lay.density, (len(e), 1)), np.tile(lay.distance, (len(e), 1)))
        mins, maxs, nb = [1.0, -1.0], [81.0, 1.0], [len(e), len(cz)]
        ev, czv = c["true_energy"], c["true_coszen"]
        pe = oracle.lookup_regular([ev, czv], np.ascontiguousarray(P[:, 0, flav]), mins, maxs, nb)
        pmu = oracle.lookup_regular([ev, czv], np.ascontiguousarray(P[:, 1, flav]), mins, maxs, nb)
        flux = oracle.barr_simple(ev, czv, c["nu_flux_nominal"], c["nubar_flux_nominal"], nubar, 1.0, 1.0, 0.0, 0.0, 0.0)
        w = oracle.reweight(c["initial_weights"], flux, pe, pmu, c["weighted_aeff"], 2.5 * 365 * 86400.0)
        sample = [np.log(c["reco_energy"]), c["reco_coszen"], c["pid"]]
        want = oracle.histogram_regular(sample, w, omin, omax, onb).reshape(ob.shape)
        np.testing.assert_allclose(maps[c.name].hist, want, rtol=1e-11, atol=1e-300, err_msg=c.name)


def test_prob3_tomography_follows_the_reference_call_for_call(oracle, tmp_path):
    """prob3.py:278-300, 378-395, 519-536 + layers.py:291-306, 411-439.  In this version of the
    reference `Layers.scaling` is followed by `setElecFrac`, which re-derives the shell densities
    from the UNSCALED table: the tomography parameters are validated (Earth-model check, sign
    assertions) but do not move the oscillograms.  The product does the same."""
    from pisa_amd.core.pipeline import Pipeline
    from pisa_amd.core.units import ureg

    prem5 = tmp_path / "osc"
    prem5.mkdir()
    (prem5 / "PREM_5layer_tomo.dat").write_text(
        "0.0 13.0\n1221.5 13.0\n3480.0 10.96\n5701.0 5.03\n6151.0 3.7\n6371.0 2.5\n")
    import os

    os.environ["PISA_RESOURCES"] = str(tmp_path)
    try:
        five = {"param.earth_model = osc/PREM_12layer.dat": "param.earth_model = osc/PREM_5layer_tomo.dat"}
        ref_maps = Pipeline(_cfg_with(tmp_path, [], name="plain5.cfg", replace=five)).get_outputs()
        pipe = Pipeline(_cfg_with(tmp_path, [
            "tomography_type = mass_of_core_w_constrain", "param.core_density_scale = 1.1",
            "param.core_density_scale.fixed = False", "param.core_density_scale.range = [0.5, 1.5]"],
            name="tomo.cfg", replace=five))
        maps = pipe.get_outputs()
        osc = pipe["prob3"]
        # scaling() saw the factors ...
        want = np.concatenate(([1.0], np.array([2.5, 3.7, 5.03, 10.96, 13.0, 13.0]) * osc.tomography_params.scaling_array))
        np.testing.assert_allclose(osc.scaled_rhos, want, rtol=1e-14)
        assert osc.tomography_params.scaling_array[3] == 1.1
        # ... and setElecFrac restored the Ye-weighted unscaled table, as in the reference
        for a, b in zip(maps, ref_maps):
            np.testing.assert_array_equal(a.hist, b.hist)
        pipe.params.core_density_scale.value = 0.9 * ureg.dimensionless
        n = len(osc.calc_times) if osc.profile else None
        maps2 = pipe.get_outputs()
        for a, b in zip(maps2, ref_maps):
            np.testing.assert_array_equal(a.hist, b.hist)
        del n
        # one global factor works with any Earth model; the core types insist on the five-shell Earth
        p12 = Pipeline(_cfg_with(tmp_path, ["tomography_type = mass_of_earth", "param.density_scale = 1.2",
                                            "param.density_scale.fixed = True"], name="mass.cfg"))
        _check(p12.get_outputs(), _oracle_maps(oracle, p12.output_binning, STD))
        with pytest.raises(ValueError):
            Pipeline(_cfg_with(tmp_path, ["tomography_type = mass_of_core_wo_constrain",
                                          "param.core_density_scale = 1.0", "param.innermantle_density_scale = 1.0",
                                          "param.middlemantle_density_scale = 1.0"], name="bad.cfg"))
        with pytest.raises(ValueError):
            Pipeline(_cfg_with(tmp_path, ["tomography_type = nonsense"], name="bad2.cfg"))
    finally:
        del os.environ["PISA_RESOURCES"]


def test_reduced_form_against_reference_order_over_random_parameters():
    """The planned grid kernels and the event-mode kernel form the layer matrices from the
    Hermitian mass-basis matrix (eigenvalues from ITS invariants, projectors as two real and three
    complex entries); the one-kernel grid form follows the reference operation for operation.
    120 random parameter points -- mixing angles anywhere, both orderings, small and near-degenerate
    mass splittings, random Hermitian NSI potentials, LRI potentials, 0.1 GeV to 10 TeV -- must
    agree to rounding on every probability."""
    from pisa_amd import _lib as L
    from pisa_amd import kernels as K
    from pisa_amd.stages.osc.layers import Layers
    from pisa_amd.stages.osc.osc_params import OscParams

    lay = Layers("osc/PREM_12layer.dat", 2.0, 20.0)
    lay.setElecFrac(0.4656, 0.4656, 0.4957)
    cz = np.linspace(-1.0, 1.0, 24)
    lay.calcLayers(cz)
    _, dens, dist = lay.device_arrays
    plan = K.GridPlan(dens, dist)
    energy = np.logspace(-1, 4, 70)
    e_d = K.to_device(energy)
    earth = lay.earth_struct()
    ee, cc = np.meshgrid(energy, cz, indexing="ij")
    ev_e, ev_cz = K.to_device(ee.ravel()), K.to_device(cc.ravel())
    rs = np.random.RandomState(77)
    worst = 0.0
    # the reference's own prob3 tolerance (numba_osc_tests.py:82, rtol 1e-10 on probabilities <= 1);
    # typical differences are 1e-13, the largest (2e-11) at near-degenerate splittings and TeV
    # energies, where the eigenvalue differences in the denominators are ill conditioned in either form
    GATE = 1e-10
    for k in range(120):
        o = OscParams()
        o.theta12, o.theta13, o.theta23 = rs.rand(3) * np.pi / 2
        o.deltacp = rs.rand() * 2 * np.pi
        o.dm21 = 10 ** rs.uniform(-6, -3.5)
        o.dm31 = (1 if rs.rand() < 0.5 else -1) * 10 ** rs.uniform(-3.3, -2.2)
        mat_pot = np.diag([1.0, 0.0, 0.0]).astype(np.complex128)
        if k % 3 == 1:                        # Hermitian NSI
            a = (rs.randn(3, 3) + 1j * rs.randn(3, 3)) * 0.2
            mat_pot = mat_pot + (a + a.conj().T) / 2
        lri = np.zeros((3, 3))
        if k % 3 == 2:                        # long-range potential (real symmetric, eV)
            b = rs.randn(3, 3) * 1e-13
            lri = (b + b.T) / 2
        p = L.make_prob3_params(o.dm_matrix, o.mix_matrix_complex, mat_pot, -1, np.zeros((3, 3), np.complex128), lri)
        nu, nubar = K.prob3_grid(p, e_d, dens, dist, e_major=True)[:2]
        nu2, nubar2, _ = K.prob3_grid_planned(p, plan, e_d, e_major=True)
        for a_, b_ in ((nu, nu2), (nubar, nubar2)):
            d = float((a_ - b_).abs().max())
            worst = max(worst, d)
            assert d < GATE, (k, d)
        # event mode at the same nodes (its own path geometry, same reduced form)
        for nubar_sign, ref in ((1, nu), (-1, nubar)):
            got = K.prob3_events(p, earth, nubar_sign, ev_e, ev_cz)
            d = float((got - ref).abs().max())
            worst = max(worst, d)
            assert d < GATE, (k, nubar_sign, d)
        rows = (nu2.sum(dim=2) - 1.0).abs().max()          # unitarity of every row
        assert float(rows) < 1e-11
    assert worst < GATE
