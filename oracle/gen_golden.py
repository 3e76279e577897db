#!/usr/bin/env python
"""Generate the committed fixtures under tests/golden/ by EXECUTING the
reference (read in place from /root/reference via oracle/ref_shim.py).

Run in the build container only (the reference does not travel):

    python oracle/gen_golden.py

Outputs (all numpy .npz, inputs + expected outputs only -- no reference text):
  prob3_ref_goldens.npz   the reference's own golden pickles for the 13 named
                          cases x 10 host functions (f8), re-encoded
                          (pisa_examples/resources/osc/numba_osc_tests_data)
  prob3_grid_prem12.npz   reference osc_probs_layers_kernel on a coarse (E,cz)
                          PREM-12 grid, nu/nubar x {NO, IO, std-NSI, decay}
  layers_ref.npz          reference Layers.calcLayers for PREM-4/12/59
  params_ref.npz          OscParams / NSI / decay / LRI matrices
  lookup_ref.npz          reference lookup_regular_* outputs
  stats_ref.npz           reference stats.llh/poisson_llh/chi2/mod_chi2
  barr_ref.npz            reference barr_simple.apply_sys_kernel
  hist_ref.npz            np.histogramdd recipe of translation.test_histogram
  stats_wide_ref.npz      reference stats.mcllh_mean / mcllh_eff / correct_chi2 / signed_sqrt_mod_chi2 / conv_llh
  side_stages_ref.npz     the services around the path: reference two_nu_osc.apply_probs_vectorized,
                          astrophysical.apply_sys_loop, genie_sys.apply_genie_sys,
                          bin_indexing.lookup_indices_vectorized_{1,2,3}d (+ the arrays of its unit test)
"""
import glob
import os
import pickle
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
warnings.filterwarnings("ignore")


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def gen_ref_pickles():
    d = os.path.join(ref_shim.REF_RESOURCES, "osc", "numba_osc_tests_data")
    out = {}
    for f in sorted(glob.glob(os.path.join(d, "*__f8.pkl"))):
        base = os.path.basename(f)[: -len("__f8.pkl")]
        with open(f, "rb") as fh:
            t = pickle.load(fh)
        for k, v in t.items():
            out["%s::%s" % (base, k)] = np.asarray(v)
    save("prob3_ref_goldens.npz", **out)


OSC_SCENARIOS = {
    # osc_example.cfg nominal (nufit v2.0 NH)
    "no": dict(theta12=33.48, theta13=8.5, theta23=42.0, deltacp=0.0, dm21=7.5e-5, dm31=2.457e-3),
    "io": dict(theta12=33.48, theta13=8.51, theta23=49.5, deltacp=254.0, dm21=7.5e-5, dm31=-2.374e-3),
    "nsi": dict(theta12=33.48, theta13=8.5, theta23=42.3, deltacp=306.0, dm21=7.5e-5, dm31=2.457e-3),
    "decay": dict(theta12=33.48, theta13=8.5, theta23=42.0, deltacp=90.0, dm21=7.5e-5, dm31=2.457e-3),
}


def scenario_matrices(name):
    op = ref_shim.ref_module("pisa.stages.osc.osc_params")
    nsi = ref_shim.ref_module("pisa.stages.osc.nsi_params")
    dec = ref_shim.ref_module("pisa.stages.osc.decay_params")
    s = OSC_SCENARIOS[name]
    o = op.OscParams()
    o.theta12 = np.deg2rad(s["theta12"])
    o.theta13 = np.deg2rad(s["theta13"])
    o.theta23 = np.deg2rad(s["theta23"])
    o.deltacp = np.deg2rad(s["deltacp"])
    o.dm21 = s["dm21"]
    o.dm31 = s["dm31"]
    mat_pot = np.diag([1.0, 0, 0]).astype(np.complex128)
    decay_flag = -1
    mat_decay = np.zeros((3, 3), np.complex128)
    if name == "nsi":
        n = nsi.StdNSIParams()
        n.eps_emu = (0.07, np.deg2rad(340))
        n.eps_etau = (0.06, np.deg2rad(35))
        n.eps_mutau = (0.003, np.deg2rad(175))
        n.eps_ee = 0.1
        n.eps_tautau = -0.05
        mat_pot = mat_pot + n.eps_matrix
    if name == "decay":
        dp = dec.DecayParams()
        dp.decay_alpha3 = 1.0e-4
        mat_decay = dp.decay_matrix
        decay_flag = 1
    return dict(
        dm=o.dm_matrix, mix=o.mix_matrix_complex, mat_pot=mat_pot, decay_flag=decay_flag,
        mat_decay=mat_decay, lri_pot=np.zeros((3, 3)),
    )


def gen_grid():
    k = ref_shim.kernels()
    L = ref_shim.layers_mod()
    lay = L.Layers("osc/PREM_12layer.dat", 2.0, 20.0)
    lay.setElecFrac(0.4656, 0.4656, 0.4957)
    n_e, n_cz = 20, 24
    e_edges = np.logspace(0, 3, n_e + 1)
    energy = np.sqrt(e_edges[:-1] * e_edges[1:])
    cz_edges = np.linspace(-1, 1, n_cz + 1)
    cz = 0.5 * (cz_edges[:-1] + cz_edges[1:])
    lay.calcLayers(cz)
    dens = lay.density.reshape(n_cz, lay.max_layers)
    dist = lay.distance.reshape(n_cz, lay.max_layers)
    out = dict(energy=energy, coszen=cz, densities=dens, distances=dist)
    for name in OSC_SCENARIOS:
        m = scenario_matrices(name)
        for key, val in m.items():
            out["%s::%s" % (name, key)] = np.asarray(val)
        for nubar in (1, -1):
            P = np.zeros((n_e, n_cz, 3, 3))
            for i in range(n_e):
                for j in range(n_cz):
                    k.osc_probs_layers_kernel(
                        m["dm"], m["mix"], m["mat_pot"], m["decay_flag"], m["mat_decay"],
                        m["lri_pot"], nubar, float(energy[i]), dens[j], dist[j], P[i, j],
                    )
            out["%s::prob_%s" % (name, "nu" if nubar > 0 else "nubar")] = P
        print("grid scenario", name, "done")
    save("prob3_grid_prem12.npz", **out)


def gen_layers():
    L = ref_shim.layers_mod()
    rs = np.random.RandomState(42)
    out = {}
    for tag, fn, depth, height, ye in [
        ("prem4", "osc/PREM_4layer.dat", 1.0, 20.0, (0.5, 0.5, 0.5)),
        ("prem4b", "osc/PREM_4layer.dat", 10.0, 18.0, (0.5, 0.5, 0.5)),
        ("prem12", "osc/PREM_12layer.dat", 2.0, 20.0, (0.4656, 0.4656, 0.4957)),
        ("prem59", "osc/PREM_59layer.dat", 2.0, 20.0, (0.4656, 0.4656, 0.4957)),
        ("prem10", "osc/PREM_10layer.dat", 2.0, 20.0, (0.466, 0.467, 0.494)),
    ]:
        lay = L.Layers(fn, depth, height)
        lay.setElecFrac(*ye)
        cz = np.concatenate(
            [
                np.array([1.0, 0.0, -0.4461133826191877, -1.0, 0.5, -0.2, -0.9, 1e-9, -1e-9]),
                # just either side of every tangency (the reference itself raises a shape
                # error AT an exact tangency value for some shells, layers.py:158)
                lay.coszen_limit[lay.coszen_limit < 1.0] + 1e-9,
                lay.coszen_limit[lay.coszen_limit < 1.0][:-1] - 1e-9,
                rs.rand(64) * 2 - 1,
                0.5 * (np.linspace(-1, 1, 201)[:-1] + np.linspace(-1, 1, 201)[1:]),
            ]
        )
        lay.calcLayers(cz)
        out[tag + "::prem"] = np.loadtxt(os.path.join(ref_shim.REF_RESOURCES, fn))
        out[tag + "::args"] = np.array([depth, height, *ye])
        out[tag + "::cz"] = cz
        out[tag + "::radii"] = lay.radii
        out[tag + "::rhos"] = lay.rhos
        out[tag + "::coszen_limit"] = lay.coszen_limit
        out[tag + "::n_layers"] = lay.n_layers
        out[tag + "::density"] = lay.density.reshape(len(cz), lay.max_layers)
        out[tag + "::distance"] = lay.distance.reshape(len(cz), lay.max_layers)
    save("layers_ref.npz", **out)


def gen_params():
    op = ref_shim.ref_module("pisa.stages.osc.osc_params")
    nsi = ref_shim.ref_module("pisa.stages.osc.nsi_params")
    dec = ref_shim.ref_module("pisa.stages.osc.decay_params")
    lri = ref_shim.ref_module("pisa.stages.osc.lri_params")
    rs = np.random.RandomState(7)
    out = {}
    angles = []
    for i in range(8):
        a = dict(
            theta12=rs.rand() * np.pi / 2, theta13=rs.rand() * np.pi / 2,
            theta23=rs.rand() * np.pi / 2, deltacp=rs.rand() * 2 * np.pi,
            dm21=rs.rand() * 1e-4, dm31=(rs.rand() - 0.5) * 1e-2,
        )
        if i == 0:
            a.update(dm21=0.0, dm31=0.0, deltacp=0.0)
        o = op.OscParams()
        for kk, vv in a.items():
            setattr(o, kk, vv)
        angles.append([a[k] for k in ("theta12", "theta13", "theta23", "deltacp", "dm21", "dm31")])
        out["osc%d::mix" % i] = o.mix_matrix_complex
        out["osc%d::mix_reparam" % i] = o.mix_matrix_reparam_complex
        out["osc%d::dm" % i] = o.dm_matrix
    out["osc::inputs"] = np.array(angles)
    # standard NSI
    vals = []
    for i in range(4):
        v = rs.rand(9)
        n = nsi.StdNSIParams()
        n.eps_ee = v[0] - 0.5
        n.eps_emu = (v[1], v[2] * 2 * np.pi)
        n.eps_etau = (v[3], v[4] * 2 * np.pi)
        n.eps_mumu = v[5] - 0.5
        n.eps_mutau = (v[6], v[7] * 2 * np.pi)
        n.eps_tautau = v[8] - 0.5
        vals.append([v[0] - 0.5, v[1], v[2] * 2 * np.pi, v[3], v[4] * 2 * np.pi, v[5] - 0.5, v[6],
                     v[7] * 2 * np.pi, v[8] - 0.5])
        out["stdnsi%d::eps" % i] = n.eps_matrix
    out["stdnsi::inputs"] = np.array(vals)
    vals = []
    for i in range(4):
        v = rs.rand(8)
        n = nsi.VacuumLikeNSIParams()
        n.eps_scale = v[0] * 2
        n.eps_prime = v[1] - 0.5
        n.phi12 = (v[2] - 0.5) * np.pi
        n.phi13 = (v[3] - 0.5) * np.pi
        n.phi23 = (v[4] - 0.5) * np.pi
        n.alpha1 = v[5] * 2 * np.pi
        n.alpha2 = v[6] * 2 * np.pi
        n.deltansi = v[7] * 2 * np.pi
        vals.append([n.eps_scale, n.eps_prime, n.phi12, n.phi13, n.phi23, n.alpha1, n.alpha2, n.deltansi])
        out["vacnsi%d::eps" % i] = n.eps_matrix
    out["vacnsi::inputs"] = np.array(vals)
    d = dec.DecayParams()
    d.decay_alpha3 = 3.3e-4
    out["decay::matrix"] = d.decay_matrix
    out["decay::alpha3"] = np.array(3.3e-4)
    l = lri.LRIParams()
    l.v_lri = 2.5e-14
    out["lri::v"] = np.array(2.5e-14)
    out["lri::emu"] = l.potential_matrix_emu
    out["lri::etau"] = l.potential_matrix_etau
    out["lri::mutau"] = l.potential_matrix_mutau
    save("params_ref.npz", **out)


def gen_lookup():
    tr = ref_shim.ref_module("pisa.core.translation")
    rs = np.random.RandomState(3)
    n = 2000
    out = {}
    x = rs.rand(n) * 1.4 - 0.2
    y = rs.rand(n) * 2.6 - 1.3
    z = rs.rand(n) * 3 - 0.5
    # exact edge values / NaN / inf
    x[:6] = [0.0, 1.0, np.nextafter(1.0, 0), np.nan, np.inf, -np.inf]
    y[6:10] = [-1.0, 1.0, np.nextafter(1.0, 0), np.nextafter(-1.0, -2)]
    out["x"], out["y"], out["z"] = x, y, z
    h1 = rs.rand(7)
    o = np.zeros(n)
    tr.lookup_regular_1d(x, h1, 0.0, 1.0, 7, o)
    out["h1"], out["o1"] = h1, o.copy()
    h2 = rs.rand(7 * 5)
    o = np.zeros(n)
    tr.lookup_regular_2d(x, y, h2, 0.0, 1.0, 7, -1.0, 1.0, 5, o)
    out["h2"], out["o2"] = h2, o.copy()
    h3 = rs.rand(7 * 5 * 3)
    o = np.zeros(n)
    tr.lookup_regular_3d(x, y, z, h3, 0.0, 1.0, 7, -1.0, 1.0, 5, 0.0, 2.0, 3, o)
    out["h3"], out["o3"] = h3, o.copy()
    h2a = rs.rand(7 * 5, 2)
    o = np.zeros((n, 2))
    tr.lookup_regular_2d_array(x, y, h2a, 0.0, 1.0, 7, -1.0, 1.0, 5, o)
    out["h2a"], out["o2a"] = h2a, o.copy()
    save("lookup_ref.npz", **out)


def gen_stats():
    st = ref_shim.ref_module("pisa.utils.stats")
    rs = np.random.RandomState(5)
    expected = rs.rand(128) * 50
    expected[:4] = [0.0, 1e-12, 1e-10, 5.0]
    actual = rs.poisson(np.maximum(expected, 0.5)).astype(np.float64)
    actual[4:8] = 0.0
    out = dict(actual=actual, expected=expected)
    for name in ("llh", "poisson_llh", "chi2", "mod_chi2"):
        v = getattr(st, name)(actual.copy(), expected.copy())
        v = np.ma.filled(np.ma.masked_invalid(np.ma.asarray(v, dtype=float)), np.nan)
        out[name] = np.asarray(v, dtype=np.float64)
        out[name + "_total"] = np.array(np.nansum(out[name]))
    save("stats_ref.npz", **out)


def gen_stats_wide():
    """the metrics beyond llh / poisson_llh / chi2 / mod_chi2, by the reference's own functions.  `uncertainties` is
    absent: expected values travel as an ndarray subclass carrying its standard deviations, which the stand-in for
    `unumpy` splits off again (nominal_values / std_devs) -- no arithmetic of that package is emulated."""
    import importlib.util

    st = ref_shim.ref_module("pisa.utils.stats")
    spec = importlib.util.spec_from_file_location(
        "pisa.utils.likelihood_functions", os.path.join(ref_shim.REF_ROOT, "pisa", "utils", "likelihood_functions.py"))
    lf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lf)
    st.likelihood_functions = lf

    class WithStd(np.ndarray):
        pass

    def with_std(values, sigma):
        a = np.array(values, dtype=np.float64).view(WithStd)
        a.s = np.array(sigma, dtype=np.float64)
        return a

    st.unp = types.SimpleNamespace(nominal_values=lambda x: np.array(x, dtype=np.float64),
                                   std_devs=lambda x: np.array(getattr(x, "s", np.zeros(np.shape(x)))))
    rs = np.random.RandomState(7)
    n = 160
    expected = rs.rand(n) * 60
    expected[:6] = [0.0, 1e-12, 1e-10, 5.0, 0.3, 1500.0]
    sigma = np.sqrt(expected) * rs.rand(n) * 0.7
    sigma[6:12] = 0.0                                   # the Poisson limit of the mixture
    sigma[12:16] = [1e-6, 1e-3, 30.0, 100.0]
    actual = rs.poisson(np.maximum(expected, 0.5)).astype(np.float64)
    actual[16:20] = 0.0
    actual[5] = 1450.0
    out = dict(actual=actual, expected=expected, sigma=sigma)
    for name in ("mcllh_mean", "mcllh_eff", "correct_chi2", "signed_sqrt_mod_chi2", "conv_llh", "mod_chi2"):
        v = getattr(st, name)(actual.copy(), with_std(expected, sigma))
        v = np.ma.filled(np.ma.masked_invalid(np.ma.asarray(v, dtype=float)), np.nan)
        out[name] = np.asarray(v, dtype=np.float64)
    save("stats_wide_ref.npz", **out)


def gen_barr():
    bs = ref_shim.ref_module("pisa.stages.flux.barr_simple")
    rs = np.random.RandomState(11)
    n = 300
    e = 10 ** (rs.rand(n) * 3)
    cz = rs.rand(n) * 2 - 1
    nu = rs.rand(n, 2) * 10
    nub = rs.rand(n, 2) * 10
    nu[:3] = 0.0
    nub[:3] = 0.0
    nu[3] = [0.0, 1.0]
    nub[3] = [0.0, 1.0]
    out = dict(true_energy=e, true_coszen=cz, nu_flux_nominal=nu, nubar_flux_nominal=nub)
    psets = [
        (1.0, 1.0, 0.0, 0.0, 0.0),
        (1.03, 0.9, 0.05, 0.7, -0.4),
        (0.8, 1.2, -0.1, -1.5, 2.0),
    ]
    out["params"] = np.array(psets)
    for ip, ps in enumerate(psets):
        for nubar in (1, -1):
            o = np.zeros((n, 2))
            for i in range(n):
                bs.apply_sys_kernel(e[i], cz[i], nu[i], nub[i], nubar, *ps, o[i])
            out["out%d_%s" % (ip, "nu" if nubar > 0 else "nubar")] = o
    save("barr_ref.npz", **out)


def gen_hist():
    """translation.test_histogram recipe (translation.py:779-818): the
    reference pins fast_histogram against np.histogramdd on these samples."""
    all_num_bins = [2, 3, 4]
    n_evts = 10000
    rand = np.random.RandomState(seed=0)
    weights = rand.rand(n_evts)
    out = dict(weights=weights)
    sample = []
    for nd, nb in enumerate(all_num_bins, start=1):
        s = rand.rand(n_evts) * nb
        sample.append(s)
        out["s%d" % (nd - 1)] = s
        edges = [np.linspace(0, b, b + 1) for b in all_num_bins[:nd]]
        ref, _ = np.histogramdd(sample=sample, bins=edges, weights=weights)
        cnt, _ = np.histogramdd(sample=sample, bins=edges, weights=None)
        out["ref%dd" % nd] = ref.ravel()
        out["cnt%dd" % nd] = cnt.ravel()
    save("hist_ref.npz", **out)


def gen_flux():
    """Honda 2-D (azimuth-averaged) table -> nominal fluxes at sample points, by the
    reference's own integral-preserving spline code (pisa/utils/flux_weights.py:50-131,
    267-349), incl. points outside the table's energy range and at cz = +-1."""
    fw = ref_shim.ref_module("pisa.utils.flux_weights")
    table = "flux/honda-2015-spl-solmin-aa.d"
    splines = fw.load_2d_table(table)
    rs = np.random.RandomState(21)
    n = 400
    e = 10 ** (rs.rand(n) * 5 - 1)          # 0.1 GeV .. 10 TeV
    cz = rs.rand(n) * 2 - 1
    e[:6] = [0.05, 0.09, 0.1, 1.0e4, 1.2e4, 3.0e4]   # below / at / above the table range
    cz[6:12] = [-1.0, 1.0, -0.95, 0.95, 0.0, -0.9]
    out = dict(table=np.array(table), true_energy=e, true_coszen=cz)
    for prim in ("nue", "numu", "nuebar", "numubar"):
        out[prim] = fw.calculate_2d_flux_weights(e, cz, splines[prim])
    save("flux_ref.npz", **out)
    # the Bartol table (Honda-like layout, coarser energy steps above 10 GeV; flux_weights.py:133-203)
    table = "flux/bartol-2004-sno-solmax-aa.d"
    splines = fw.load_2d_table(table)
    e = 10 ** (rs.rand(n) * 5 - 1)
    cz = rs.rand(n) * 2 - 1
    e[:6] = [0.05, 0.09, 0.1, 1.0e4, 1.2e4, 3.0e4]
    cz[6:12] = [-1.0, 1.0, -0.95, 0.95, 0.0, -0.9]
    out = dict(table=np.array(table), true_energy=e, true_coszen=cz)
    for prim in ("nue", "numu", "nuebar", "numubar"):
        out[prim] = fw.calculate_2d_flux_weights(e, cz, splines[prim])
    save("flux_bartol_ref.npz", **out)


def gen_side():
    """the small services around the hot path, each by the reference's own function on seeded inputs"""
    p = os.path.join(ref_shim.REF_ROOT, "pisa", "stages")
    ref_shim.install()
    ref_shim._pkg("pisa.stages.xsec", os.path.join(p, "xsec"))
    rs = np.random.RandomState(31)
    n = 400
    out = {}
    # --- osc.two_nu_osc
    two = ref_shim.ref_module("pisa.stages.osc.two_nu_osc")
    e = 10 ** (rs.rand(n) * 3 - 0.5)
    cz = rs.rand(n) * 2 - 1
    cz[:4] = [-1.0, 1.0, 0.0, -0.5]
    flux = rs.rand(n, 2) * 5
    w0 = rs.rand(n) + 0.5
    out.update(two_e=e, two_cz=cz, two_flux=flux, two_w0=w0)
    cases = [(np.deg2rad(45.0), 2.5e-3), (0.6, -2.4e-3), (1.0, 0.0)]
    out["two_params"] = np.array(cases)
    for ic, (t23, dm31) in enumerate(cases):
        for code, tag in ((0, "nue"), (1, "numu"), (3, "nutau")):
            w = w0.copy()
            for i in range(n):
                two.apply_probs_vectorized(flux[i], t23, dm31, e[i], cz[i], code, w[i:i + 1])
            out["two_%d_%s" % (ic, tag)] = w
    # --- flux.astrophysical
    astro = ref_shim.ref_module("pisa.stages.flux.astrophysical")
    ea = 10 ** (rs.rand(n) * 5 + 2)
    nominal = 0.787e-18 * np.power(ea / astro.PIVOT, -2.5)            # astrophysical.py:69-72
    out.update(astro_e=ea, astro_nominal=nominal)
    acases = [(0.0, 1.0), (0.3, 1.7), (-0.45, 0.2)]
    out["astro_params"] = np.array(acases)
    for ic, (delta, norm) in enumerate(acases):
        o = np.zeros(n)
        astro.apply_sys_loop(ea, np.zeros(n), delta, norm, nominal, o)
        out["astro_%d" % ic] = o
    # --- xsec.genie_sys
    genie = ref_shim.ref_module("pisa.stages.xsec.genie_sys")
    lin = [rs.randn(n) * 0.3 for _ in range(3)]
    quad = [rs.randn(n) * 0.2 for _ in range(3)]
    out.update(genie_lin=np.array(lin), genie_quad=np.array(quad), genie_w0=w0)
    gcases = [(0.0, 0.0, 0.0), (1.0, -0.5, 0.3), (-2.0, 2.0, 1.5), (-4.0, 3.0, -3.0)]
    out["genie_params"] = np.array(gcases)
    for ic, ps in enumerate(gcases):
        for k in (1, 2, 3):
            w = w0.copy()
            genie.apply_genie_sys(list(ps[:k]), lin[:k], quad[:k], out=w)
            out["genie_%d_%d" % (ic, k)] = w
    # --- core.bin_indexing
    bi = ref_shim.ref_module("pisa.core.bin_indexing")
    edges = [np.array([0.0, 1.0, 2.5, 2.75, 7.0]), np.logspace(0, 2, 7), np.linspace(-1, 1, 4)]
    cols = [rs.uniform(-1, 8, n), 10 ** rs.uniform(-0.3, 2.3, n), rs.uniform(-1.3, 1.3, n)]
    for c, ed in zip(cols, edges):
        c[:len(ed)] = ed
    cols[0][10], cols[1][11], cols[2][12] = np.nan, np.nan, np.nan
    out.update(idx_edges0=edges[0], idx_edges1=edges[1], idx_edges2=edges[2], idx_cols=np.array(cols))
    funcs = {1: bi.lookup_indices_vectorized_1d, 2: bi.lookup_indices_vectorized_2d, 3: bi.lookup_indices_vectorized_3d}
    for nd in (1, 2, 3):
        res = np.zeros(n, dtype=np.int64)
        for i in range(n):
            funcs[nd](*([c[i:i + 1] for c in cols[:nd]] + edges[:nd] + [res[i:i + 1]]))
        out["idx_%dd" % nd] = res
    # --- osc.decoherence, 3-flavour form.  pint is absent: the quantities are handed over as an INERT stand-in whose
    # m_as() returns the magnitude as it is, every value already in the unit the reference asks for (rad, GeV, eV**2,
    # km) -- no conversion is emulated; the 2-flavour form (which converts km -> m and GeV -> eV) is not pinned here
    dec = ref_shim.ref_module("pisa.stages.osc.decoherence")

    class Q:
        def __init__(self, m):
            self.m = m

        shape = property(lambda self: np.shape(self.m))

        def m_as(self, _unit):
            return self.m

        def __sub__(self, other):
            return Q(self.m - other.m)

    dec.ureg = types.SimpleNamespace(Quantity=Q)
    ed = 10 ** (rs.rand(n) * 3)
    ld = rs.uniform(10.0, 12800.0, n)
    out.update(dec_e=ed, dec_l=ld)
    dcases = [(np.deg2rad(33.0), np.deg2rad(8.0), np.deg2rad(50.0), 8e-5, 3e-3, 1e-11, 5e-10, 2.5e-13),
              (np.deg2rad(33.6), np.deg2rad(8.5), np.deg2rad(42.0), 7.5e-5, 2.457e-3, 0.0, 0.0, 0.0),
              (np.deg2rad(30.0), np.deg2rad(9.0), np.deg2rad(47.0), 7.0e-5, -2.4e-3, 1e-23, 1e-22, 3e-23)]
    out["dec_params"] = np.array(dcases)
    for ic, (t12, t13, t23, dm21, dm31, g21, g31, g32) in enumerate(dcases):
        # DecoherenceParams' constructor routes the angles through pint (OscParams' property keeps sin(theta) and hands
        # back arcsin of it): the functions get a plain namespace with that value
        thetas = {k: Q(float(np.arcsin(np.sin(v)))) for k, v in (("theta12", t12), ("theta13", t13), ("theta23", t23))}
        view = types.SimpleNamespace(dm21=Q(dm21), dm31=Q(dm31), dm32=Q(dm31 - dm21), gamma21=Q(g21), gamma31=Q(g31),
                                     gamma32=Q(g32), **thetas)
        for flav in ("nue", "numu"):
            pe, pm, pt = np.zeros(n), np.zeros(n), np.zeros(n)
            dec.calc_decoherence_probs(view, flav, Q(ed), Q(ld), pe, pm, pt, two_flavor=False)
            out["dec_%d_%s" % (ic, flav)] = np.stack([pe, pm, pt], axis=1)
    # --- reco.simple_param: the three smearing functions drawing from ONE RandomState(0), container after container
    ref_shim._pkg("pisa.stages.reco", os.path.join(p, "reco"))
    sp = ref_shim.ref_module("pisa.stages.reco.simple_param")
    er = 10 ** (rs.rand(300) * 2.5)
    czr = rs.rand(300) * 2 - 1
    out.update(reco_e=er, reco_cz=czr)
    state = np.random.RandomState(0)
    e_params = {"nu*_cc": [10.0, 0.3, -0.2], "*_nc": [10.0, 0.5, 0.1], "muons": [5.0, 0.6, 0.0]}
    cz_params = {"nu*_cc": [10.0, 0.4, -0.5], "*_nc": [10.0, 0.6, -0.3], "muons": [5.0, 0.1, 0.0]}
    pid_params = {"numu*_cc": [0.9, 0.3, 12.0], "nue*": [0.3, 0.1, 30.0], "nutau*": [0.3, 0.1, 30.0], "muons": [1.0, 1.0, 0.0]}
    for key in ("numu_cc", "nutau_cc", "nue_nc", "muons", "numubar_cc"):
        out["reco_%s_energy" % key] = sp.simple_reco_energy_parameterization(key, er, e_params, state)
        out["reco_%s_coszen" % key] = sp.simple_reco_coszen_parameterization(key, er, czr, cz_params, state)
        out["reco_%s_pid" % key] = sp.simple_pid_parameterization(key, er, pid_params, 1.0, 0.0, state)
    # the arrays of test_lookup_indices (bin_indexing.py:164-226): binnings 7 x 4 x 2 over [0,7] x [0,4] x [0,2]
    out["idx_test_x"] = np.array([-5, 0.5, 1.5, 7.0, 6.5, 8.0, 6.5])
    out["idx_test_y"] = np.array([-5, 0.5, 1.5, 1.5, 3.0, 1.5, 2.5])
    out["idx_test_z"] = np.array([-5, 0.5, 1.5, 1.5, 0.5, 6.0, 0.5])
    out["idx_test_1d"] = np.array([-1, 0, 1, 6, 6, 7, 6])
    out["idx_test_2d"] = np.array([-1, 0, 5, 25, 27, 28, 26])
    out["idx_test_3d"] = np.array([-1, 0, 11, 51, 54, 56, 52])
    save("side_stages_ref.npz", **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["pickles", "layers", "params", "lookup", "stats", "barr", "hist", "grid", "flux", "side", "stats_wide"]
    fns = dict(pickles=gen_ref_pickles, layers=gen_layers, params=gen_params, lookup=gen_lookup,
               stats=gen_stats, barr=gen_barr, hist=gen_hist, grid=gen_grid, flux=gen_flux, side=gen_side, stats_wide=gen_stats_wide)
    for w in which:
        fns[w]()
