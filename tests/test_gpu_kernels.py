"""Parity tests proper: every HIP kernel, called through the C ABI
(pisa_amd._lib / pisa_amd.kernels), against the CPU oracle and the committed
golden fixtures.  Needs a real MI355X:  pytest -m gpu.

Tolerances
  prob3 probabilities : rtol 1e-10, atol 1e-14  (the reference's own AC_KW,
                        numba_osc_tests.py:82; device sincos/atan2 differ from
                        glibc by <= 2 ulp so bit equality is not attainable)
  layers, lookups, bin indices, histogram limbs: bit exact
  histogram sums / maps: rtol 1e-12 (oracle sums sequentially in fp64, the
                        device sums exactly and rounds once)
  LLH / chi2 totals   : rtol 1e-10 (north star)
"""
import ctypes as C

import numpy as np
import pytest

from tests.conftest import PROB3_ATOL, PROB3_RTOL, load_golden

pytestmark = pytest.mark.gpu
AC = dict(rtol=PROB3_RTOL, atol=PROB3_ATOL)


@pytest.fixture(scope="module")
def K():
    import torch

    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from pisa_amd import kernels

    return kernels


@pytest.fixture(scope="module")
def L():
    from pisa_amd import _lib

    return _lib


def _args(g, case):
    return {k.split("::")[1]: g[k] for k in g.files if k.startswith(case + "::")}


def _cases(g, func):
    return sorted({k.split("::")[0] for k in g.files if k.startswith(func + "__")})


# ------------------------------------------------------------------ prob3
def test_propagate_array_reference_goldens(K, L):
    g = load_golden("prob3_ref_goldens.npz")
    cases = _cases(g, "propagate_scalar")
    assert len(cases) == 13
    for c in cases:
        a = _args(g, c)
        p = L.make_prob3_params(a["dm"], a["mix"], a["mat_pot"], int(a["decay_flag"]),
                                a["mat_decay"], a["lri_pot"])
        out = K.propagate_array(
            p, int(a["nubar"]), K.to_device([float(a["energy"])]), K.to_device(a["densities"]),
            K.to_device(a["distances"])).cpu().numpy()[0]
        np.testing.assert_allclose(out, a["probability"], err_msg=c, **AC)


def test_propagate_array_host_entry_point(L):
    """the numpy-in/numpy-out call a reference-side binding would use"""
    g = load_golden("prob3_ref_goldens.npz")
    a = _args(g, "propagate_scalar__nufit32_no")
    p = L.make_prob3_params(a["dm"], a["mix"], a["mat_pot"], -1, a["mat_decay"], a["lri_pot"])
    n = 37
    e = np.full(n, float(a["energy"]))
    out = np.full((n, 3, 3), np.nan)
    rho, dist = np.ascontiguousarray(a["densities"]), np.ascontiguousarray(a["distances"])
    L.check(L.lib().pisa_hip_propagate_array_host(
        C.byref(p), 1, e.ctypes.data, rho.ctypes.data, dist.ctypes.data, n, len(rho), 0,
        out.ctypes.data))
    assert np.all(out == out[0])  # broadcast test of numba_osc_tests.py:266-312
    np.testing.assert_allclose(out[0], a["probability"], **AC)
    # too many layers -> error status, as the hard 120 cap of the reference
    with pytest.raises(L.PisaHipError):
        L.check(L.lib().pisa_hip_propagate_array_host(
            C.byref(p), 1, e.ctypes.data, rho.ctypes.data, dist.ctypes.data, n, 121, 0,
            out.ctypes.data))


def test_prob3_grid_golden_and_oracle(K, L, oracle):
    g = load_golden("prob3_grid_prem12.npz")
    e, dens, dist = g["energy"], g["densities"], g["distances"]
    n_e, n_cz = len(e), dens.shape[0]
    for name in ("no", "io", "nsi", "decay"):
        p = L.make_prob3_params(g[name + "::dm"], g[name + "::mix"], g[name + "::mat_pot"],
                                int(g[name + "::decay_flag"]), g[name + "::mat_decay"],
                                g[name + "::lri_pot"])
        for e_major in (True, False):
            nu, nubar = K.prob3_grid(p, K.to_device(e), K.to_device(dens), K.to_device(dist),
                                     e_major=e_major)
            nu, nubar = nu.cpu().numpy(), nubar.cpu().numpy()
            if e_major:
                nu, nubar = nu.reshape(n_e, n_cz, 3, 3), nubar.reshape(n_e, n_cz, 3, 3)
            else:
                nu = nu.reshape(n_cz, n_e, 3, 3).transpose(1, 0, 2, 3)
                nubar = nubar.reshape(n_cz, n_e, 3, 3).transpose(1, 0, 2, 3)
            np.testing.assert_allclose(nu, g[name + "::prob_nu"], err_msg=name, **AC)
            np.testing.assert_allclose(nubar, g[name + "::prob_nubar"], err_msg=name, **AC)


@pytest.mark.parametrize("fused_amp", ["0", "split"])
def test_prob3_grid_planned_development_forms(fused_amp):
    """the two other forms of the planned kernels (layer matrices stored by stage AB and read back; one
    row per workgroup) exist in the development build only: tests/dev_cases.py on libpisa_hip_dev.so"""
    from tests.conftest import run_dev_case

    run_dev_case("planned_variant", fused_amp)


def test_prob3_grid_planned(K, L):
    """planned grid form (terms hoisted per (E, density), mirrored layers share one
    matrix, layer matrices formed inside the chain kernel from the per-density records, rows packed by
    length into 4-wave workgroups, chain multiplied in parts) == direct grid kernel to rounding (the
    product is associated differently), == reference goldens within the prob3
    tolerance; the compact (P_e, P_mu) gather tables are exact copies of P"""
    g = load_golden("prob3_grid_prem12.npz")
    e, dens, dist = K.to_device(g["energy"]), K.to_device(g["densities"]), K.to_device(g["distances"])
    n_e, n_cz = len(g["energy"]), g["densities"].shape[0]
    plan = K.GridPlan(dens, dist)
    # "io": the vacuum ordering of the eigenvalues (resolved on the host in this form) differs
    for name in ("no", "io", "nsi", "decay"):
        p = L.make_prob3_params(g[name + "::dm"], g[name + "::mix"], g[name + "::mat_pot"],
                                int(g[name + "::decay_flag"]), g[name + "::mat_decay"],
                                g[name + "::lri_pot"])
        for e_major in (True, False):
            nu, nubar, pepmu = K.prob3_grid(p, e, dens, dist, e_major=e_major, want_pepmu=True)
            nu2, nubar2, pepmu2 = K.prob3_grid_planned(p, plan, e, e_major=e_major)
            for a, b in ((nu, nu2), (nubar, nubar2), (pepmu, pepmu2)):
                assert float((a - b).abs().max()) < 3e-13
            pm = pepmu2.cpu().numpy()
            for side, P in ((0, nu2.cpu().numpy()), (1, nubar2.cpu().numpy())):
                for f in range(3):
                    np.testing.assert_array_equal(pm[side, f, :, 0], P[:, 0, f])  # fill_probs(P, 0, flav)
                    np.testing.assert_array_equal(pm[side, f, :, 1], P[:, 1, f])  # fill_probs(P, 1, flav)
            got_nu, got_nubar = nu2.cpu().numpy(), nubar2.cpu().numpy()
            if e_major:
                got_nu, got_nubar = got_nu.reshape(n_e, n_cz, 3, 3), got_nubar.reshape(n_e, n_cz, 3, 3)
            else:
                got_nu = got_nu.reshape(n_cz, n_e, 3, 3).transpose(1, 0, 2, 3)
                got_nubar = got_nubar.reshape(n_cz, n_e, 3, 3).transpose(1, 0, 2, 3)
            np.testing.assert_allclose(got_nu, g[name + "::prob_nu"], err_msg=name, **AC)
            np.testing.assert_allclose(got_nubar, g[name + "::prob_nubar"], err_msg=name, **AC)


def test_prob3_random_vs_oracle(K, L, oracle):
    """seeded random parameters / paths incl. zero-length layers and cache hits"""
    rs = np.random.RandomState(123)
    g = load_golden("prob3_grid_prem12.npz")
    n, nl = 2000, 12
    e = 10 ** (rs.rand(n) * 4 - 1)
    rho = rs.rand(n, nl) * 6
    dist = rs.rand(n, nl) * 800
    dist[rs.rand(n, nl) < 0.2] = 0.0
    rho[:, 7] = rho[:, 2]; dist[:, 7] = dist[:, 2] + 1e-9  # cache hit (numba_osc_kernels.py:236-241)
    rho[:, 9] = rho[:, 7]; dist[:, 9] = dist[:, 7]          # chained cache hit
    for name in ("io", "nsi", "decay"):
        pa = [g[name + "::dm"], g[name + "::mix"], g[name + "::mat_pot"],
              int(g[name + "::decay_flag"]), g[name + "::mat_decay"], g[name + "::lri_pot"]]
        p = L.make_prob3_params(*pa)
        for nubar in (1, -1):
            ref = oracle.propagate_array(*pa, nubar, e, rho, dist)
            out = K.propagate_array(p, nubar, K.to_device(e), K.to_device(rho),
                                    K.to_device(dist)).cpu().numpy()
            np.testing.assert_allclose(out, ref, err_msg="%s %d" % (name, nubar), **AC)


def test_prob3_empty_and_ragged(K, L):
    g = load_golden("prob3_grid_prem12.npz")
    p = L.make_prob3_params(g["no::dm"], g["no::mix"], g["no::mat_pot"], -1, g["no::mat_decay"],
                            g["no::lri_pot"])
    import torch

    empty = torch.empty(0, dtype=torch.float64, device="cuda")
    out = K.propagate_array(p, 1, empty, K.to_device(np.zeros((0, 4))), K.to_device(np.zeros((0, 4))))
    assert out.shape == (0, 3, 3)
    # n not a multiple of the workgroup size
    e = np.linspace(1, 50, 257)
    out = K.propagate_array(p, 1, K.to_device(e), K.to_device(g["densities"][3]),
                            K.to_device(g["distances"][3])).cpu().numpy()
    assert out.shape == (257, 3, 3) and np.all(np.isfinite(out))
    np.testing.assert_allclose(out.sum(axis=2), 1.0, rtol=1e-9)


# ----------------------------------------------------------------- layers
@pytest.mark.parametrize("tag", ["prem4", "prem4b", "prem12", "prem59", "prem10"])
def test_calc_layers_bit_exact(K, L, oracle, tag):
    g = load_golden("layers_ref.npz")
    earth = L.make_earth(g[tag + "::radii"], g[tag + "::rhos"], g[tag + "::coszen_limit"],
                         g[tag + "::prem"][-1, 0] - g[tag + "::args"][0])
    max_layers = 2 * len(g[tag + "::radii"])
    nl, dens, dist = K.calc_layers(earth, K.to_device(g[tag + "::cz"]), max_layers)
    np.testing.assert_array_equal(nl.cpu().numpy(), g[tag + "::n_layers"])
    np.testing.assert_array_equal(dens.cpu().numpy(), g[tag + "::density"])
    np.testing.assert_array_equal(dist.cpu().numpy(), g[tag + "::distance"])


def test_prob3_events_vs_oracle(K, L, oracle):
    """event mode: in-kernel layers == oracle layers + oracle propagate"""
    import torch

    gl = load_golden("layers_ref.npz")
    gg = load_golden("prob3_grid_prem12.npz")
    rs = np.random.RandomState(5)
    n = 3000
    e = 10 ** (rs.rand(n) * 3)
    cz = rs.rand(n) * 2 - 1
    # "prem12_equal": two mantle shells given the same density -- the reference's layer cache may then
    # hand a layer the matrix of a DIFFERENT shell, which the kernel's general (staged) form resolves;
    # the other two models have pairwise distinct densities and take the direct form
    for tag in ("prem12", "prem4", "prem12_equal"):
        base = "prem12" if tag == "prem12_equal" else tag
        depth, height, yi, yo, ym = gl[base + "::args"]
        prem = np.array(gl[base + "::prem"], dtype=np.float64, copy=True)
        if tag == "prem12_equal":
            prem[6, 1] = prem[7, 1]
        lay = oracle.Layers(prem, depth, height)
        lay.setElecFrac(yi, yo, ym)
        lay.calcLayers(cz)
        earth = L.make_earth(lay.radii, lay.rhos, lay.coszen_limit, lay.r_detector)
        gp = load_golden("params_ref.npz")
        for name in ("no", "io", "nsi", "decay", "lri"):
            src = "no" if name == "lri" else name
            pa = [gg[src + "::dm"], gg[src + "::mix"], gg[src + "::mat_pot"],
                  int(gg[src + "::decay_flag"]), gg[src + "::mat_decay"], gg[src + "::lri_pot"]]
            if name == "lri":      # long-range-interaction potential (lri_params.py:31-108): the XL matrix
                pa[5] = np.array(gp["lri::mutau"], dtype=np.float64)
                assert np.abs(pa[5]).max() > 0
            p = L.make_prob3_params(*pa)
            for nubar in (1, -1):
                ref = oracle.propagate_array(*pa, nubar, e, lay.density, lay.distance)
                if name == "lri":   # the potential is not a no-op at these energies
                    plain = oracle.propagate_array(*(pa[:5] + [np.zeros((3, 3))]), nubar, e, lay.density, lay.distance)
                    assert np.abs(ref - plain).max() > 1e-4
                out = K.prob3_events(p, earth, nubar, K.to_device(e), K.to_device(cz)).cpu().numpy()
                np.testing.assert_allclose(out, ref, err_msg="%s %s %d" % (tag, name, nubar), **AC)
                # the gather pair alone (what the fused reweighting asks for: P[e -> flav], P[mu -> flav]):
                # the same bits as the full matrices
                d_e, d_cz = K.to_device(e), K.to_device(cz)
                pairs = [torch.full((n, 2), np.nan, dtype=torch.float64, device="cuda") for _ in range(3)]
                sets = [L.EventSet(n, d_e.data_ptr(), d_cz.data_ptr(), None, pairs[f].data_ptr(), nubar, f)
                        for f in range(3)]
                status = torch.zeros(1, dtype=torch.int32, device="cuda")
                K.prob3_events_multi(p, earth, sets, status)
                assert int(status.item()) == 0
                for f in range(3):
                    np.testing.assert_array_equal(pairs[f].cpu().numpy(), out[:, :2, f],
                                                  err_msg="%s %s %d pair %d" % (tag, name, nubar, f))


# ------------------------------------------------------------ translation
def test_lookup_golden(K, L):
    g = load_golden("lookup_ref.npz")
    x, y, z = (K.to_device(g[k]) for k in "xyz")
    out = K.lookup_regular([x], K.to_device(g["h1"]), L.make_binning([0.0], [1.0], [7]))
    np.testing.assert_array_equal(out.cpu().numpy(), g["o1"])
    b2 = L.make_binning([0.0, -1.0], [1.0, 1.0], [7, 5])
    out = K.lookup_regular([x, y], K.to_device(g["h2"]), b2)
    np.testing.assert_array_equal(out.cpu().numpy(), g["o2"])
    out = K.lookup_regular([x, y, z], K.to_device(g["h3"]),
                           L.make_binning([0.0, -1.0, 0.0], [1.0, 1.0, 2.0], [7, 5, 3]))
    np.testing.assert_array_equal(out.cpu().numpy(), g["o3"])
    out = K.lookup_regular([x, y], K.to_device(g["h2a"]), b2)
    np.testing.assert_array_equal(out.cpu().numpy(), g["o2a"])


def test_histogram_golden_recipe(K, L):
    """translation.py:779-818: == np.histogramdd, summed and averaged"""
    g = load_golden("hist_ref.npz")
    nbs = [2, 3, 4]
    sample = []
    w = K.to_device(g["weights"])
    for nd in (1, 2, 3):
        sample.append(K.to_device(g["s%d" % (nd - 1)]))
        b = L.make_binning([0.0] * nd, [float(v) for v in nbs[:nd]], nbs[:nd])
        h = K.histogram_regular(sample, w, b).cpu().numpy()
        np.testing.assert_allclose(h, g["ref%dd" % nd], rtol=1e-13)
        c = K.histogram_regular(sample, None, b).cpu().numpy()
        np.testing.assert_array_equal(c, g["cnt%dd" % nd])
        avg = K.histogram_regular(sample, w, b, averaged=True).cpu().numpy()
        np.testing.assert_allclose(avg, g["ref%dd" % nd] / g["cnt%dd" % nd], rtol=1e-13)


def test_histogram_edges_empty_negative_and_large(K, L, oracle):
    x = np.array([0.0, 1.0, np.nextafter(1.0, 0), -1e-300, np.nan, 0.5, np.inf])
    b = L.make_binning([0.0], [1.0], [4])
    np.testing.assert_array_equal(K.histogram_regular([K.to_device(x)], None, b).cpu().numpy(),
                                  [1, 0, 1, 1])
    import torch

    e = torch.empty(0, dtype=torch.float64, device="cuda")
    np.testing.assert_array_equal(K.histogram_regular([e], e, b).cpu().numpy(), [0, 0, 0, 0])
    # empty bins average to 0 (NaN -> 0, translation.py:125-127)
    avg = K.histogram_regular([K.to_device([0.1])], K.to_device([2.0]), b, averaged=True)
    np.testing.assert_array_equal(avg.cpu().numpy(), [2.0, 0, 0, 0])
    # signed weights spanning 40 orders of magnitude: exact accumulation
    rs = np.random.RandomState(9)
    n = 200000
    xs = rs.rand(n)
    w = rs.randn(n) * 10 ** (rs.rand(n) * 40 - 30)
    h = K.histogram_regular([K.to_device(xs)], K.to_device(w), b).cpu().numpy()
    import math

    idx = np.minimum((xs * 4).astype(int), 3)
    exact = np.array([math.fsum(w[idx == k]) for k in range(4)])
    np.testing.assert_allclose(h, exact, rtol=1e-15, atol=1e-34)
    # many bins -> global-accumulator path (no LDS privatisation)
    nb = [200, 200]
    b2 = L.make_binning([0.0, 0.0], [1.0, 1.0], nb)
    ys = rs.rand(n)
    ww = rs.rand(n)
    h2 = K.histogram_regular([K.to_device(xs), K.to_device(ys)], K.to_device(ww), b2).cpu().numpy()
    ref = oracle.histogram_regular([xs, ys], ww, [0.0, 0.0], [1.0, 1.0], nb)
    np.testing.assert_allclose(h2, ref, rtol=1e-13)
    # columns that are only 8-byte aligned (views one element into a buffer) take the kernel's scalar
    # path instead of the 16-byte pair loads: same bits, odd and even lengths, 1-3 dimensions
    cols3 = [rs.rand(n + 1) for _ in range(3)]
    w3 = rs.rand(n + 1)
    for nd in (1, 2, 3):
        b3 = L.make_binning([0.0] * nd, [1.0] * nd, [5, 4, 3][:nd])
        for m in (n, n - 1):
            aligned = K.histogram_regular([K.to_device(c[1:m + 1]) for c in cols3[:nd]], K.to_device(w3[1:m + 1]), b3)
            shifted = K.histogram_regular([K.to_device(c)[1:m + 1] for c in cols3[:nd]], K.to_device(w3)[1:m + 1], b3)
            assert K.to_device(cols3[0])[1:m + 1].data_ptr() % 16 == 8
            assert torch.equal(aligned, shifted)
            ref3 = oracle.histogram_regular([c[1:m + 1] for c in cols3[:nd]], w3[1:m + 1], [0.0] * nd, [1.0] * nd,
                                            [5, 4, 3][:nd])
            np.testing.assert_allclose(aligned.cpu().numpy(), ref3.reshape(-1), rtol=1e-13)
    # non-finite weight is an error, not a silent NaN bin
    with pytest.raises(OverflowError):
        K.histogram_regular([K.to_device([0.1])], K.to_device([np.inf]), b)


# ------------------------------------------- fused reweight + hist + metric
def test_fused_reweight_hist_vs_oracle(K, L, oracle):
    from pisa_amd import synthetic

    wl = synthetic.Workload(n_events=120000, grid=(40, 20), out_binning="dragon", seed=3)
    st = synthetic.DeviceState(wl)
    st.eval(wl.osc_params(theta23_deg=44.0, dm31=2.5e-3))
    hist = st.ws.hist.cpu().numpy()
    sumw2 = st.ws.sumw2.cpu().numpy()
    from oracle.pipeline_oracle import oracle_eval

    ref = oracle_eval(wl)
    np.testing.assert_allclose(hist, ref["hist"], rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(sumw2, ref["sumw2"], rtol=1e-12, atol=1e-300)
    # probability tables feeding the fused kernel
    np.testing.assert_allclose(st.prob_nu.cpu().numpy(), ref["prob_nu"], **AC)
    np.testing.assert_allclose(st.prob_nubar.cpu().numpy(), ref["prob_nubar"], **AC)
    # LLH / mod_chi2 against pseudo-data
    data = np.random.RandomState(0).poisson(ref["hist"].sum(axis=0)).astype(float)
    for kind in ("llh", "mod_chi2", "poisson_llh", "chi2"):
        got = float(K.metric(kind, K.to_device(data), st.ws.hist, st.ws.sumw2).item())
        _, want = oracle.metric(kind, data, ref["hist"].sum(axis=0), ref["sumw2"].sum(axis=0))
        np.testing.assert_allclose(got, want, rtol=1e-10, err_msg=kind)


def test_fused_bit_reproducible_and_shardable(K, L):
    """run-to-run identical; event shards summed as integers == unsharded
    (this is what makes the LLH independent of the GPU count)"""
    from pisa_amd import synthetic

    wl = synthetic.Workload(n_events=60000, grid=(20, 10), out_binning="dragon", seed=1)
    p = wl.osc_params()
    full = synthetic.DeviceState(wl)
    full.make_pseudo_data(p)
    full.accumulate(p)
    l1 = full.ws.limbs.clone()
    full.finalize()
    h1 = full.ws.hist.clone()
    full.accumulate(p)
    assert bool((full.ws.limbs == l1).all())
    # tail of an evaluation: one fused finalize+metric launch (leaves the limbs
    # zeroed for the next accumulate) == the two separate kernels, bit for bit
    v_fused = float(full.eval(p).item())
    assert bool((full.ws.limbs == 0).all()) and bool((full.ws.hist == h1).all())
    v_again = float(full.eval(p).item())  # accumulates onto the zeroed limbs, no memset
    full.fused_tail = False
    v_sep = float(full.eval(p).item())
    assert bool((full.ws.limbs == l1).all()) and bool((full.ws.hist == h1).all())
    full.fused_tail = True
    v_host = full.eval_host(p)  # metric written straight into pinned host memory
    assert v_fused == v_again == v_sep == v_host
    full.check_status()
    total = None
    for rank in range(3):
        sh = synthetic.DeviceState(wl, rank=rank, world_size=3)
        sh.accumulate(p)
        total = sh.ws.limbs.clone() if total is None else total + sh.ws.limbs
    sh.ws.limbs.copy_(total)
    K.hist_finalize(sh.ws)
    assert bool((sh.ws.hist == h1).all())  # bit identical maps
    llh_a = K.metric("llh", full.data, full.ws.hist, full.ws.sumw2)
    llh_b = K.metric("llh", full.data, sh.ws.hist, sh.ws.sumw2)
    assert float(llh_a.item()) == float(llh_b.item())


def test_indexed_equals_coordinate_form(K, L, oracle):
    """pre-digitised (node, bin) columns + gather tables give the same exact
    limbs as binning the coordinates on the fly; indices match the oracle rule"""
    from pisa_amd import synthetic

    wl = synthetic.Workload(n_events=50001 * 12, grid=(30, 20), out_binning="dragon", seed=5)
    p = wl.osc_params(theta23_deg=47.0)
    a = synthetic.DeviceState(wl, indexed=True, planned=True)
    b = synthetic.DeviceState(wl, indexed=False, planned=True)
    a.accumulate(p)
    b.accumulate(p)
    # same exact sums (the un-normalised limb words may differ between kernels)
    ha, sa = (t.clone() for t in a.finalize())
    hb, sb = b.finalize()
    assert bool((ha == hb).all()) and bool((sa == sb).all())
    ev = wl.events[4]
    idx = K.event_indices([K.to_device(s) for s in ev["sample"]], wl.out_binning).cpu().numpy()
    ob = wl.ob
    ones = np.ones(len(idx))
    ref = oracle.histogram_regular(ev["sample"], ones, ob["mins"], ob["maxs"], ob["nbins"])
    np.testing.assert_array_equal(np.bincount(idx[idx >= 0], minlength=wl.n_bins), ref)
    assert (idx < 0).sum() == len(idx) - int(ref.sum())


def test_unfused_stage_kernels(K, oracle):
    rs = np.random.RandomState(2)
    n = 10001
    w0, flux, pe, pmu, aeff = rs.rand(n), rs.rand(n, 2), rs.rand(n), rs.rand(n), rs.rand(n)
    w = K.to_device(w0)
    K.apply_osc_weights(K.to_device(flux), K.to_device(pe), K.to_device(pmu), w)
    K.apply_aeff(K.to_device(aeff), 3.25, w)
    np.testing.assert_array_equal(w.cpu().numpy(), oracle.reweight(w0, flux, pe, pmu, aeff, 3.25))
    P = rs.rand(n, 3, 3)
    np.testing.assert_array_equal(K.fill_probs(K.to_device(P), 1, 2).cpu().numpy(), P[:, 1, 2])


def test_metric_golden_and_errors(K):
    g = load_golden("stats_ref.npz")
    a, e = K.to_device(g["actual"]), K.to_device(g["expected"])
    for name in ("llh", "poisson_llh", "chi2", "mod_chi2"):
        total, pb = K.metric(name, a, e, per_bin=True)
        np.testing.assert_allclose(pb.cpu().numpy(), g[name], rtol=1e-12, equal_nan=True)
        np.testing.assert_allclose(float(total.item()), float(g[name + "_total"]), rtol=1e-12)
    with pytest.raises(ValueError):
        K.metric("llh", K.to_device([-1.0, 2.0]), K.to_device([1.0, 2.0]))
    # chi2 returns exactly 0 when all |delta| < 5 eps (stats.py:160-161)
    t = K.metric("chi2", K.to_device([1.0, 2.0]), K.to_device([1.0, 2.0]))
    assert float(t.item()) == 0.0
    # the free functions of utils/stats.py: per-bin values in the shape of the inputs
    from pisa_amd.core.map import Map
    from pisa_amd.utils import stats

    shape = (2, -1) if g["actual"].size % 2 == 0 else (1, -1)
    a2, e2 = g["actual"].reshape(shape), g["expected"].reshape(shape)
    for name in ("llh", "poisson_llh", "chi2", "mod_chi2"):      # (the others: test_wide_metrics_... below)
        got = getattr(stats, name)(a2, e2)
        assert got.shape == a2.shape
        np.testing.assert_allclose(got, g[name].reshape(shape), rtol=1e-12, equal_nan=True)
    sigma = 0.3 * np.sqrt(e2)
    want = (a2 - np.clip(e2, stats.SMALL_POS, None)) ** 2 / (sigma ** 2 + np.clip(e2, stats.SMALL_POS, None))
    np.testing.assert_allclose(stats.mod_chi2(a2, e2, sigma=sigma), want, rtol=1e-12)
    binning = [dict(name="x", num_bins=a2.shape[0], domain=[0, 1]), dict(name="y", num_bins=a2.shape[1], domain=[0, 1])]
    exp_map = Map(name="e", hist=e2, binning=binning, error_hist=sigma)
    np.testing.assert_allclose(stats.mod_chi2(Map(name="a", hist=a2, binning=binning), exp_map), want, rtol=1e-12)
    with pytest.raises(ValueError):
        stats.chi2(a2, e2.ravel())


def test_barr_flux_golden(K):
    g = load_golden("barr_ref.npz")
    args = [K.to_device(g[k]) for k in ("true_energy", "true_coszen", "nu_flux_nominal",
                                        "nubar_flux_nominal")]
    for ip, ps in enumerate(g["params"]):
        for nubar, tag in ((1, "nu"), (-1, "nubar")):
            out = K.barr_simple(*args, nubar, *ps).cpu().numpy()
            np.testing.assert_allclose(out, g["out%d_%s" % (ip, tag)], rtol=1e-12, atol=1e-300)
    # all containers of a pipeline in one launch (what the stage calls): ragged, one of them empty,
    # more sets than one launch holds -- every set bit-identical to the single-container call
    import torch

    n = args[0].numel()
    sizes = [n, 1, 0, n - 3] + [7 + k for k in range(16)]
    signs = [1, -1, 1, -1] + [(-1) ** k for k in range(16)]
    cols = [tuple(a[:m].contiguous() for a in args) + (sg, torch.full((m, 2), np.nan, dtype=torch.float64, device="cuda"))
            for m, sg in zip(sizes, signs)]
    ps = g["params"][-1]
    K.barr_simple_multi(K.barr_sets(cols), *ps)
    for (e, cz, nu, nub, sg, out), m in zip(cols, sizes):
        if m:
            assert torch.equal(out, K.barr_simple(e, cz, nu, nub, sg, *ps))


def test_event_mode_engine_vs_oracle(K, L, oracle):
    """config C2/C5 shape: prob3 event by event (in-kernel layers, all containers
    in one launch) + fused reweight + histogram, NSI matter potential"""
    from oracle.pipeline_oracle import oracle_eval_events
    from pisa_amd import synthetic

    g = load_golden("prob3_grid_prem12.npz")
    wl = synthetic.Workload(n_events=36000, grid=(10, 10), out_binning="example2d", seed=8)
    st = synthetic.DeviceState(wl, osc_mode="events")
    p = wl.osc_params(theta23_deg=48.0, deltacp_deg=200.0, mat_pot=g["nsi::mat_pot"])
    st.accumulate(p)
    st.finalize()
    st.check_status()
    hist, sumw2 = st.maps()
    ref = oracle_eval_events(wl)
    np.testing.assert_allclose(hist, ref["hist"], rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(sumw2, ref["sumw2"], rtol=1e-10, atol=1e-300)
    assert hist.sum() > 0


def test_profile_hook_is_per_host_thread():
    """`pisa_hip_profile_events` (include/pisa_hip.h): the event pair belongs to the calling host
    thread.  A launch from another thread neither records the pair nor clears it; the next launch
    of the owning thread does."""
    import threading

    import torch

    from pisa_amd import _lib
    from pisa_amd import kernels as K

    lib = _lib.lib()
    x = K.to_device(np.random.RandomState(0).rand(200000))
    w = K.to_device(np.ones(200000))
    b = _lib.make_binning([0.0], [1.0], [10])
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record(); stop.record()          # materialise the handles
    torch.cuda.synchronize()
    assert start.elapsed_time(stop) < 1.0  # back to back: nothing in between
    assert lib.pisa_hip_profile_events(start.cuda_event, stop.cuda_event) == 0
    res = {}

    def other():
        with torch.cuda.stream(torch.cuda.Stream()):
            big = K.to_device(np.random.RandomState(1).rand(4000000))
            for _ in range(20):
                res["h"] = K.histogram_regular([big], torch.ones_like(big), b)
            torch.cuda.synchronize()

    t = threading.Thread(target=other)
    t.start(); t.join()
    torch.cuda.synchronize()
    assert abs(float(res["h"].sum()) - 4000000) < 1e-6
    assert start.elapsed_time(stop) < 1.0  # the other thread's launches did not touch the pair
    h = K.histogram_regular([x], w, b)     # this thread's launch records it
    torch.cuda.synchronize()
    assert lib.pisa_hip_profile_events(None, None) == 0
    assert abs(float(h.sum()) - 200000) < 1e-9
    dt = start.elapsed_time(stop)
    assert 0.0 < dt < 50.0
    K.histogram_regular([x], w, b)         # disabled: the pair keeps its last recording
    torch.cuda.synchronize()
    assert start.elapsed_time(stop) == dt


def test_limb_decoder_is_correctly_rounded_on_adversarial_accumulators(K, L):
    """`hist_finalize_kernel` / the tail kernel decode an accumulator (six un-normalised int64 limbs,
    value = sum limb_j 2^(32 j - 116)) with ONE rounding to nearest-even.  Checked against exact
    rational arithmetic on accumulators built to hurt: negative and mixed-sign limbs, carries that
    ripple through every limb, values exactly halfway between two doubles (ties to even, both
    directions, with and without a sticky bit far below), single bits at either end of the range,
    zero, and random fills of every magnitude."""
    import torch

    from pisa_amd.engine import limbs_to_float

    rs = np.random.RandomState(9)
    cases = []
    cases.append([0] * 6)
    cases.append([1, 0, 0, 0, 0, 0])                     # 2^-116
    cases.append([-1, 0, 0, 0, 0, 0])
    cases.append([0, 0, 0, 0, 0, 1 << 30])               # near the top of the range
    cases.append([0, 0, 0, 0, 0, -(1 << 30)])
    cases.append([0xFFFFFFFF] * 5 + [0])                 # carries everywhere
    cases.append([-0xFFFFFFFF] * 5 + [1])                # borrows everywhere
    cases.append([(1 << 62) - 1] * 6)                    # heavily un-normalised sums
    cases.append([-(1 << 62)] * 5 + [1 << 20])
    # ties: a 54-bit pattern whose lowest bit is exactly half an ulp, placed at several offsets
    for shift in (0, 5, 31, 32, 40, 63, 64, 77, 100):
        for mant in ((1 << 53) | 1, (1 << 53) | 3, (1 << 54) - 1, (1 << 53) + 2 + 1):
            for sticky in (0, 1):
                for sign in (1, -1):
                    total = sign * ((mant << (shift + 1)) + (sticky if shift > 0 else 0))
                    limbs, t = [], total
                    for _ in range(5):
                        limbs.append(t & 0xFFFFFFFF)
                        t >>= 32
                    limbs.append(t)
                    if abs(limbs[5]) < (1 << 62):
                        cases.append(limbs)
    for _ in range(3000):
        bits = rs.randint(1, 63, size=6)
        vals = [int(rs.randint(0, 2 ** 31)) << 31 | int(rs.randint(0, 2 ** 31)) for _ in range(6)]
        limbs = [(v & ((1 << int(b)) - 1)) * (1 if rs.rand() < 0.6 else -1) for v, b in zip(vals, bits)]
        if rs.rand() < 0.3:
            for k in rs.choice(6, size=rs.randint(1, 5), replace=False):
                limbs[k] = 0
        cases.append(limbs)
    n = len(cases)
    want = np.array([limbs_to_float(c) for c in cases])
    ok = np.isfinite(want) & (np.abs(want) < 2.0 ** 76)      # the accumulators' range: 6 x 32 - 116 bits
    ws = K.HistWorkspace(1, n)
    arr = np.zeros((1, n, 2, 6), dtype=np.int64)
    arr[0, :, 0, :] = np.array(cases, dtype=object).astype(np.int64)
    arr[0, :, 1, :] = arr[0, ::-1, 0, :]                     # the second quantity: the same cases reversed
    ws.limbs.copy_(torch.from_numpy(arr))
    hist, sumw2 = K.hist_finalize(ws)
    got, got2 = hist.cpu().numpy()[0], sumw2.cpu().numpy()[0]
    bad = np.nonzero(ok & (got != want))[0]
    assert bad.size == 0, [(cases[i], got[i], want[i]) for i in bad[:5]]
    bad2 = np.nonzero(ok[::-1] & (got2 != want[::-1]))[0]
    assert bad2.size == 0
    assert ok.sum() > 1500
    # beyond the range the status word says so (the engine raises on it)
    assert (~ok).sum() > 0 and int(ws.status.item()) != 0
    ws2 = K.HistWorkspace(1, int(ok.sum()))
    ws2.limbs.copy_(torch.from_numpy(np.ascontiguousarray(arr[:, ok][:, :, [0, 0]])))
    K.hist_finalize(ws2)
    assert int(ws2.status.item()) == 0


@pytest.mark.parametrize("n_cont,n_bins", [(12, 128), (3, 10), (1, 1), (2, 3), (5, 257), (4, 700), (1, 4096)])
def test_split_tail_equals_one_workgroup_tail_bit_for_bit(K, L, n_cont, n_bins):
    """`pisa_hip_finalize_metric_split` (four workgroups per point, bins k mod 4 each, partial sums joined by the
    caller as (p0 + p2) + (p1 + p3)) against `pisa_hip_finalize_metric_multi` on the same limbs: metric value, maps,
    cleared limbs and status identical, for every metric it takes, with and without the per-bin scales / extra maps,
    one and three points, bin counts that are not multiples of four and that need several rounds of the 256-wide tree;
    a negative expectation turns the sum into NaN with the status word set in both; chi2 is refused."""
    import torch

    lib = L.lib()
    rs = np.random.RandomState(n_cont * 1000 + n_bins)
    dev = K.device()
    for n_pts in (1, 3):
        shape = (n_pts, n_cont, n_bins, 2, 6)
        fill = np.zeros(shape, dtype=np.int64)
        fill[..., 1:5] = rs.randint(0, 2 ** 36, size=shape[:-1] + (4,))
        fill[..., 3] += rs.randint(0, 2 ** 31, size=shape[:-1]) << 8
        if n_bins > 2:
            fill[:, 0, 1, :, :] = 0          # an empty bin
        fill_d = torch.from_numpy(fill).to(dev)
        data = torch.from_numpy(rs.poisson(40.0, n_bins).astype(np.float64)).to(dev)
        scale = torch.from_numpy(rs.uniform(0.5, 1.5, size=(n_cont, n_bins))).to(dev)
        extra = torch.from_numpy(rs.uniform(0.0, 3.0, size=(2, n_bins))).to(dev)
        for kind in ("llh", "poisson_llh", "mod_chi2"):
            for with_scale in (False, True):
                res = []
                # one workgroup | four (`_split`) | four and sixteen through `pisa_hip_finalize_metric_parts`
                for parts, entry in ((1, "multi"), (4, "split"), (4, "parts"), (16, "parts")):
                    limbs = fill_d.clone()
                    hist = torch.full((n_pts, n_cont, n_bins), -7.0, dtype=torch.float64, device=dev)
                    sumw2 = torch.full_like(hist, -7.0)
                    tot = torch.full((n_pts * 16,), float("nan"), dtype=torch.float64, device=dev)
                    st = torch.zeros(1, dtype=torch.int32, device=dev)
                    mst = torch.zeros(1, dtype=torch.int32, device=dev)
                    head = (limbs.data_ptr(), n_pts, n_cont, n_bins, hist.data_ptr(), sumw2.data_ptr(),
                            K.METRIC_KIND[kind], data.data_ptr(), scale.data_ptr() if with_scale else None, 0,
                            extra.data_ptr() if with_scale else None, tot.data_ptr())
                    rest = (st.data_ptr(), mst.data_ptr(), 1, None)
                    if entry == "parts":
                        rc = lib.pisa_hip_finalize_metric_parts(*head, parts, *rest)
                    else:
                        rc = (lib.pisa_hip_finalize_metric_split if entry == "split"
                              else lib.pisa_hip_finalize_metric_multi)(*head, *rest)
                    assert rc == 0
                    torch.cuda.synchronize()
                    t = tot.cpu().numpy()
                    if parts > 1:
                        vals = []
                        for p in t[: n_pts * parts].reshape(n_pts, parts):
                            p = [float(v) for v in p]
                            w = parts // 2
                            while w >= 1:            # the kernel's reduction tree, its last levels
                                for i in range(w):
                                    p[i] = p[i] + p[i + w]
                                w //= 2
                            vals.append(p[0])
                    else:
                        vals = [float(v) for v in t[:n_pts]]
                    assert int(limbs.abs().sum().item()) == 0
                    res.append((vals, hist.cpu().numpy(), sumw2.cpu().numpy(), int(st.item()), int(mst.item())))
                v0, h0, s0, st0, m0 = res[0]
                for v1, h1, s1, st1, m1 in res[1:]:
                    assert all(np.isfinite(v0)) and v0 == v1, (kind, with_scale, v0, v1)
                    assert np.array_equal(h0, h1) and np.array_equal(s0, s1) and (h0 != -7.0).all()
                    assert (st0, m0) == (st1, m1) == (0, 0)
    # negative observed count: NaN + status in both forms
    bad = data.clone(); bad[n_bins // 2] = -1.0
    for split in (False, True):
        limbs = fill_d[:1].clone()
        hist = torch.empty((1, n_cont, n_bins), dtype=torch.float64, device=dev); sumw2 = torch.empty_like(hist)
        tot = torch.zeros(4, dtype=torch.float64, device=dev)
        st = torch.zeros(1, dtype=torch.int32, device=dev); mst = torch.zeros(1, dtype=torch.int32, device=dev)
        fn = lib.pisa_hip_finalize_metric_split if split else lib.pisa_hip_finalize_metric_multi
        assert fn(limbs.data_ptr(), 1, n_cont, n_bins, hist.data_ptr(), sumw2.data_ptr(), K.METRIC_KIND["llh"],
                  bad.data_ptr(), None, 0, None, tot.data_ptr(), st.data_ptr(), mst.data_ptr(), 1, None) == 0
        torch.cuda.synchronize()
        t = tot.cpu().numpy()
        v = (t[0] + t[2]) + (t[1] + t[3]) if split else t[0]
        assert v != v and int(mst.item()) != 0
    assert lib.pisa_hip_finalize_metric_split(limbs.data_ptr(), 1, n_cont, n_bins, hist.data_ptr(), sumw2.data_ptr(),
                                            K.METRIC_KIND["chi2"], data.data_ptr(), None, 0, None, tot.data_ptr(),
                                            st.data_ptr(), mst.data_ptr(), 1, None) != 0


def test_apply_osc_weights_reads_strided_columns_in_place(K):
    """prob_e / prob_mu handed over as columns of one table (the gather tables' (P_e, P_mu) pairs, stride 2; a 3 x 3
    table's entries, stride 9): same bits as with compacted copies"""
    import torch

    rs = np.random.RandomState(2)
    n = 40001
    flux = K.to_device(rs.rand(n, 2))
    for width, ce, cm in ((2, 0, 1), (9, 1, 4)):
        tab = K.to_device(rs.rand(n, width))
        w0 = K.to_device(rs.rand(n))
        a, b = w0.clone(), w0.clone()
        K.apply_osc_weights(flux, tab[:, ce], tab[:, cm], a)
        K.apply_osc_weights(flux, tab[:, ce].contiguous(), tab[:, cm].contiguous(), b)
        assert torch.equal(a, b) and not torch.equal(a, w0)
        want = w0.cpu().numpy() * ((flux.cpu().numpy()[:, 0] * tab.cpu().numpy()[:, ce]) + (flux.cpu().numpy()[:, 1] * tab.cpu().numpy()[:, cm]))
        assert np.array_equal(a.cpu().numpy(), want)
    # columns of different strides: compacted, same numbers
    t2, t9 = K.to_device(rs.rand(n, 2)), K.to_device(rs.rand(n, 9))
    a, b = w0.clone(), w0.clone()
    K.apply_osc_weights(flux, t2[:, 1], t9[:, 3], a)
    K.apply_osc_weights(flux, t2[:, 1].contiguous(), t9[:, 3].contiguous(), b)
    assert torch.equal(a, b)


def test_wide_metrics_reference_vectors_and_restatement():
    """`pisa_hip_metric` kinds 4-8 (correct_chi2, signed_sqrt_mod_chi2, mcllh_mean, mcllh_eff, conv_llh) against the
    reference's own values (tests/golden/stats_wide_ref.npz), against the restatement on larger seeded maps, through
    `Map.metric` / `utils.stats`, and their sign rules (stats.py:359-368: the mcllh pair refuses negative inputs, the
    chi2 family and conv_llh only clip)"""
    import os

    from oracle import stages_oracle as so
    from pisa_amd import kernels as K
    from pisa_amd.core.binning import MultiDimBinning, OneDimBinning
    from pisa_amd.core.map import Map
    from pisa_amd.utils import stats

    W = np.load(os.path.join(os.path.dirname(__file__), "golden", "stats_wide_ref.npz"))
    kinds = ("mcllh_mean", "mcllh_eff", "correct_chi2", "signed_sqrt_mod_chi2", "conv_llh")

    def dev(a):
        return K.to_device(np.ascontiguousarray(a, dtype=np.float64))

    def scale_of(kind, k, lam, s):
        """size of the terms a value is the difference of (gamma functions of ~lam^2 / sigma^2 cancel in the mixture)"""
        if kind.startswith("mcllh"):
            with np.errstate(all="ignore"):
                alpha = np.where(s > 0, np.maximum(lam, 1e-10) ** 2 / np.maximum(s, 1e-300) ** 2, 0.0)
            return 1.0 + (k + alpha) * (1.0 + np.abs(np.log(np.maximum(k + alpha, 1e-300))))
        return 1.0

    for kind in kinds:
        total, per_bin = K.metric(kind, dev(W["actual"]), dev(W["expected"]), dev(W["sigma"] ** 2), per_bin=True)
        got = per_bin.cpu().numpy()
        tol = 1e-12 * scale_of(kind, W["actual"], W["expected"], W["sigma"])
        assert np.all(np.abs(got - W[kind]) <= tol + 1e-12 * np.abs(W[kind])), (kind, np.abs(got - W[kind]).max())
        np.testing.assert_allclose(float(total.item()), np.nansum(got), rtol=1e-13)
    rs = np.random.RandomState(3)
    n = 6000                                                      # the two-stage reduction (> 4096 bins)
    lam = rs.rand(n) * 40 + 0.01
    s = np.sqrt(lam) * rs.rand(n)
    s[:50] = 0.0
    k = rs.poisson(lam).astype(np.float64)
    for kind in kinds:
        total, per_bin = K.metric(kind, dev(k), dev(lam), dev(s ** 2), per_bin=True)
        want = so.metric_wide(kind, k, lam, s)
        got = per_bin.cpu().numpy()
        tol = 1e-12 * scale_of(kind, k, lam, s)
        assert np.all(np.abs(got - want) <= tol + 1e-11 * np.abs(want)), (kind, np.abs(got - want).max())
        assert abs(float(total.item()) - np.nansum(got)) <= 1e-13 * np.abs(got).sum()     # the reduction itself
    # Map / stats front ends
    b = MultiDimBinning([OneDimBinning(name="x", num_bins=16, domain=[0, 1]), OneDimBinning(name="y", num_bins=10, domain=[0, 1])])
    data = Map("data", k[:160].reshape(16, 10), b)
    templ = Map("t", lam[:160].reshape(16, 10), b, error_hist=s[:160].reshape(16, 10))
    for kind in kinds:
        want = so.metric_wide(kind, k[:160], lam[:160], s[:160])
        binned = getattr(data, kind)(templ, binned=True)
        assert binned.shape == (16, 10)
        np.testing.assert_allclose(binned.ravel(), want, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(data.metric_total(templ, kind), np.nansum(want), rtol=1e-10)
        np.testing.assert_allclose(getattr(stats, kind)(k[:160], lam[:160], sigma=s[:160]), want, rtol=1e-9, atol=1e-9)
        # plain numbers carry no errors: sigma = 0
        np.testing.assert_allclose(getattr(stats, kind)(k[:160], lam[:160]), so.metric_wide(kind, k[:160], lam[:160], np.zeros(160)),
                                   rtol=1e-9, atol=1e-9)
    neg = k[:160].copy()
    neg[3] = -1.0
    for kind in ("mcllh_mean", "mcllh_eff"):
        with pytest.raises(ValueError):
            getattr(stats, kind)(neg, lam[:160], sigma=s[:160])
    for kind in ("correct_chi2", "signed_sqrt_mod_chi2"):
        np.testing.assert_allclose(getattr(stats, kind)(neg, lam[:160], sigma=s[:160]), so.metric_wide(kind, neg, lam[:160], s[:160]),
                                   rtol=1e-12)
    with pytest.raises(ValueError):
        data.metric(templ, "barlow_llh")


def test_vectorizer_helpers():
    """pisa/utils/vectorizer.py's functions on device tensors (in place) and on numpy arrays, against numpy; the
    reference's own unit test (`test_imul_and_scale`, vectorizer.py:111-117)"""
    from pisa_amd import kernels as K
    from pisa_amd.utils import vectorizer as V

    rs = np.random.RandomState(2)
    n = 10007
    a, b, o = rs.rand(n) + 0.1, rs.randn(n), rs.randn(n)
    b[:10] = 0.0
    cases = [("scale", (a, 2.5), a * 2.5), ("mul", (a, b), a * b), ("imul", (a,), o * a), ("imul_and_scale", (a, -3.0), o * (a * -3.0)),
             ("itruediv", (b,), np.where(b == 0, 0.0, o / np.where(b == 0, 1.0, b))), ("assign", (a,), a), ("pow", (a, 1.7), a ** 1.7),
             ("sqrt", (a,), np.sqrt(a)), ("replace_where_counts_gt", (a, b, 0.3), np.where(b > 0.3, a, o))]
    for name, args, want in cases:
        host_out = o.copy()
        getattr(V, name)(*args, out=host_out)
        dev_out = K.to_device(o.copy())
        dev_args = [K.to_device(x) if isinstance(x, np.ndarray) else x for x in args]
        assert getattr(V, name)(*dev_args, out=dev_out) is dev_out
        for got in (host_out, dev_out.cpu().numpy()):
            if name == "pow":
                np.testing.assert_allclose(got, want, rtol=1e-14)
            else:
                assert np.array_equal(got, want), name
    lin = np.linspace(0, 1, 1000)
    out = np.ones_like(lin)
    V.imul_and_scale(vals=lin, scale=10.0, out=out)
    assert np.allclose(out, np.linspace(0, 10, 1000))
    two_d = np.ones((20, 3))
    V.imul(np.arange(60.0).reshape(20, 3), out=two_d)
    assert np.array_equal(two_d, np.arange(60.0).reshape(20, 3))
