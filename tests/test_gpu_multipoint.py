"""Several independent parameter points in one sweep of the events (`HotPathEngine.eval_many`:
pisa_hip_prob3_grid_planned_multi + pisa_hip_reweight_hist_multi + pisa_hip_finalize_metric_multi)
against the point-by-point path: per point the probability tables, the integer limbs, the maps and the
metric must be the same BITS -- a fit that takes its finite-difference stencil in one sweep
(pisa/analysis/analysis.py:2493-2670 with the l-bfgs-b / slsqp settings) then follows the very
trajectory of the point-by-point fit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _canonical(limbs):
    """carry-normalised digits of every accumulator: limbs [..., 6] int64 (un-normalised sums of signed
    digits) -> the unique representation with digits 0 .. 2^32 - 1 below a signed top digit.  Two limb
    sets describe the same exact sums iff these agree (the sweep kernel and the single-point kernel cut a
    weight into digits differently: truncated shifts against rounded add/subtract pairs)."""
    out = limbs.clone()
    for j in range(out.shape[-1] - 1):
        carry = out[..., j] >> 32          # arithmetic shift: floor division
        out[..., j] -= carry << 32
        out[..., j + 1] += carry
    return out


def _points(wl, n, seed=11, **kw):
    rs = np.random.RandomState(seed)
    return [wl.osc_params(theta23_deg=31.0 + 28.0 * rs.rand(), dm31=1e-3 + 6e-3 * rs.rand(),
                          deltacp_deg=360.0 * rs.rand(), **kw) for _ in range(n)]


@pytest.mark.parametrize("k", [2, 3, 5, 9])
def test_eval_many_is_bit_identical_to_point_by_point(k):
    from pisa_amd import synthetic

    wl = synthetic.Workload(n_events=12 * 20011, grid=(60, 40), out_binning="dragon", seed=5)
    st = synthetic.DeviceState(wl, compact=True)
    assert st.index16
    st.make_pseudo_data(wl.osc_params(), seed=0)
    assert st.multi_capable()
    pts = _points(wl, k)
    serial, limbs, maps = [], [], []
    for p in pts:
        st.accumulate(p)
        limbs.append(st.ws.limbs.clone())
        serial.append(st.eval_host(p, "llh"))
        maps.append((st.ws.hist.clone(), st.ws.sumw2.clone()))
    got = st.eval_many(pts, "llh")
    st.check_status()
    w = st.last_many
    assert got == serial
    for i in range(k):
        assert bool((w["hist"][i] == maps[i][0]).all()) and bool((w["sumw2"][i] == maps[i][1]).all()), i
    # the integer limbs of the sweep (the tail of eval_many leaves them zeroed: the sweep alone, through
    # the C ABI)
    import ctypes as C

    from pisa_amd import _lib
    from pisa_amd import kernels as K

    assert int(w["limbs"].abs().max()) == 0
    _lib.check(_lib.lib().pisa_hip_reweight_hist_multi(
        st._cont_arr, len(st._cont_arr), C.byref(st.grid.binning), C.c_void_p(w["tables"].data_ptr()), k, None,
        C.byref(st.out_binning), C.c_void_p(w["limbs"].data_ptr()), 1, C.c_void_p(st.ws.status.data_ptr()),
        K._stream()))
    for i in range(k):
        assert bool((_canonical(w["limbs"][i]) == _canonical(limbs[i])).all()), i
    w["limbs"].zero_()
    # the interleaved tables hold the single-point tables of every point
    for i, p in enumerate(pts):
        st.compute_probs(p)
        assert bool((w["tables"][:, :, :, i, :] == st.pepmu).all()), i
    # chi2-type metric through the same tail
    assert st.eval_many(pts, "mod_chi2") == [st.eval_host(p, "mod_chi2") for p in pts]
    # and the single-point state of the engine is untouched by a sweep
    a = st.eval_host(pts[0], "llh")
    st.eval_many(pts, "llh")
    assert st.eval_host(pts[0], "llh") == a == serial[0]


def test_eval_many_per_point_scales_ragged_sizes_and_large_batches():
    """per-point aeff scales (a free aeff_scale / nutau_norm in the stencil), an event count that is no
    multiple of the 256-event blocks, a 10 x 10 x 2 binning (three points per sweep fit the LDS budget
    of 64 KiB ... six that of 128 KiB), and a batch beyond PISA_HIP_MAX_POINTS"""
    from pisa_amd import _lib, synthetic

    wl = synthetic.Workload(n_events=12 * 1001, grid=(30, 20), out_binning="example3d", seed=6)
    st = synthetic.DeviceState(wl, compact=True)
    st.make_pseudo_data(wl.osc_params(), seed=0)
    pts = _points(wl, 7)
    base = np.array([c.scale for c in st.cont])
    rs = np.random.RandomState(3)
    scales = base[None, :] * (0.8 + 0.4 * rs.rand(len(pts), len(base)))
    got = st.eval_many(pts, "llh", scales=scales)
    want = []
    for p, sc in zip(pts, scales):
        for name, v in zip(st.names, sc):
            st.set_scale(name, v)
        want.append(st.eval_host(p, "llh"))
    assert got == want
    for name, v in zip(st.names, base):
        st.set_scale(name, v)
    many = _points(wl, _lib.MAX_POINTS + 3, seed=2)
    assert st.eval_many(many, "llh") == [st.eval_host(p, "llh") for p in many]
    st.check_status()


def test_eval_many_falls_back_where_the_sweep_does_not_apply():
    """event-mode oscillation, 40 B columns, binnings beyond the LDS accumulators: same numbers through
    the point-by-point path"""
    from pisa_amd import synthetic

    wl = synthetic.Workload(n_events=12 * 3000, grid=(30, 20), out_binning="fine3d", seed=7)
    st = synthetic.DeviceState(wl, compact=True)
    st.make_pseudo_data(wl.osc_params(), seed=0)
    assert not st.multi_capable()
    pts = _points(wl, 3)
    assert st.eval_many(pts, "llh") == [st.eval_host(p, "llh") for p in pts]
    wl2 = synthetic.Workload(n_events=12 * 3000, grid=(30, 20), out_binning="dragon", seed=7)
    st2 = synthetic.DeviceState(wl2, compact=False)
    st2.make_pseudo_data(wl2.osc_params(), seed=0)
    assert not st2.multi_capable()
    pts = _points(wl2, 3)
    assert st2.eval_many(pts, "llh") == [st2.eval_host(p, "llh") for p in pts]


def test_eval_many_decay_points_and_mixed_batches_are_refused():
    from pisa_amd import _lib, synthetic

    wl = synthetic.Workload(n_events=12 * 2000, grid=(30, 20), out_binning="dragon", seed=8)
    st = synthetic.DeviceState(wl, compact=True)
    st.make_pseudo_data(wl.osc_params(), seed=0)
    dec = _points(wl, 3, decay_alpha3=1e-4)
    assert st.eval_many(dec, "llh") == [st.eval_host(p, "llh") for p in dec]
    mixed = [dec[0], wl.osc_params()]
    with pytest.raises(_lib.PisaHipError):
        st.eval_many(mixed, "llh")


def test_eval_many_at_the_headline_size():
    """1e7 events, 200 x 100 grid: five points in one sweep = five evaluations, bit for bit"""
    from pisa_amd import synthetic

    wl = synthetic.Workload(n_events=10_000_000, grid=(200, 100), out_binning="dragon", seed=0)
    st = synthetic.DeviceState(wl, compact=True)
    st.make_pseudo_data(wl.osc_params(), seed=0)
    pts = _points(wl, 5)
    assert st.eval_many(pts, "llh") == [st.eval_host(p, "llh") for p in pts]
    st.check_status()
