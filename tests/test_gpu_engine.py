"""Engine-level behaviour on the GPU: stream-overlapped batch evaluation."""
import numpy as np
import pytest
import torch

from tests.conftest import bench_result

pytestmark = pytest.mark.gpu


def test_eval_batch_equals_sequential():
    """osc kernels of point k+1 on a second HIP stream overlap the fused kernel of
    point k (double-buffered tables): same bits as point-by-point evaluation"""
    from pisa_amd import synthetic

    wl = synthetic.Workload(n_events=240000, grid=(60, 40), out_binning="dragon", seed=2)
    st = synthetic.DeviceState(wl)
    st.make_pseudo_data(wl.osc_params(), seed=0)
    rs = np.random.RandomState(1)
    plist = [wl.osc_params(theta23_deg=35 + 20 * rs.rand(), dm31=2e-3 + 1e-3 * rs.rand()) for _ in range(7)]
    seq = [float(st.eval(p, "llh").item()) for p in plist]
    for _ in range(3):  # repeated to exercise buffer reuse / stream ordering
        got = st.eval_batch(plist, "llh").cpu().numpy()
        assert list(got) == seq
    st.check_status()
    # state left behind is that of the last point
    last = float(st.metric("llh").item())
    assert last == seq[-1]


@pytest.mark.parametrize("n_events", [240000, 1203])
def test_bin_run_order_is_bit_identical(n_events):
    """events stored in (output bin, node) order go through the register-accumulating
    kernel (PISA_HIP_CONT_BIN_RUNS); the exact accumulation makes the maps
    independent of event order and of the kernel variant: same bits"""
    import torch

    from pisa_amd import synthetic

    wl = synthetic.Workload(n_events=n_events, grid=(60, 40), out_binning="dragon", seed=5)
    p = wl.osc_params(theta23_deg=47.0)
    maps = []
    for order, lds in (("node", False), ("node", True), ("bin", False), (False, False)):
        st = synthetic.DeviceState(wl, sort_events=order, lds_order=lds)
        st.make_pseudo_data(wl.osc_params(), seed=0)
        llh = float(st.eval(p, "llh").item())
        st.check_status()
        h, s2 = st.finalize()
        maps.append((h.cpu().numpy().copy(), s2.cpu().numpy().copy(), llh))
    for h, s2, llh in maps[1:]:
        assert np.array_equal(h, maps[0][0])
        assert np.array_equal(s2, maps[0][1])
        assert llh == maps[0][2]
    assert maps[0][0].sum() > 0


def test_large_binning_matches_oracle_in_any_order():
    """4800 output bins do not fit LDS accumulators: the kernel keeps a window of the binning in
    LDS and the events are stored by bin partition (default, "part": one LDS window per
    partition, node-sorted inside) or by bin ("bin"); node order and the unsorted sample fall
    back to per-event global atomics.  Same bits in every order, and the oracle's maps within
    1e-10 relative."""
    from oracle import pipeline_oracle
    from pisa_amd import synthetic

    wl = synthetic.Workload(n_events=120000, grid=(40, 30), out_binning="fine3d", seed=3)
    p = wl.osc_params(theta23_deg=44.0)
    res = []
    for order in (True, "part", "bin", "node", False):
        st = synthetic.DeviceState(wl, sort_events=order)
        st.accumulate(p)
        st.check_status()
        h, s2 = st.finalize()
        res.append((h.cpu().numpy().copy(), s2.cpu().numpy().copy()))
    for h, s2 in res[1:]:
        assert np.array_equal(h, res[0][0]) and np.array_equal(s2, res[0][1])
    ref = pipeline_oracle.oracle_eval(wl, wl.last_matrices)
    ref_h = np.asarray(ref["hist"]).reshape(len(wl.events), -1)
    ref_s2 = np.asarray(ref["sumw2"]).reshape(len(wl.events), -1)
    scale = np.abs(ref_h).max()
    assert np.allclose(res[0][0], ref_h, rtol=1e-10, atol=1e-13 * scale)
    assert np.allclose(res[0][1], ref_s2, rtol=1e-10, atol=1e-13 * np.abs(ref_s2).max())


def test_dropping_unbinned_events_changes_nothing():
    """events outside the output binning (static reco coordinates) or outside the calc grid
    never contribute; an engine that does not keep them resident gives the same bits"""
    from pisa_amd import synthetic

    wl = synthetic.Workload(n_events=240000, grid=(60, 40), out_binning="dragon", seed=8)
    p = wl.osc_params(theta23_deg=44.0)
    full = synthetic.DeviceState(wl)
    lean = synthetic.DeviceState(wl, drop_unbinned=True)
    assert 0 < lean.n_local < full.n_local
    for st in (full, lean):
        st.make_pseudo_data(wl.osc_params(), seed=0)
    a, b = full.eval_host(p), lean.eval_host(p)
    assert a == b
    assert bool((full.ws.hist == lean.ws.hist).all()) and bool((full.ws.sumw2 == lean.ws.sumw2).all())
    full.check_status()
    lean.check_status()


def test_compact_layout_matches_exact_association(oracle):
    """24 B/event form: initial_weights*weighted_aeff folded into the flux pair once.
    Same product with the static factors associated first: equal to the 40 B form to a few ulp
    per weight, to the oracle within the parity bar; flux updates refresh the folded column."""
    from oracle import pipeline_oracle
    from pisa_amd import kernels as K
    from pisa_amd import synthetic

    wl = synthetic.Workload(n_events=240000, grid=(60, 40), out_binning="dragon", seed=9)
    p = wl.osc_params(theta23_deg=48.0)
    mats = dict(wl.last_matrices)
    exact = synthetic.DeviceState(wl)
    comp = synthetic.DeviceState(wl, compact=True)
    for st in (exact, comp):
        st.make_pseudo_data(wl.osc_params(), seed=0)
    np.testing.assert_array_equal(exact.data.cpu().numpy(), comp.data.cpu().numpy())
    a, b = exact.eval_host(p), comp.eval_host(p)
    np.testing.assert_allclose(b, a, rtol=1e-12)
    he, se = exact.maps()
    hc, sc = comp.maps()
    np.testing.assert_allclose(hc, he, rtol=1e-14, atol=0)
    np.testing.assert_allclose(sc, se, rtol=1e-14, atol=0)
    ref = pipeline_oracle.oracle_eval(wl, mats)
    np.testing.assert_allclose(hc, np.asarray(ref["hist"]).reshape(hc.shape), rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(sc, np.asarray(ref["sumw2"]).reshape(sc.shape), rtol=1e-10, atol=1e-300)
    # run-to-run and order independence hold for the compact form as well
    comp2 = synthetic.DeviceState(wl, compact=True, sort_events=False)
    comp2.set_data(comp.data.cpu().numpy())
    assert comp2.eval_host(p) == b
    # a flux systematic changes nu_flux of one container: both engines follow
    i = 1
    new_flux = K.to_device(wl.events[i]["nu_flux"] * np.array([1.07, 0.96]))
    for st in (exact, comp):
        st.update_flux(i, new_flux)
    a2, b2 = exact.eval_host(p), comp.eval_host(p)
    assert a2 != a
    np.testing.assert_allclose(b2, a2, rtol=1e-12)
    np.testing.assert_allclose(comp.maps()[0], exact.maps()[0], rtol=1e-14, atol=0)
    exact.check_status()
    comp.check_status()


@pytest.mark.parametrize("n_events", [240000, 12 * 1001, 12 * 3])
def test_index16_form_is_bit_identical_to_compact(n_events):
    """20 B/event form (both indices in 16 bits, quad-blocked flux pairs, padded columns): the
    same weights in the same arithmetic as the 24 B compact form, so maps and LLH are identical
    bit for bit -- for event counts that are not multiples of the 256-event blocks, with events
    outside the binning, after a flux update, without the events that never land in a bin, and
    through the plain (non-lean) entry points."""
    from pisa_amd import kernels as K
    from pisa_amd import synthetic

    wl = synthetic.Workload(n_events=n_events, grid=(60, 40), out_binning="dragon", seed=21)
    p = wl.osc_params(theta23_deg=44.0)
    wide = synthetic.DeviceState(wl, compact=True, index16=False)
    narrow = synthetic.DeviceState(wl, compact=True)
    assert narrow.index16 and not wide.index16
    assert narrow.cont[0].d_weighted_flux_q and not narrow.cont[0].d_weighted_flux
    wide.make_pseudo_data(wl.osc_params(), seed=0)
    narrow.set_data(wide.data.cpu().numpy())
    assert narrow.eval_host(p) == wide.eval_host(p)
    for a, b in zip(narrow.maps(), wide.maps()):
        assert np.array_equal(a, b)
    assert wide.maps()[0].sum() > 0
    i = 4
    new_flux = K.to_device(wl.events[i]["nu_flux"] * np.array([0.9, 1.2]))
    before = wide.eval_host(p)
    for st in (wide, narrow):
        st.update_flux(i, new_flux)
    assert narrow.eval_host(p) == wide.eval_host(p) != before
    # generic entry points (accumulate / finalize) and the event set without unbinned events
    dropped = synthetic.DeviceState(wl, compact=True, drop_unbinned=True)
    assert dropped.index16
    dropped.update_flux(i, new_flux)
    dropped.accumulate(p)
    wide.accumulate(p)
    for a, b in zip(dropped.finalize(), wide.finalize()):
        assert torch_equal(a, b)
    narrow.check_status()
    dropped.check_status()


def test_node_flux_tables_match_per_event_flux():
    """Flux living on the oscillation grid: the engine forms flux x probability per node
    (`pisa_hip_flux_prob_tables`) and the events carry only w0*aeff.  Same maps and LLH as the
    per-event form fed with the flux looked up at every event's node (different association of
    the same three factors: a few ulp), through every entry point, and after a flux update."""
    import torch

    from pisa_amd import kernels as K
    from pisa_amd import synthetic

    wl = synthetic.Workload(n_events=60011, grid=(60, 40), out_binning="dragon", seed=33)
    g = wl.grid
    ee, cc = np.meshgrid(g.energy, g.coszen, indexing="ij")
    rs = np.random.RandomState(2)
    node_flux = []
    for ev in wl.events:
        f_mu = 1e4 * ee ** -2.7 * (1 + 0.5 * cc ** 2) * (0.8 + 0.4 * rs.rand(*ee.shape))
        fn = np.stack([f_mu * (0.5 - 0.2 * cc), f_mu], axis=-1).reshape(-1, 2)
        gx, gy = K.to_device(np.log(ev["true_energy"])), K.to_device(ev["true_coszen"])
        node = K.event_indices([gx, gy], g.binning).cpu().numpy()
        assert node.min() >= 0
        ev["nu_flux"] = fn[node]
        ev["nu_flux_nodes"] = fn
        node_flux.append(fn)
    p = wl.osc_params(theta23_deg=47.0, deltacp_deg=120.0)
    per_event = synthetic.DeviceState(wl, compact=True)
    on_nodes = synthetic.DeviceState(wl, compact=True, node_flux=True)
    assert on_nodes.index16 and on_nodes.cont[0].d_pepmu
    per_event.make_pseudo_data(wl.osc_params(), seed=0)
    on_nodes.set_data(per_event.data.cpu().numpy())

    def same(a, b, rtol=1e-13):
        np.testing.assert_allclose(a, b, rtol=rtol, atol=0)

    def same_llh(a, b):   # a sum of differences of the maps: conditioned worse than the maps
        same(a, b, rtol=1e-11)

    same_llh(on_nodes.eval_host(p), per_event.eval_host(p))             # lean path
    for a, b in zip(on_nodes.maps(), per_event.maps()):
        same(a, b)
    assert per_event.maps()[1].sum() > 0
    on_nodes.accumulate(p)                                              # generic entry points
    per_event.accumulate(p)
    for a, b in zip(on_nodes.finalize(), per_event.finalize()):
        same(a.cpu().numpy(), b.cpu().numpy())
    pts = [wl.osc_params(theta23_deg=t) for t in (40.0, 45.0, 50.0)]    # two-stream batch
    same_llh(on_nodes.eval_batch(pts).cpu().numpy(), per_event.eval_batch(pts).cpu().numpy())
    # a flux systematic: 640 kB per container instead of a pass over the events
    i = 3
    before = on_nodes.eval_host(p)
    scaled = node_flux[i] * np.array([0.9, 1.3])
    on_nodes.update_flux_nodes(i, K.to_device(scaled))
    gx, gy = K.to_device(np.log(wl.events[i]["true_energy"])), K.to_device(wl.events[i]["true_coszen"])
    node = K.event_indices([gx, gy], g.binning).long()
    per_event.update_flux(i, K.to_device(scaled)[node])
    after = on_nodes.eval_host(p)
    same_llh(after, per_event.eval_host(p))
    assert after != before
    # bit-identical run to run and for any event order (exact accumulation)
    assert on_nodes.eval_host(p) == after
    unsorted = synthetic.DeviceState(wl, compact=True, node_flux=True, sort_events=False)
    unsorted.set_data(per_event.data.cpu().numpy())
    unsorted.update_flux_nodes(i, K.to_device(scaled))
    assert unsorted.eval_host(p) == after
    on_nodes.check_status()


def torch_equal(a, b):
    import torch

    return torch.equal(a, b)


def test_index16_form_needs_a_fallback_where_it_does_not_apply():
    """ABI: for a binning with 65535 bins or more the 16-bit columns are ignored: the call falls
    back to the other forms of the container, and is refused when the 16-bit form is all a
    container offers."""
    import ctypes

    import torch

    from pisa_amd import _lib, synthetic
    from pisa_amd import kernels as K

    huge = _lib.make_binning([0.0, 0.0], [1.0, 1.0], [256, 256])  # 65536 bins: no 16-bit bin numbers
    small = _lib.make_binning(synthetic.DRAGON["mins"], synthetic.DRAGON["maxs"], synthetic.DRAGON["nbins"])
    # engine on the small binning (16-bit form built), then asked for the huge one: the bin
    # numbers of its index columns belong to the small binning, so only the status matters here
    wl_small = synthetic.Workload(n_events=12 * 500, grid=(20, 10), out_binning="dragon", seed=3)
    st = synthetic.DeviceState(wl_small, compact=True)
    assert st.index16
    st.compute_probs(wl_small.osc_params())
    ws = K.HistWorkspace(len(st.cont), 65536, st.dev)
    K.reweight_hist(st._cont_arr, st.grid.binning, st.prob_nu, st.prob_nubar, st.pepmu,
                    huge, ws)  # falls back to the packed 40 B columns
    only16 = []
    for c in st.cont:
        d = _lib.Container()
        ctypes.memmove(ctypes.byref(d), ctypes.byref(c), ctypes.sizeof(d))
        d.d_node = d.d_bin = d.d_node_bin = d.d_aeff_w0 = None
        d.d_grid_x = d.d_grid_y = d.d_nu_flux = d.d_weighted_aeff = d.d_initial_weights = None
        for k in range(3):
            d.d_sample[k] = None
        only16.append(d)
    ws_small = K.HistWorkspace(len(st.cont), wl_small.n_bins, st.dev)
    K.reweight_hist(only16, st.grid.binning, st.prob_nu, st.prob_nubar, st.pepmu, small, ws_small)
    st.accumulate(wl_small.osc_params())
    K.hist_finalize(ws_small)
    assert torch.equal(ws_small.hist, st.finalize()[0])  # the 16-bit form stands alone
    with pytest.raises(_lib.PisaHipError):
        K.reweight_hist(only16, st.grid.binning, st.prob_nu, st.prob_nubar, st.pepmu, huge, ws)
    torch.cuda.synchronize()


def test_large_binning_compact_and_dropped():
    """the LDS-window path (4800 bins) with the compact columns and without the events that
    can never land in a bin: same maps as the reference-order columns to a few ulp"""
    from pisa_amd import synthetic

    wl = synthetic.Workload(n_events=120000, grid=(40, 30), out_binning="fine3d", seed=4)
    p = wl.osc_params(theta23_deg=41.0)
    ref = synthetic.DeviceState(wl)
    ref.accumulate(p)
    h0, s0 = (t.cpu().numpy().copy() for t in ref.finalize())
    for kw in (dict(compact=True), dict(compact=True, drop_unbinned=True), dict(drop_unbinned=True)):
        st = synthetic.DeviceState(wl, **kw)
        st.accumulate(p)
        st.check_status()
        h, s = (t.cpu().numpy() for t in st.finalize())
        if kw.get("compact"):
            np.testing.assert_allclose(h, h0, rtol=1e-14, atol=0, err_msg=str(kw))
            np.testing.assert_allclose(s, s0, rtol=1e-14, atol=0, err_msg=str(kw))
        else:
            assert np.array_equal(h, h0) and np.array_equal(s, s0)
    assert h0.sum() > 0
    # the host-polled evaluation works for binnings beyond the one-launch tail as well
    # (separate finalize and metric kernels, metric written to pinned host memory)
    ref.make_pseudo_data(wl.osc_params(), seed=0)
    assert ref.eval_host(p, "llh") == float(ref.eval(p, "llh").item())


def test_rccl_limb_allreduce_single_rank(tmp_path, monkeypatch):
    """The N > 1 code path on the one GPU a test box has: a 1-rank RCCL group, the engine told
    it is one of two ranks so that every evaluation goes through the int64 limb all-reduce on
    the device (identity on a 1-rank group).  Checks that RCCL accepts the int64 SUM, that the
    collective is ordered between the raw-stream launches of the fused kernel and of the tail,
    and that the polled host result is the one of the plain path, bit for bit."""
    import time

    import torch
    import torch.distributed as dist

    from pisa_amd import synthetic

    wl = synthetic.Workload(n_events=240000, grid=(60, 30))
    st = synthetic.DeviceState(wl, compact=True)
    st.make_pseudo_data(wl.osc_params(), seed=0)
    points = [wl.osc_params(theta23_deg=t) for t in (38.0, 45.0, 51.0)]
    ref = [st.eval_host(p, "llh") for p in points]
    dist.init_process_group("nccl", init_method="file://%s" % (tmp_path / "store"), rank=0,
                            world_size=1, device_id=torch.device("cuda", 0))
    try:
        def timed():
            t0 = time.perf_counter()
            for _ in range(50):
                st.eval_host(points[0], "llh")
            return (time.perf_counter() - t0) / 50 * 1e6

        t_plain = timed()
        st.world_size = 2
        got = [st.eval_host(p, "llh") for p in points]
        assert got == ref
        assert st._rccl, "direct RCCL communicator not created"
        t_direct = timed()
        # the same through torch.distributed (the fall-back path)
        st.close()
        monkeypatch.setenv("PISA_HIP_DIRECT_RCCL", "0")
        got = [st.eval_host(p, "llh") for p in points]
        assert got == ref
        assert st._rccl is False
        t_torch = timed()
        print("eval: plain %.1f us, 1-rank RCCL direct %.1f us, through torch.distributed %.1f us"
              % (t_plain, t_direct, t_torch))
        st.check_status()
        # the all-gather of the point groups on an RCCL group (device tensors; one rank here: the only RCCL a one-GPU
        # box has), and a dealt `eval_many` end to end: one group of one rank evaluates every point itself
        from pisa_amd.engine import PointGroups

        pg = PointGroups(0, 1, 1)
        assert pg.gather([1.5, -2.25, 3.0], 3, st.dev) == [1.5, -2.25, 3.0]
        st.world_size = 1
        st.points = pg
        many = st.eval_many(points, "llh")
        assert many == ref
        assert pg.block(len(points)) == (0, 3) and pg.gather(many, 3, st.dev) == ref      # the values through RCCL, bit for bit
        st.points = None
    finally:
        dist.destroy_process_group()


def test_bench_multi_rank_code_path_on_one_rank(tmp_path):
    """bench.py's N > 1 flow (RCCL process group from the launcher's environment, direct limb
    all-reduce inside every evaluation, barriers, max over ranks, communicator teardown) run
    end to end with the one rank a single-GPU box has."""
    import json
    import os
    import subprocess
    import sys

    import socket

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:  # a free port for the rendezvous
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
           "--gpus", "1", "--force-dist", "--events", "1.2e6", "--steps", "40", "--warmup", "5",
           "--no-cpu-baseline", "--no-drop-probe", "--legs", "multi_point,fit_c4_engine,l3_exceeding",
           "--detail-out", str(tmp_path / "detail.json")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    compact, line = bench_result(out.stdout, str(tmp_path / "detail.json"))
    assert compact["legs_run"] == ["fit_c4_engine", "multi_point"] and compact["nccl_comm_count"] == 1
    assert compact["weak_value"] == pytest.approx(line["weak_value"], rel=1e-5)
    # N > 1 headline = strong scaling of ONE sample (north star); the weak-scaling rate beside it
    assert line["n_gpus"] == 1 and line["scaling"] == "strong" and line["value"] > 0
    assert line["value"] == line["strong_value"] and line["weak_value"] > 0
    assert line["weak"]["samples_per_step"] == 1 and line["allreduce_ms"] > 0
    assert line["nccl_comm_count"] == 1      # ncclCommCount of the direct communicator
    # the legs that run on several ranks do, through the all-reduce of K limb sets (the others are N = 1 only)
    assert set(line["legs"]) == {"multi_point", "fit_c4_engine"} and np.isfinite(line["last_llh"])
    assert all(line["legs"]["multi_point"]["K%d" % k]["same_bits_as_point_by_point"] for k in (3, 5, 9))
    assert line["legs"]["fit_c4_engine"]["same_history"]


def _two_rank_worker(rank, world, port, out_dir):
    import os

    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    from pisa_amd import synthetic

    wl = synthetic.Workload(n_events=240000, grid=(60, 30))
    st = synthetic.DeviceState(wl, rank=rank, world_size=world, compact=True)
    st.make_pseudo_data(wl.osc_params(), seed=0)
    vals = [st.eval_host(wl.osc_params(theta23_deg=t), "llh") for t in (38.0, 45.0, 51.0)]
    assert st._rccl and st._rccl.count() == world
    h, s2 = st.maps()
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), np.concatenate([vals, h.ravel(), s2.ravel()]))
    st.close()
    dist.destroy_process_group()


@pytest.mark.skipif(__import__("torch").cuda.device_count() < 2, reason="needs two GPUs")
def test_rccl_limb_allreduce_two_ranks(tmp_path):
    """two processes, two GPUs: events sharded, limbs all-reduced over the direct RCCL communicator;
    both ranks end with the single-GPU maps and LLH, bit for bit (skipped on 1-GPU boxes)"""
    import socket

    import torch.multiprocessing as mp

    from pisa_amd import synthetic

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    wl = synthetic.Workload(n_events=240000, grid=(60, 30))
    st = synthetic.DeviceState(wl, compact=True)
    st.make_pseudo_data(wl.osc_params(), seed=0)
    vals = [st.eval_host(wl.osc_params(theta23_deg=t), "llh") for t in (38.0, 45.0, 51.0)]
    h, s2 = st.maps()
    want = np.concatenate([vals, h.ravel(), s2.ravel()])
    for r in range(2):
        np.testing.assert_array_equal(np.load(str(tmp_path / ("rank%d.npy" % r))), want)


def test_scaled_tail_kernel_matches_host_scaling(oracle):
    """`pisa_hip_finalize_metric_scaled`: per-(container, bin) factors of a stage behind the histogram
    (hypersurfaces.py:251-259: weights = clip(weights s, 0, inf), errors *= s) and an added map enter
    the metric inside the tail kernel.  Same value as the oracle metric of the host-scaled maps; the
    maps themselves are written unscaled; without factors and addend it is the plain tail, bit for bit."""
    import torch

    from pisa_amd import kernels as K
    from pisa_amd import synthetic

    wl = synthetic.Workload(n_events=60000, grid=(40, 30), out_binning="dragon", seed=4)
    st = synthetic.DeviceState(wl, compact=True)
    p = wl.osc_params(theta23_deg=44.0)
    st.make_pseudo_data(wl.osc_params(), seed=1)
    data = st.data.cpu().numpy()
    rs = np.random.RandomState(3)
    n_cont, n_bins = len(st.cont), st.n_bins
    scale = 1.0 + 0.3 * rs.randn(n_cont, n_bins)
    scale[0, :5] = -0.2                                  # negative factors: the weights clip at zero
    extra = np.stack([rs.rand(n_bins) * 5.0, rs.rand(n_bins) * 2.0])
    scale_d, extra_d = K.to_device(scale), K.to_device(extra)
    st.accumulate(p)
    hist, sumw2 = (t.cpu().numpy() for t in st.finalize())
    lam = np.clip(hist * scale, 0, np.inf).sum(axis=0) + extra[0]
    var = ((np.sqrt(sumw2) * scale) ** 2).sum(axis=0) + extra[1]
    for kind in ("llh", "poisson_llh", "chi2", "mod_chi2"):
        st.accumulate(p)
        got = st.tail_host(kind, scale_d, extra_d)
        _, want = oracle.metric(kind, data, lam, var)
        np.testing.assert_allclose(got, want, rtol=1e-12, err_msg=kind)
        h2, s2 = (t.cpu().numpy() for t in (st.ws.hist, st.ws.sumw2))
        assert np.array_equal(h2, hist) and np.array_equal(s2, sumw2)      # maps as histogrammed
        st.accumulate(p)
        plain = st.tail_host(kind)
        st.accumulate(p)
        assert st.tail_host(kind, None, None) == plain
        st.accumulate(p)
        only_extra = st.tail_host(kind, None, extra_d)
        _, want2 = oracle.metric(kind, data, hist.sum(axis=0) + extra[0], sumw2.sum(axis=0) + extra[1])
        np.testing.assert_allclose(only_extra, want2, rtol=1e-12, err_msg=kind)
    st.check_status()


@pytest.mark.gpu
@pytest.mark.parametrize("out_dims", [3, 1])
def test_coordinate_form_paths_are_bit_identical(out_dims):
    """the SURVEY 8(d)-shaped kernel (coordinates digitised in the kernel) has a pair path (16-byte
    loads, all columns 16-byte aligned), a scalar path (any 8-byte aligned pointers: what a foreign
    caller of the C-ABI may hand over) and two gather sources (compact tables when given, else the
    full matrices): all of them, and the pre-digitised form, give the same limbs bit for bit"""
    import ctypes

    import torch

    from pisa_amd import _lib, kernels as K, synthetic

    wl = synthetic.Workload(n_events=12 * 4001, grid=(30, 20), out_binning="dragon", seed=11)   # odd containers
    st = synthetic.DeviceState(wl, indexed=False)
    p = wl.osc_params(theta23_deg=47.0)
    st.compute_probs(p)
    D = synthetic.DRAGON
    out_b = st.out_binning if out_dims == 3 else _lib.make_binning(D["mins"][:1], D["maxs"][:1], D["nbins"][:1])
    n_bins = int(np.prod(D["nbins"][:out_dims]))

    def run(cs, pepmu):
        ws = K.HistWorkspace(len(cs), n_bins, st.dev)
        K.reweight_hist(cs, st.grid.binning, st.prob_nu, st.prob_nubar, pepmu, out_b, ws)
        assert int(ws.status.item()) == 0
        return ws.limbs.clone()

    pairs = run(st._cont_arr, st.pepmu)
    assert int(pairs.abs().sum().item()) > 0
    if out_dims == 3:
        ref = synthetic.DeviceState(wl, indexed=True, compact=False)
        ref.accumulate(p)
        assert torch.equal(pairs, ref.ws.limbs)
    assert torch.equal(run(st._cont_arr, None), pairs)   # gathers from the full matrices
    # the same columns moved by one element: 8-byte aligned only -> scalar path
    keep, shifted = [], []
    per = 5 + 3
    for ci, c in enumerate(st.cont):
        d = _lib.Container()
        ctypes.memmove(ctypes.byref(d), ctypes.byref(c), ctypes.sizeof(d))
        cols = st._keep[per * ci:per * (ci + 1)]
        assert cols[0].data_ptr() == c.d_grid_x and cols[4].data_ptr() == c.d_initial_weights \
            and cols[7].data_ptr() == c.d_sample[2]
        moved = []
        for t in cols:
            buf = torch.empty(t.numel() + 1, dtype=torch.float64, device=st.dev)
            buf[1:] = t.reshape(-1)
            assert buf[1:].data_ptr() % 16 == 8
            keep.append(buf)
            moved.append(buf[1:].data_ptr())
        d.d_grid_x, d.d_grid_y, d.d_nu_flux, d.d_weighted_aeff, d.d_initial_weights = moved[:5]
        for k in range(3):
            d.d_sample[k] = moved[5 + k]
        shifted.append(d)
    assert torch.equal(run(shifted, st.pepmu), pairs)
    assert torch.equal(run(shifted, None), pairs)
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("index16", [True, False])
def test_flux_refresh_of_several_containers_in_one_launch(index16):
    """`update_flux_many` (pisa_hip_fold_flux_multi) folds the rewritten flux columns of several
    containers exactly like `update_flux` container by container, for a subset as well, and follows
    arrays that are rewritten in place (the argument block is reused)"""
    import torch

    from pisa_amd import kernels as K, synthetic

    wl = synthetic.Workload(n_events=12 * 3001, grid=(30, 20), out_binning="dragon", seed=4)
    a = synthetic.DeviceState(wl, compact=True, index16=index16)
    b = synthetic.DeviceState(wl, compact=True, index16=index16)
    assert a.index16 == index16
    p = wl.osc_params(theta23_deg=43.0)
    rs = np.random.RandomState(3)
    fluxes = [K.to_device(ev["nu_flux"] * (0.5 + rs.rand(*ev["nu_flux"].shape))) for ev in wl.events]
    for i, f in enumerate(fluxes):
        a.update_flux(i, f)
    b.update_flux_many(list(enumerate(fluxes)))
    a.accumulate(p)
    b.accumulate(p)
    assert torch.equal(a.ws.limbs, b.ws.limbs)
    # in place, same tensors: only a subset announced
    for f in fluxes[:5]:
        f.mul_(1.25)
    for i in range(5):
        a.update_flux(i, fluxes[i])
    b.update_flux_many([(i, fluxes[i]) for i in range(5)])
    a.accumulate(p)
    b.accumulate(p)
    assert torch.equal(a.ws.limbs, b.ws.limbs)
    assert int(a.ws.limbs.abs().sum().item()) > 0
    b.update_flux_many([])      # nothing moved: no launch


@pytest.mark.gpu
def test_bench_line_contract_single_gpu(tmp_path):
    """`python bench.py` prints ONE JSON line -- compact, strict JSON, below 4 KB (conftest.bench_result) -- with the
    fields the driver and the judge read: the metric of BASELINE.json, whole-job value, the roofline of the dominant
    kernel (measured with HIP events in this run; `frac` = achieved / peak) and the CPU baseline of the oracle port;
    the legs in full, the thread scan and the referee's report go to the detail file"""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--events", "1.2e6", "--steps", "30", "--warmup", "5",
           "--legs", "none", "--no-drop-probe", "--no-batch-probe", "--detail-out", str(tmp_path / "detail.json")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    d, detail = bench_result(out.stdout, str(tmp_path / "detail.json"))
    assert "bench_detail {" in out.stderr
    base = json.load(open(os.path.join(root, "BASELINE.json")))
    assert d["metric"].split(",")[0] in base["metric"]
    assert d["unit"] == "evals/s" and d["n_gpus"] == 1 and d["steps"] == 30 and d["warmup"] == 5
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f64"
    assert d["data"].startswith("synthetic") and "workload" in d["config"]
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 1.0) < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.0 < r["frac"] < 1.0
    assert abs(r["achieved"] - r["bytes_per_event"] * r["events_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert "traffic" in r and "traffic_source" in detail["roofline"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["sample"] and c["cpu_model"]
    assert c["single_thread"] > 0 and detail["cpu_baseline"]["single_thread"]["cores"] == 1
    assert c["value"] < d["value"]
    assert abs(c["llh_rel_diff"]) < 1e-9 and d["llh_gate"]["referee_met"] is True
    assert "thread_scan" in detail["cpu_baseline"] and "thread_scan" not in c


@pytest.mark.parametrize("index16", [True, False])
def test_one_pass_flux_refresh_is_the_two_pass_refresh(index16):
    """A `flux.barr_simple` systematic with the flux per event: `update_flux_barr` (resident-order
    nominal fluxes + parameter-free factors -> the folded column, one elementwise pass) leaves the very
    bits of `pisa_hip_barr_simple_multi` in container order followed by `update_flux_many` (gather into
    the resident order + fold) -- in the quad-blocked 20 B layout and the plain 24 B one, ragged event
    counts, padding untouched by garbage, and the evaluation that follows agrees as well"""
    import torch

    from pisa_amd import kernels as K
    from pisa_amd import synthetic

    wl = synthetic.Workload(n_events=12 * 10007, grid=(40, 30), out_binning="dragon", seed=12)
    a = synthetic.DeviceState(wl, compact=True, index16=index16)
    b = synthetic.DeviceState(wl, compact=True, index16=index16)
    assert a.index16 == index16
    a.make_pseudo_data(wl.osc_params(), seed=0)
    b.set_data(a.data.cpu().numpy())
    cols = []
    for ev in wl.events:
        cols.append((K.to_device(ev["true_energy"]), K.to_device(ev["true_coszen"]), K.to_device(ev["nu_flux"]),
                     K.to_device(ev["nu_flux"] * 0.7)))
    b.enable_barr(cols)
    outs = [torch.empty((c[0].numel(), 2), dtype=torch.float64, device="cuda") for c in cols]
    sets = K.barr_sets([(e, cz, nu, nub, ev["nubar"], out) for (e, cz, nu, nub), ev, out in zip(cols, wl.events, outs)])
    p = wl.osc_params(theta23_deg=46.0)
    for ps in ((1.03, 0.97, 0.04, 0.3, -0.2), (0.95, 1.08, -0.06, -0.7, 0.5), (1.0, 1.0, 0.0, 0.0, 0.0)):
        K.barr_simple_multi(sets, *ps)
        a.update_flux_many(list(enumerate(outs)))
        b.update_flux_barr(*ps)
        for wa, wb in zip(a._wflux, b._wflux):
            assert torch.equal(wa, wb)
        assert a.eval_host(p, "llh") == b.eval_host(p, "llh")
    a.check_status()
    b.check_status()
    # an energy that is not positive has no place in the one-pass form (the two-pass calls keep the
    # reference's answers there)
    bad = list(cols)
    e0 = cols[0][0].clone()
    e0[3] = -1.0
    bad[0] = (e0,) + cols[0][1:]
    with pytest.raises(ValueError):
        b.enable_barr(bad)


@pytest.mark.parametrize("n_events", [13, 4099, 250_007])
def test_accumulate_launch_shape_does_not_change_a_bit(n_events):
    """The accumulate kernels' workgroups are dealt to the containers by load (one per CU by default,
    `plan_blocks_balanced`); the integer accumulation makes maps and metric independent of the number of workgroups
    and of how they are dealt: a single workgroup, fewer workgroups than containers, the default, many more than CUs
    and other workgroup sizes give the same bits -- for the 16-bit index form, the compact form, the reference-order
    form and the coordinate form.  The launch shape can be chosen in the development build of the library only
    (tests/dev_cases.py `launch_shape` on libpisa_hip_dev.so)."""
    from tests.conftest import run_dev_case

    run_dev_case("launch_shape", n_events)


@pytest.mark.parametrize("kw", [dict(compact=True), dict(compact=False), dict(compact=True, index16=False)],
                         ids=["20B", "40B", "24B"])
def test_one_call_evaluation_equals_the_separate_calls(kw):
    """`pisa_hip_evaluator_eval` (prob3 -> accumulate -> tail enqueued and awaited inside ONE C-ABI call) against
    the three separate calls of the same entry points: the same launches, so the same bits -- llh / mod_chi2 through
    the four-workgroup tail, chi2 through the one-workgroup tail, after a change of a container's scale
    (aeff.py:78-86), after new pseudo-data, and with the maps read back in between (limbs not zero)."""
    from pisa_amd import synthetic

    wl = synthetic.Workload(n_events=120_000, grid=(40, 30), out_binning="dragon", seed=4)
    a = synthetic.DeviceState(wl, **kw)
    b = synthetic.DeviceState(wl, **kw)
    b.one_call = False
    data = a.make_pseudo_data(wl.osc_params(), seed=0)
    b.set_data(data)
    rs = np.random.RandomState(2)
    pts = [wl.osc_params(theta23_deg=38 + 14 * rs.rand(), dm31=2.2e-3 + 6e-4 * rs.rand()) for _ in range(6)]
    for kind in ("llh", "mod_chi2", "chi2"):
        for p in pts[:3]:
            va, vb = a.eval_host(p, kind), b.eval_host(p, kind)
            assert va == vb and np.isfinite(va), (kind, va, vb)
    assert a._evaluator is not None and b._evaluator is None
    ha, hb = a.maps(), b.maps()
    assert np.array_equal(ha[0], hb[0]) and np.array_equal(ha[1], hb[1]) and ha[0].sum() > 0
    a.set_scale("numu_cc", 0.7 * a.cont[1].scale)
    b.set_scale("numu_cc", 0.7 * b.cont[1].scale)
    assert a.eval_host(pts[3], "llh") == b.eval_host(pts[3], "llh")
    data2 = np.random.RandomState(5).poisson(data + 3.0).astype(np.float64)
    a.set_data(data2)
    b.set_data(data2)
    assert a.eval_host(pts[4], "llh") == b.eval_host(pts[4], "llh")
    # a plain accumulate in between leaves limbs that are not zero: the next one-call evaluation clears them
    a.accumulate(pts[5])
    a.finalize()
    assert a.eval_host(pts[4], "llh") == b.eval_host(pts[4], "llh")
    a.check_status()
    b.check_status()
    # negative pseudo-data: NaN and the status word (stats.py:231-240 raises), on both paths
    bad = data2.copy()
    bad[3] = -1.0
    for st in (a, b):
        st.set_data(bad)
        v = st.eval_host(pts[0], "llh")
        assert v != v and st.metric_status_host() != 0


@pytest.mark.parametrize("n_events", [120_000, 2_400_000, 9_999_996])
def test_partitioned_window_order_same_bits(n_events):
    """The 20 B form with 4 800 output bins: the resident order cut into partitions = the kernel's LDS windows
    (`window_partition_order`, pisa_hip_container::d_part_start: every deposit an LDS deposit, chunks that
    straddle a partition boundary flush twice) against the general window path (window placed by a scan of the
    chunk, deposits outside it through global atomics), the node order and the 40 B form: same limbs, bit for
    bit, and the same LLH through the unfused tail."""
    from pisa_amd import synthetic

    wl = synthetic.Workload(n_events=n_events, grid=(40, 30), out_binning="fine3d", seed=3)
    p = wl.osc_params(theta23_deg=44.0)
    ref = None
    for kw in (dict(compact=True), dict(compact=True, block_order=False), dict(compact=True, sort_events="node"),
               dict(compact=False)):
        st = synthetic.DeviceState(wl, **kw)
        if kw == dict(compact=True):
            assert st.index16 and all(c.d_part_start for c in st.cont) and st.cont[0].part_width == 672
        else:
            assert not any(c.d_part_start for c in st.cont)
        data = st.make_pseudo_data(wl.osc_params(), seed=0)
        st.accumulate(p)
        st.check_status()
        limbs = st.ws.limbs.clone()
        llh = st.eval_host(p, "llh")
        if ref is None:
            ref = (limbs, llh)
            assert int((limbs != 0).sum()) > 0
        elif kw.get("compact"):
            assert torch.equal(limbs, ref[0]), kw      # same weights (static factors folded first): same limbs
            assert llh == ref[1], kw
        else:
            assert abs(llh - ref[1]) <= 1e-9 * abs(ref[1])   # reference operation order: <= 3 ulp per weight


@pytest.mark.parametrize("n,n_nodes,n_bins,frac_out", [
    (833333, 20000, 128, 0.55), (20000, 2400, 128, 0.3), (4096 * 3, 600, 200, 0.0), (4096 * 3 + 17, 600, 200, 1.0),
    (1000, 60, 7, 0.5), (255, 10, 3, 0.2), (256, 10, 3, 0.0), (257, 1, 1, 0.0), (9000, 5, 31, 0.9), (70001, 19999, 4800, 0.4),
    # an ODD count beyond the size at which the library's radix sort switches to its one-sweep form: the sort's temporary
    # storage started at an odd multiple of 4 bytes and the launch hung (found under the counter passes of round 5)
    (3333333, 20000, 128, 0.55)])
def test_native_resident_order_is_the_torch_formulation(n, n_nodes, n_bins, frac_out):
    """`pisa_hip_deposit_block_order` (csrc/order.hip, round 5: one key, one radix sort, one workgroup per window, the block
    interleave in closed form) returns, element by element, the permutation of `engine.deposit_block_order` (the torch
    formulation: ~40 launches, four sorts, two host synchronisations per container) -- sizes around the 256-event block and the
    4 096-event window, no / only idle events, one node, more bins than bank pairs, events outside the grid (node -1)."""
    import torch

    from pisa_amd import engine
    from pisa_amd import kernels as K

    rs = np.random.RandomState(n % 9973)
    node = rs.randint(0, n_nodes, size=n).astype(np.int32)
    obin = rs.randint(0, n_bins, size=n).astype(np.int32)
    out = rs.rand(n) < frac_out
    obin[out & (rs.rand(n) < 0.7)] = -1
    node[out & (rs.rand(n) < 0.4)] = -1
    if frac_out == 1.0:
        obin[:] = -1
    d_node, d_bin = torch.from_numpy(node).to(K.device()), torch.from_numpy(obin).to(K.device())
    want = engine.deposit_block_order(d_bin, d_node, window=4096, banks=32).cpu().numpy()
    got = engine.deposit_block_order_native(d_bin, d_node, n_nodes).cpu().numpy()
    assert np.array_equal(np.sort(got), np.arange(n))
    assert np.array_equal(got, want)
    assert np.array_equal(got, engine.deposit_block_order_native(d_bin, d_node, n_nodes).cpu().numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("n,n_cols", [(1, 0), (255, 1), (256, 2), (100003, 3), (833333, 3)])
def test_native_pack_is_the_tensor_formulation(n, n_cols):
    """`pisa_hip_pack_resident_columns` (one launch per container at set-up) against the tensor operations of
    `HotPathEngine.__init__` it replaces: every permuted column, the interleaved pairs, the folded static weight and the
    16-bit index column with its padding, bit for bit -- a random permutation (also shorter than the sample: dropped events),
    negative indices in both columns"""
    import torch

    from pisa_amd import engine as E

    g = torch.Generator(device="cpu").manual_seed(n + n_cols)
    dev = torch.device("cuda")
    m = n + 17
    perm = torch.randperm(m, generator=g)[:n].to(dev)
    f = lambda *shape: torch.rand(shape, generator=g, dtype=torch.float64).to(dev)        # noqa: E731
    gx, gy, flux, aeff, w0 = f(m), f(m), f(m, 2), f(m), f(m)
    cols = [f(m) for _ in range(n_cols)]
    node = torch.randint(-1, 20000, (m,), generator=g, dtype=torch.int32).to(dev)
    obin = torch.randint(-1, 128, (m,), generator=g, dtype=torch.int32).to(dev)
    got = E.pack_resident_columns(perm, gx, gy, flux, aeff, w0, cols, node, obin)
    want_cols = [t[perm].contiguous() for t in (gx, gy, flux, aeff, w0)]
    for a, b in zip(got[:5], want_cols):
        assert torch.equal(a, b)
    for a, b in zip(got[5], cols):
        assert torch.equal(a, b[perm])
    nd, ob = node[perm], obin[perm]
    assert torch.equal(got[6], nd) and torch.equal(got[7], ob)
    assert torch.equal(got[8], torch.stack([nd, ob], dim=1))
    assert torch.equal(got[9], torch.stack([aeff[perm], w0[perm]], dim=1))
    assert torch.equal(got[10], w0[perm] * aeff[perm])
    n_pad = -(-n // 256) * 256
    v = torch.full((n_pad,), 0xFFFFFFFF, dtype=torch.int64, device=dev)
    v[:n] = torch.where(nd < 0, 0xFFFF, nd.long()) | (torch.where(ob < 0, 0xFFFF, ob.long()) << 16)
    assert torch.equal(got[11], torch.where(v >= 2 ** 31, v - 2 ** 32, v).to(torch.int32))


def test_warm_up_is_idempotent_and_leaves_nothing_behind():
    """`pisa_amd.warm_up()` (round 6): one small engine + one evaluation + one large pageable upload, synchronously or on a
    background thread; a second call does nothing; the memory it used is back with the allocator; an engine made
    afterwards gives the bits it gives without it."""
    import pisa_amd
    from pisa_amd import synthetic

    pisa_amd.warm_up(background=True)
    ms = pisa_amd.warm_up_wait()
    assert ms is not None and ms > 0
    pisa_amd.warm_up()                         # idempotent
    assert pisa_amd.warm_up_wait() == ms
    torch.cuda.synchronize()
    wl = synthetic.Workload(n_events=24000, grid=(24, 16), out_binning="dragon", seed=11)
    st = synthetic.DeviceState(wl)
    st.make_pseudo_data(wl.osc_params(), seed=1)
    assert np.isfinite(st.eval_host(wl.osc_params(theta23_deg=45.0), "llh"))


@pytest.mark.parametrize("n,n_nodes,n_bins,width,frac_out,n_wg", [
    (100003, 20000, 4800, 672, 0.6, 21), (100003, 20000, 4800, 672, 0.6, None), (833333, 20000, 4800, 672, 0.68, 21),
    (40000, 300, 1500, 672, 0.5, 5), (9000, 50, 700, 672, 0.3, 3), (300, 10, 1400, 672, 0.5, 2), (255, 10, 1400, 672, 0.5, None),
    (50000, 1000, 4800, 672, 0.02, 8), (50000, 1000, 4800, 672, 1.0, 8), (70001, 65000, 6000, 672, 0.5, 16)])
def test_native_partitioned_order_is_the_torch_formulation(n, n_nodes, n_bins, width, frac_out, n_wg):
    """`pisa_hip_partition_order_sort` / `_assemble` (csrc/order.hip, round 6) against `engine.window_partition_order`, the
    torch formulation (a nonzero + stable argsort per partition, the bank order, concatenations, an index shuffle): the same
    permutation element by element and the same partition table -- partitions above and below the 8 192 events from which
    the bank order applies, empty partitions, sizes around the 256-event block, events outside the grid, aligned and
    proportional sharing of the idle blocks, too few idle events (None from both), no depositing event at all."""
    from pisa_amd import engine
    from pisa_amd import kernels as K

    rs = np.random.RandomState(n % 9973 + n_bins)
    node = rs.randint(0, n_nodes, size=n).astype(np.int32)
    # (bins crowd into the lower partitions: sizes differ, the upper ones may stay empty)
    obin = np.minimum((rs.rand(n) ** 2 * n_bins).astype(np.int32), n_bins - 1)
    out = rs.rand(n) < frac_out
    obin[out & (rs.rand(n) < 0.7)] = -1
    node[out & (rs.rand(n) < 0.4)] = -1
    if frac_out == 1.0:
        obin[:] = -1
    d_node, d_bin = torch.from_numpy(node).to(K.device()), torch.from_numpy(obin).to(K.device())
    want = engine.window_partition_order(d_bin, d_node, n_bins, width, n_wg=n_wg)
    got = engine.window_partition_order_native(d_bin, d_node, n_bins, width, n_nodes, n_wg=n_wg)
    if want is None:
        assert got is None
        return
    assert got is not None and got[1] == want[1]
    g, w_ = got[0].cpu().numpy(), want[0].cpu().numpy()
    assert np.array_equal(np.sort(g), np.arange(n))
    assert np.array_equal(g, w_)
    again = engine.window_partition_order_native(d_bin, d_node, n_bins, width, n_nodes, n_wg=n_wg)
    assert np.array_equal(again[0].cpu().numpy(), g)
