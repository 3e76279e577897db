"""Multi-rank path on CPU (gloo, world_size 2 and 3): event sharding + integer
limb all-reduce.  The GPU kernels are not involved; what is checked is the
host-side contract that makes the LLH independent of the GPU count:
  * shards are contiguous, disjoint and cover every event;
  * per-rank exact fixed-point partial sums, SUM-all-reduced as int64, decode to
    the correctly rounded exact sum of ALL events -- the same bits for any world
    size (and equal to math.fsum)."""
import math
import os
import socket

import numpy as np
import pytest

from tests.conftest import bench_result
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _weights():
    rs = np.random.RandomState(12)
    n = 4001
    w = rs.rand(n) * 10 ** (rs.rand(n) * 30 - 20)
    w[::7] *= -1.0
    bins = rs.randint(0, 5, size=n)
    return w, bins


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pisa_amd.engine import allreduce_limbs, float_to_limbs, limbs_to_float, shard_bounds

    w, bins = _weights()
    lo, hi = shard_bounds(len(w), rank, world)
    limbs = np.zeros((5, 6), dtype=object)
    for x, b in zip(w[lo:hi], bins[lo:hi]):
        for j, v in enumerate(float_to_limbs(x)):
            limbs[b, j] += v
    t = torch.tensor(limbs.astype(np.int64))
    # the direct RCCL path is only taken on an RCCL ("nccl") group: on gloo every rank gets None
    from pisa_amd import rccl

    assert rccl.LimbAllReduce.create(torch.device("cpu")) is None
    allreduce_limbs(t, world)
    vals = [limbs_to_float(t[b].tolist()) for b in range(5)]
    covered = torch.tensor([hi - lo], dtype=torch.int64)
    dist.all_reduce(covered)
    if rank == 0:
        np.save(out_path, np.array(vals + [float(covered.item())]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_limb_allreduce_is_exact_and_world_size_independent(tmp_path, world):
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    res = np.load(out)
    w, bins = _weights()
    assert res[-1] == len(w)  # every event in exactly one shard
    exact = [math.fsum(w[bins == b]) for b in range(5)]
    np.testing.assert_array_equal(res[:5], exact)  # bit identical to the exact sum


def test_direct_rccl_needs_a_process_group():
    from pisa_amd import rccl

    assert rccl.LimbAllReduce.create(torch.device("cpu")) is None  # torch.distributed not initialised


def test_single_rank_equals_multi_rank_bits():
    from pisa_amd.engine import float_to_limbs, limbs_to_float, shard_bounds

    w, _ = _weights()
    full = [0] * 6
    for x in w:
        for j, v in enumerate(float_to_limbs(x)):
            full[j] += v
    parts = []
    for world in (1, 2, 4, 8):
        tot = [0] * 6
        seen = 0
        for r in range(world):
            lo, hi = shard_bounds(len(w), r, world)
            seen += hi - lo
            for x in w[lo:hi]:
                for j, v in enumerate(float_to_limbs(x)):
                    tot[j] += v
        assert seen == len(w)
        parts.append(limbs_to_float(tot))
    assert len(set(parts)) == 1 and parts[0] == limbs_to_float(full) == math.fsum(w)


# ---- the engine's OWN multi-rank control flow (sharding, eval -> accumulate -> allreduce -> tail)
#      on CPU ranks: only the two kernels are replaced by host stand-ins
def _toy_containers():
    rs = np.random.RandomState(5)
    out = []
    for k, n in enumerate((1001, 7, 640, 1)):      # ragged sizes, one container smaller than the world
        out.append(dict(true_energy=rs.rand(n), w0=rs.rand(n) * 10 ** (rs.rand(n) * 12 - 6),
                        bins=rs.randint(0, 6, size=n), sign=-1.0 if k == 2 else 1.0))
    return out


def _make_cpu_engine(rank, world):
    from types import SimpleNamespace

    from pisa_amd.engine import HotPathEngine, float_to_limbs, limbs_to_float, local_slices

    class CpuEngine(HotPathEngine):
        """HotPathEngine with its device kernels replaced: `accumulate` sums this rank's shard in
        exact fixed point on the host, `_tail` decodes the (all-reduced) limbs and evaluates a chi2.
        `eval`, `allreduce` and the sharding rule are the product's own code."""

        def __init__(self):  # pylint: disable=super-init-not-called
            self.dev = torch.device("cpu")
            self.rank, self.world_size, self.group, self._rccl = rank, world, None, None
            self.containers = _toy_containers()
            self._slices = local_slices([len(c["w0"]) for c in self.containers], rank, world)
            self.n_bins = 6
            self.ws = SimpleNamespace(limbs=torch.zeros((len(self.containers), 6, 2, 6), dtype=torch.int64))
            self.data = np.full(6, 3.0)
            self.metric_out = torch.zeros(1, dtype=torch.float64)
            self._out_block = None
            self.tail_calls = 0

        def accumulate(self, params=None):
            acc = np.zeros((len(self.containers), 6, 2, 6), dtype=object)
            for ci, (c, (lo, hi)) in enumerate(zip(self.containers, self._slices)):
                w = c["sign"] * c["w0"][lo:hi] * (1.0 + params * c["true_energy"][lo:hi])
                for x, b in zip(w, c["bins"][lo:hi]):
                    for j, v in enumerate(float_to_limbs(float(x))):
                        acc[ci, b, 0, j] += v
                    for j, v in enumerate(float_to_limbs(float(x) * float(x))):
                        acc[ci, b, 1, j] += v
            self.ws.limbs.copy_(torch.tensor(acc.astype(np.int64)))

        def _tail(self, kind, out):
            self.tail_calls += 1
            lim = self.ws.limbs
            hist = np.array([[limbs_to_float(lim[c, b, 0].tolist()) for b in range(6)]
                             for c in range(lim.shape[0])])
            total = hist.sum(axis=0)
            out[0] = float(np.sum((self.data - total) ** 2))
            self.hist = hist
            return out

    return CpuEngine()


def _engine_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = _make_cpu_engine(rank, world)
    vals = [float(eng.eval(p, "chi2")[0]) for p in (0.0, 0.37, 1.9)]
    assert eng._rccl is False          # gloo group: all ranks agreed on torch.distributed
    assert eng.tail_calls == 3
    n_local = torch.tensor([sum(hi - lo for lo, hi in eng._slices)], dtype=torch.int64)
    dist.all_reduce(n_local)
    # every rank holds the same maps and the same metric after the all-reduce
    mine = torch.tensor(vals + list(eng.hist.ravel()), dtype=torch.float64)
    ref = mine.clone()
    dist.broadcast(ref, src=0)
    assert torch.equal(mine, ref)
    if rank == 0:
        np.save(out_path, np.array(vals + [float(n_local.item())] + list(eng.hist.ravel())))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_engine_eval_control_flow_on_gloo_ranks(tmp_path, world):
    """HotPathEngine.eval / .allreduce / the shard partition run by `world` CPU processes give,
    bit for bit, what a single rank gives (the all-reduce adds integers)."""
    out = str(tmp_path / "eng.npy")
    mp.spawn(_engine_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    res = np.load(out)
    single = _make_cpu_engine(0, 1)
    want = [float(single.eval(p, "chi2")[0]) for p in (0.0, 0.37, 1.9)]
    assert single._rccl is None        # world_size 1: no collective at all
    assert res[3] == sum(len(c["w0"]) for c in _toy_containers())
    np.testing.assert_array_equal(res[:3], want)
    np.testing.assert_array_equal(res[4:], single.hist.ravel())


# ---- utils.kde on several ranks: containers dealt round-robin, one all-reduce of the finished maps
def _fake_map(i, n_bins):
    rs = np.random.RandomState(100 + i)
    return rs.rand(n_bins) * 10 ** (rs.rand(n_bins) * 8 - 4), rs.rand(n_bins)


def _kde_exchange_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pisa_amd.stages.utils.kde import kde as kde_stage

    n_cont, n_bins = 12, 37
    owned = kde_stage.owned_containers(n_cont, rank, world)
    local = {i: _fake_map(i, n_bins) for i in owned}
    full = kde_stage.exchange_maps(local, n_cont, n_bins, with_errors=True)
    n_owned = torch.tensor([len(owned)])
    dist.all_reduce(n_owned)
    assert int(n_owned) == n_cont                      # every container has exactly one owner
    assert sorted(full) == list(range(n_cont))
    if rank == world - 1:                              # any rank holds everything afterwards
        np.save(out_path, np.stack([np.stack(full[i]) for i in range(n_cont)]))
    dist.destroy_process_group()


def test_kde_containers_are_dealt_by_longest_processing_time():
    """utils.kde on several ranks: containers in order of decreasing event count, each to the least
    loaded rank -- complete, disjoint, the same on every rank, and balanced where round-robin is not"""
    from pisa_amd.stages.utils.kde import kde as kde_stage

    sizes = [900000, 40000, 30000, 850000, 20000, 10000, 870000, 35000, 25000, 860000, 15000, 5000]
    for world in (1, 2, 3, 8, 16):
        owned = [kde_stage.owned_containers(len(sizes), r, world, sizes=sizes) for r in range(world)]
        assert sorted(i for o in owned for i in o) == list(range(len(sizes)))
        loads = [sum(sizes[i] for i in o) for o in owned]
        if world <= 4:
            # the four large containers land on different ranks (as far as there are ranks)
            assert max(loads) <= sum(sizes) / world + max(sizes) * (1 - 1 / world) + 1
            assert max(loads) - min(loads) <= max(sizes)
    rr = [sum(sizes[i] for i in kde_stage.owned_containers(12, r, 3)) for r in range(3)]
    lpt = [sum(sizes[i] for i in kde_stage.owned_containers(12, r, 3, sizes=sizes)) for r in range(3)]
    assert max(lpt) < max(rr)       # round-robin puts all four large containers on rank 0
    eq = [len(kde_stage.owned_containers(12, r, 8, sizes=[1000] * 12)) for r in range(8)]
    assert sorted(eq) == [1, 1, 1, 1, 2, 2, 2, 2]


@pytest.mark.parametrize("world", [2, 5])
def test_kde_stage_map_exchange_on_gloo_ranks(tmp_path, world):
    """the KDE stage's multi-GPU rule: estimators are independent, so containers are dealt to the
    ranks and the finished maps exchanged -- bit-identical to what one rank computes"""
    out = str(tmp_path / "kde.npy")
    mp.spawn(_kde_exchange_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    for i in range(12):
        m, e = _fake_map(i, 37)
        np.testing.assert_array_equal(got[i, 0], m)
        np.testing.assert_array_equal(got[i, 1], e)


# ---- bench.py's own N > 1 control flow (process group, barriers, max over ranks, the legs that run on
#      several ranks, the JSON line) and HotPathEngine.eval_many's (one all-reduce of K limb sets, chunking)
#      on CPU ranks over gloo: the HIP launches are replaced by host stand-ins, everything else is the
#      product's / the bench's code
def _make_bench_state(wl, rank=0, world_size=1, points=None, **_kw):
    from pisa_amd.engine import HotPathEngine, allreduce_limbs, float_to_limbs, limbs_to_float, local_slices

    group = None
    if points is not None:   # hybrid point x event topology: the engine sees its coordinates inside its group
        kw = points.engine_kwargs()
        rank, world_size, group = kw["rank"], kw["world_size"], kw["group"]

    class CpuState(HotPathEngine):
        def __init__(self):  # pylint: disable=super-init-not-called
            self.dev = torch.device("cpu")
            self.rank, self.world_size, self.group, self._rccl = rank, world_size, group, None
            self.points = points
            self.wl = wl
            self.names = [ev["name"] for ev in wl.events]
            self.cont = list(wl.events)
            self.n_bins = 5
            self._slices = local_slices([len(ev["true_energy"]) for ev in wl.events], rank, world_size)
            self.n_local = sum(hi - lo for lo, hi in self._slices)
            self.index16, self.indexed, self.osc_events, self.node_flux, self.fused_tail = True, True, False, False, True
            self.plan, self.energy_d = object(), None
            n_c = len(self.cont)
            from types import SimpleNamespace

            self.ws = SimpleNamespace(limbs=torch.zeros((n_c, self.n_bins, 2, 6), dtype=torch.int64))
            self.metric_out = torch.zeros(1, dtype=torch.float64)
            self.data = None
            self._out_block = None
            self.sweeps = 0

        @staticmethod
        def _knob(params):
            return 100.0 * params.dm[6] + params.mix[8]

        def _limbs_of(self, params):
            acc = np.zeros((len(self.cont), self.n_bins, 2, 6), dtype=object)
            k = self._knob(params)
            for ci, (ev, (lo, hi)) in enumerate(zip(self.cont, self._slices)):
                w = ev["initial_weights"][lo:hi] * (1.0 + k * ev["true_coszen"][lo:hi] ** 2)
                b = np.minimum((ev["true_energy"][lo:hi] ** 0.3).astype(int), self.n_bins - 1)
                for x, bb in zip(w, b):
                    for j, v in enumerate(float_to_limbs(float(x))):
                        acc[ci, bb, 0, j] += v
                    for j, v in enumerate(float_to_limbs(float(x) * float(x))):
                        acc[ci, bb, 1, j] += v
            return torch.tensor(acc.astype(np.int64))

        def _metric_of(self, limbs):
            hist = np.array([[limbs_to_float(limbs[c, b, 0].tolist()) for b in range(self.n_bins)]
                             for c in range(limbs.shape[0])])
            total = hist.sum(axis=0)
            lam = np.maximum(total, 1e-10)
            return float(np.sum(self.data * np.log(lam) - lam)) if self.data is not None else float(total.sum())

        def accumulate(self, params=None):
            self.ws.limbs.copy_(self._limbs_of(params))

        def _tail(self, kind, out):
            out[0] = self._metric_of(self.ws.limbs)
            return out

        def eval_host(self, params, kind="llh"):
            self.accumulate(params)
            self.allreduce()
            return float(self._tail(kind, self.metric_out)[0])

        def make_pseudo_data(self, params, seed=0):
            self.data = None
            total = self.eval_host(params)
            self.data = np.full(self.n_bins, max(1.0, round(total / self.n_bins)))

        def check_status(self):
            pass

        # eval_many: the product's control flow around these two stand-ins
        def multi_capable(self, plan=None):
            return True

        def _multi_ws(self, k):
            return dict(limbs=torch.zeros((k, len(self.cont), self.n_bins, 2, 6), dtype=torch.int64), zero=True)

        def _many_sweep(self, w, params_list, scales, plan, energy):
            self.sweeps += 1
            for i, p in enumerate(params_list):
                w["limbs"][i] = self._limbs_of(p)

        def _many_tail(self, w, n, kind):
            return [self._metric_of(w["limbs"][i]) for i in range(n)]

    return CpuState()


def _bench_worker(rank, world, port, out_path):
    import json
    import sys

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    got = {}
    bench.main(["--gpus", str(world), "--events", "360", "--steps", "3", "--warmup", "1", "--min-timed-s", "0",
                "--grid", "12x8", "--legs", "all"],
               hooks=dict(device_state=_make_bench_state, legs=("multi_point", "point_parallel", "fit_c4_engine"), result=got.update))
    if rank == 0:
        with open(out_path, "w") as fh:
            json.dump(got, fh)


def test_bench_multi_gpu_control_flow_on_gloo_ranks(tmp_path):
    """`bench.py --gpus 2` end to end on two CPU ranks: the strong-scaling headline (events sharded, limbs
    all-reduced, MAX over ranks), the weak-scaling pass beside it, the multi-point leg (eval_many's
    all-reduce of K limb sets), one JSON line on rank 0 with the contract's fields -- and the LLH the two
    ranks report equals the one a single rank computes on the whole sample"""
    import json
    import sys

    out = str(tmp_path / "bench.json")
    mp.spawn(_bench_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    line = json.load(open(out))
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["steps"] == 3 and line["warmup"] == 1
    assert line["unit"] == "evals/s" and line["value"] > 0 and line["weak_value"] > 0
    assert set(line["legs"]) == {"multi_point", "point_parallel", "fit_c4_engine"}
    pp = line["legs"]["point_parallel"]["2x1"]         # two groups of one rank: the sample replicated, 18 points dealt 9 + 9
    assert pp["same_bits_as_event_sharded"] and pp["points_per_call"] == 18 and pp["evals_per_s"] > 0
    assert line["topology"] == "2x1" and line["point_parallel_evals_per_s"] == pp["evals_per_s"]
    fit = line["legs"]["fit_c4_engine"]
    assert fit["same_history"] and fit["point_by_point"]["llh_evaluations"] == fit["stencil_in_one_sweep"]["llh_evaluations"]
    mpt = line["legs"]["multi_point"]
    assert all(mpt["K%d" % k]["same_bits_as_point_by_point"] for k in (3, 5, 9))
    assert line["batched_evals_per_s5"] == mpt["K5"]["evals_per_s"]
    # the same sample on ONE rank: the same LLH bits (integer limbs: the shard count does not matter)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from pisa_amd import synthetic

    wl = synthetic.Workload(n_events=360, grid=(12, 8), out_binning="dragon", seed=0)
    single = _make_bench_state(wl)
    single.make_pseudo_data(wl.osc_params(), seed=0)
    last = bench.param_list(wl, 4)[-1]
    assert single.eval_host(last, "llh") == line["last_llh"]
    # eval_many on one rank: same values, one sweep per batch, batches beyond MAX_POINTS are split
    from pisa_amd import _lib

    pts = bench.param_list(wl, _lib.MAX_POINTS + 2)
    assert single.eval_many(pts, "llh") == [single.eval_host(p, "llh") for p in pts]
    assert single.sweeps == 2


def _point_groups_worker(rank, world, port, n_groups, out_dir):
    import sys

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from pisa_amd import synthetic
    from pisa_amd.engine import PointGroups

    pg = PointGroups(rank, world, n_groups)
    assert pg.topology == "%dx%d" % (n_groups, world // n_groups)
    wl = synthetic.Workload(n_events=360, grid=(12, 8), out_binning="dragon", seed=0)
    st = _make_bench_state(wl, points=pg)
    assert (st.rank, st.world_size) == (rank % (world // n_groups), world // n_groups)
    st.make_pseudo_data(wl.osc_params(), seed=0)     # (inside the group: sharded + reduced over its ranks only)
    out = {}
    for k in (1, 2, 3, 5, 7):
        pts = bench.param_list(wl, k)
        sweeps = st.sweeps
        out["K%d" % k] = st.eval_many(pts, "llh")
        lo, hi = pg.block(k)
        assert st.sweeps - sweeps == (1 if hi - lo > 1 else 0)     # this group's block in ONE sweep of its events
    np.save(os.path.join(out_dir, "r%d.npy" % rank), np.array([out["K%d" % k] for k in (1, 2, 3, 5, 7)], dtype=object),
            allow_pickle=True)
    dist.destroy_process_group()


@pytest.mark.parametrize("n_groups", [1, 2, 4])
def test_point_groups_deal_points_and_keep_the_single_rank_bits(tmp_path, n_groups):
    """Hybrid point x event parallelism (`engine.PointGroups`): four gloo ranks as 1 x 4 (event sharding alone), 2 x 2 and
    4 x 1 (the sample replicated, points dealt); `eval_many` of 1, 2, 3, 5 and 7 points -- fewer points than groups
    included: 3 points on 4 groups leave one group idle and the blocks ragged (0, 1, 1, 1) -- returns on EVERY rank the
    list one rank computes on the whole sample, bit for bit (a point is evaluated entirely inside one group; integer limbs)."""
    import sys

    mp.spawn(_point_groups_worker, args=(4, _free_port(), n_groups, str(tmp_path)), nprocs=4, join=True)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from pisa_amd import synthetic

    wl = synthetic.Workload(n_events=360, grid=(12, 8), out_binning="dragon", seed=0)
    single = _make_bench_state(wl)
    single.make_pseudo_data(wl.osc_params(), seed=0)
    want = [[single.eval_host(p, "llh") for p in bench.param_list(wl, k)] for k in (1, 2, 3, 5, 7)]
    from pisa_amd.engine import PointGroups as _PG

    assert [_PG(0, 4, 4, make_group=lambda r: tuple(r)).block(3, g) for g in range(4)] == [(0, 0), (0, 1), (1, 2), (2, 3)]
    for rank in range(4):
        got = np.load(str(tmp_path / ("r%d.npy" % rank)), allow_pickle=True)
        for g, w_ in zip(got, want):
            assert list(g) == w_
    from pisa_amd.engine import PointGroups

    with pytest.raises(ValueError):
        PointGroups(0, 4, 3)
    pg = PointGroups(3, 4, 2, make_group=lambda ranks: tuple(ranks))
    assert (pg.group_id, pg.shard, pg.shard_group) == (1, 1, (2, 3))
    assert [pg.block(7, g) for g in range(2)] == [(0, 3), (3, 7)]


def bench_hooks():
    """what `PISA_BENCH_HOOKS=tests.test_distributed_cpu:bench_hooks` hands to bench.main in every rank"""
    return dict(device_state=_make_bench_state, legs=("multi_point",))


def test_bench_gpus_2_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2 ...` as a PLAIN subprocess, the way the driver runs N = 1: the parent
    starts torch.distributed.run with two fresh ranks itself (gloo stand-in selected by the environment
    variable it forwards), relays ONE JSON line and the exit code; the line says that both ranks hold the
    same LLH bits.  Without the stand-in and without two HIP devices it must end non-zero within seconds."""
    import json
    import subprocess
    import sys
    import time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--events", "360", "--steps", "3",
           "--warmup", "1", "--min-timed-s", "0", "--grid", "12x8", "--legs", "multi_point",
           "--detail-out", str(tmp_path / "detail.json")]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PISA_BENCH_HOOKS"] = "tests.test_distributed_cpu:bench_hooks"
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    compact, line = bench_result(res.stdout, str(tmp_path / "detail.json"))
    assert compact["legs_run"] == ["multi_point"] and compact["llh_bits_identical"] is True and compact["hooks_used"] is True
    assert compact["llh_bits"] == line["llh_bits_per_rank"][0] and compact["batched_evals_per_s9"] > 0
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["value"] > 0 and line["scaling"] == "strong"
    assert line["llh_bits_identical"] is True and len(line["llh_bits_per_rank"]) == 2
    assert set(line["legs"]) == {"multi_point"}
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        env.pop("PISA_BENCH_HOOKS")
        t0 = time.time()
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        assert res.returncode != 0 and not res.stdout.strip()
        assert "--gpus 2 asked for" in res.stderr
        assert time.time() - t0 < 120


# ---- utils.kde.apply_function itself on several ranks (CPU, gloo): the containers it owns, the lazily produced event
#      weights, ONE batched estimator call per rank, the exchange of the finished maps -- with the native batch call
#      replaced by a host stand-in (the stage's own code otherwise)
class _FakeContainer(dict):
    def __init__(self, name, size):
        super().__init__()
        self.name, self.size, self.representation, self.weights_asked = name, size, None, 0

    def device(self, key):
        assert key == "weights" and self.representation == "events"
        self.weights_asked += 1
        return torch.full((self.size,), float(len(self.name)), dtype=torch.float64)


class _FakeData(list):
    representation = None


def _kde_stage_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pisa_amd.core.binning import MultiDimBinning, OneDimBinning
    from pisa_amd.stages.utils import kde as kde_mod
    from pisa_amd.utils import kde_hist

    binning = MultiDimBinning([OneDimBinning("reco_energy", domain=[1.0, 10.0], num_bins=3, is_lin=True),
                               OneDimBinning("reco_coszen", domain=[-1, 1], num_bins=2, is_lin=True),
                               OneDimBinning("pid", bin_edges=[0.0, 0.5, 1.0], is_lin=True)])
    stage = kde_mod.kde(calc_mode="events", apply_mode=binning)
    stage.regularized_apply_mode = binning
    sizes = [500, 40, 30, 450, 20, 10, 470, 35, 25, 460, 15, 5]
    stage.data = _FakeData(_FakeContainer("c%02d" % i, n) for i, n in enumerate(sizes))
    stage._static_sample = lambda c: dict(sample=torch.zeros((c.size, 3), dtype=torch.float64), channels=None, versions=())
    calls = []

    def fake_batch(samples, **kw):
        calls.append(len(samples))
        assert kw["n_threads"] == stage.kde_workers and kw["binning"] is binning
        out = []
        for s in samples:
            w = s["weights"]()          # the stage hands over a callable: evaluated when the sample's turn comes
            out.append(np.full(binning.shape, float(w.sum())))
        return out

    kde_hist_batch = kde_hist.kde_histogramdd_batch
    kde_hist.kde_histogramdd_batch = fake_batch
    try:
        stage.apply_function()
    finally:
        kde_hist.kde_histogramdd_batch = kde_hist_batch
    owned = kde_mod.kde.owned_containers(len(sizes), rank, world, sizes=sizes)
    assert calls == [len(owned)]                                   # one batched call with exactly this rank's containers
    assert [c.weights_asked for c in stage.data] == [1 if i in owned else 0 for i in range(len(sizes))]
    for i, c in enumerate(stage.data):                             # every rank ends with every map
        np.testing.assert_array_equal(c["weights"], np.full(int(np.prod(binning.shape)), float(sizes[i] * len(c.name))))
    if rank == 0:
        np.save(out_path, np.array([c["weights"][0] for c in stage.data]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_kde_stage_apply_function_on_gloo_ranks(tmp_path, world):
    out = str(tmp_path / "stage.npy")
    mp.spawn(_kde_stage_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    sizes = [500, 40, 30, 450, 20, 10, 470, 35, 25, 460, 15, 5]
    np.testing.assert_array_equal(np.load(out), np.array([3.0 * n for n in sizes]))
