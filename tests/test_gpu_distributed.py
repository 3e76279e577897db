"""The N > 1 path on hardware with ONE GPU: `python bench.py --gpus 2|4` as the driver would run it, its ranks all on
HIP device 0 and exchanging over gloo (a test-only hook, `bench.py: share_device`).  Everything else is the product
path: the self-launch, the sharded device state, the real kernels on every rank's contiguous shard of the events, the
integer limb all-reduce between the fused kernel and the tail (through `torch.distributed`, the fall-back of the
direct RCCL call), barriers and max-over-ranks timing, the weak-scaling leg, the bit-identity check of the line.
What it cannot cover is RCCL itself over xGMI (one rank per GPU): `tests/test_gpu_engine.py` runs the RCCL
communicator with one rank, `tests/test_distributed_cpu.py` the control flow on CPU ranks."""
import json
import os
import subprocess
import sys

import pytest

from tests.conftest import bench_result

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def share_device_hooks():
    """what `PISA_BENCH_HOOKS=tests.test_gpu_distributed:share_device_hooks` hands to bench.main in every rank"""
    return dict(share_device=True, legs=())


def share_device_hooks_points():
    return dict(share_device=True, legs=("point_parallel",))


def _bench(gpus, hooks=True, legs="none"):
    import tempfile

    detail = os.path.join(tempfile.mkdtemp(prefix="pisa_bench_"), "detail.json")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--events", "2400000", "--steps", "6",
           "--warmup", "2", "--min-timed-s", "0", "--legs", legs, "--no-cpu-baseline", "--no-drop-probe", "--no-batch-probe",
           "--detail-out", detail]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    if hooks:
        env["PISA_BENCH_HOOKS"] = "tests.test_gpu_distributed:share_device_hooks" + ("_points" if legs != "none" else "")
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-4000:]
    compact, line = bench_result(res.stdout, detail)
    assert compact["n_gpus"] == gpus and compact.get("llh_bits_identical") == line["llh_bits_identical"]
    return line


def test_two_and_four_ranks_on_one_device_reproduce_the_single_rank_bits():
    one = _bench(1, hooks=False)
    assert one["n_gpus"] == 1 and one["llh_bits_per_rank"] is None and one["config"]["events"] >= 2399990
    for n in (2, 4):
        line = _bench(n)
        assert line["n_gpus"] == n and line["scaling"] == "strong" and line["steps"] == 6 and line["value"] > 0
        assert line["llh_bits_identical"] is True and len(line["llh_bits_per_rank"]) == n and len(set(line["llh_bits_per_rank"])) == 1
        # the same events, sharded differently, summed as integers: the very same LLH as one rank holds, bit for bit
        assert line["last_llh"] == one["last_llh"], (n, line["last_llh"], one["last_llh"])
        assert line["weak"]["samples_per_step"] == n and line["weak"]["value"] > 0
        assert line["phase_ms"]["events_this_rank"] * n >= 2399990 and line["nccl_comm_count"] is None    # gloo here


def test_point_groups_on_one_device_reproduce_the_single_rank_bits():
    """Hybrid point x event parallelism (`engine.PointGroups`, round 5) with the real kernels: four ranks on HIP device 0 as
    4 x 1 (the sample replicated per rank, 36 points dealt 9 each) and 2 x 2 (two shards per group, the limb all-reduce
    inside the group): the values every rank holds after the all-gather are, bit for bit, those of the event-sharded
    engine on the same points -- which are the single-rank bits (the test above)."""
    line = _bench(4, legs="point_parallel")
    pp = line["legs"]["point_parallel"]
    assert set(pp) >= {"4x1", "2x2"}
    for topo, k in (("4x1", 36), ("2x2", 18)):
        assert pp[topo]["same_bits_as_event_sharded"] is True and pp[topo]["points_per_call"] == k and pp[topo]["evals_per_s"] > 0
    assert line["topology"] == "4x1" and line["point_parallel_evals_per_s"] == pp["4x1"]["evals_per_s"]


def test_random_shardings_on_one_device_reproduce_the_single_rank_bits():
    """`scripts/dev/fuzz_ranks.py`: 2-6 ranks on HIP device 0 over gloo, random samples (down to fewer events than ranks),
    grids, binnings, layouts, metrics; single points through the one-call evaluator or the three calls, several points
    in one sweep -- every value and every map bit for bit the single-rank engine's.  Round 4: 200 trials, no difference."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "dev", "fuzz_ranks.py"), "6", "909"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, (res.stdout[-3000:], res.stderr[-2000:])
    assert "6 trials, 0 bad" in res.stdout


def _dist_case(n_ranks, case, out_dir, *args):
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_cases.py"), case, str(out_dir)] + [str(a) for a in args]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env.setdefault("OMP_NUM_THREADS", "1")
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-4000:]
    return [json.load(open(os.path.join(str(out_dir), "fit_r%d.json" % r))) for r in range(n_ranks)]


def test_fit_under_configured_point_groups_follows_the_single_rank_fit(tmp_path):
    """`Analysis.fit_hypo(batched_gradient=True)` end to end under `engine.configure_point_groups` (round-5 verdict, Next
    #7a): the cfg-text pipeline's hist stage builds its engine with the configured topology, the stencil of every
    L-BFGS-B iterate is dealt to the groups (2 x 1: two groups, the sample replicated; 2 x 2 on four ranks: two shards per
    group, limbs all-reduced inside the group) and gathered -- and the fit history (metric and parameters of every
    evaluation) is, value for value, the one a single rank computes."""
    one = tmp_path / "one"
    one.mkdir()
    want = _dist_case(1, "fit_point_groups", one, 0)[0]
    assert want["topology"] is None and want["evaluations"] >= 6 and len(want["history"]) == want["evaluations"]
    for n_ranks, n_groups, topo in ((2, 2, "2x1"), (4, 2, "2x2")):
        out = tmp_path / topo
        out.mkdir()
        got = _dist_case(n_ranks, "fit_point_groups", out, n_groups)
        for g in got:
            assert g["topology"] == topo and g["engine_world"] == n_ranks // n_groups
            assert g["history"] == want["history"] and g["metric_val"] == want["metric_val"], (topo, g["rank"])
            assert g["evaluations"] == want["evaluations"]
