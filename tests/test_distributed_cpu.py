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
