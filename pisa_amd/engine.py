"""Device-resident evaluation engine of the hot path.

One `HotPathEngine` per process / GPU owns the event columns of all containers
in HBM (uploaded once) and evaluates, per parameter point,

    prob3 on the calc grid (nu + nubar)                 1 launch
    fused lookup + reweight + histogram(+sumw2)         1 launch (+1 slab reduce)
    [all-reduce of the integer histogram limbs]         RCCL, only if world_size > 1
    fixed point -> fp64 maps, LLH / chi2 reduction      2 launches

Per-eval host->device traffic is the kernel-argument blocks only; device->host
is the 8-byte metric (or the maps on `get_outputs`).

This is what the Stage classes in ``pisa_amd.stages`` drive when a pipeline has
the shape  loader -> [flux] -> osc.prob3 -> aeff.aeff -> utils.hist
(pisa/core/pipeline.py:537-558 runs those stages one after another on host
numpy arrays; here the three apply_functions are one pass over HBM).
"""
import os

import numpy as np
import torch

from . import _lib
from . import kernels as K


class GridSpec:
    """2-D (true_energy x true_coszen) calc grid.

    Node coordinates are the reference's `weighted_centers`
    (pisa/core/binning.py:901-911): geometric mean of log-spaced edges
    (`np.logspace`, binning.py:416-421), midpoints of linear ones; a binned
    container flattens `meshgrid(indexing='ij')` (container.py:769-773), so the
    node index is iE*n_cz + jcz when the binning order is (energy, coszen).
    Lookups use the regularised binning: ln(E) for the log dimension
    (container.py:992-1005)."""

    def __init__(self, e_range=(1.0, 1000.0), n_e=200, cz_range=(-1.0, 1.0), n_cz=200,
                 energy_first=True):
        self.n_e, self.n_cz, self.energy_first = int(n_e), int(n_cz), bool(energy_first)
        self.e_edges = np.logspace(np.log10(e_range[0]), np.log10(e_range[1]), self.n_e + 1)
        self.cz_edges = np.linspace(cz_range[0], cz_range[1], self.n_cz + 1)
        self.energy = np.sqrt(self.e_edges[:-1] * self.e_edges[1:])
        self.coszen = 0.5 * (self.cz_edges[:-1] + self.cz_edges[1:])
        ln_dom = np.log(np.array([e_range[0], e_range[1]], dtype=np.float64))
        if self.energy_first:
            mins, maxs, nb = [ln_dom[0], cz_range[0]], [ln_dom[1], cz_range[1]], [self.n_e, self.n_cz]
        else:
            mins, maxs, nb = [cz_range[0], ln_dom[0]], [cz_range[1], ln_dom[1]], [self.n_cz, self.n_e]
        self.binning = _lib.make_binning(mins, maxs, nb)

    @property
    def size(self):
        return self.n_e * self.n_cz


def shard_bounds(n, rank, world_size):
    """contiguous equal slices of a container's events over the ranks"""
    return (rank * n) // world_size, ((rank + 1) * n) // world_size


def allreduce_limbs(limbs, world_size, group=None):
    """Integer SUM all-reduce of the histogram limbs over RCCL/xGMI (gloo in the
    CPU tests).  The limbs are exact fixed-point partial sums, so the reduced
    value -- and every map / LLH derived from it -- does not depend on the ring
    order or on the number of ranks."""
    if world_size > 1:
        import torch.distributed as dist

        dist.all_reduce(limbs, op=dist.ReduceOp.SUM, group=group)
    return limbs


def bin_window_order(obin, n_bins, lds_window=682):
    """Event order for binnings too large for LDS accumulators (> 682 bins).

    The fused kernel then keeps only a WINDOW of the binning in LDS: the
    `lds_window` bins that start at the lowest bin of a workgroup's chunk
    (hist.hip); deposits outside it fall back to global atomics.  So events are
    sorted by output bin -- a chunk of ~10^4 consecutive events spans few bins --
    and, inside segments of S consecutive events, dealt round-robin over the
    segment's bins, so that the 64 lanes of a wavefront add into different bins
    instead of queueing on one LDS address.  S is chosen so that a segment holds
    ~192 bins (> 128 lanes' worth, and chunk + 2 segments stay inside the window).
    Returns the permutation (device int64)."""
    n = obin.numel()
    perm1 = torch.argsort(obin, stable=True)
    if n < 2:
        return perm1
    b = obin[perm1].long()
    per_bin = max(1.0, n / max(1, n_bins))
    seg_len = int(min(max(1024, 192 * per_bin), max(1024, 0.3 * lds_window * per_bin)))
    idx = torch.arange(n, device=obin.device)
    is_start = torch.ones(n, dtype=torch.bool, device=obin.device)
    is_start[1:] = b[1:] != b[:-1]
    run_start = torch.cummax(torch.where(is_start, idx, torch.zeros_like(idx)), 0).values
    seg = idx // seg_len
    rank_in_run = idx - torch.maximum(run_start, seg * seg_len)
    key = (seg * (seg_len + 1) + rank_in_run) * (n_bins + 2) + (b + 1)
    return perm1[torch.argsort(key, stable=True)]


def bin_partition_order(obin, node, n_bins, width=672):
    """Event order for binnings too large for LDS accumulators, second form ("part").

    The binning is cut into partitions of `width` consecutive bins (one LDS window of the fused
    kernel each, hist.hip); an event goes to the partition of its bin, the events outside the
    binning -- which deposit nothing -- are spread evenly over the partitions, and inside a
    partition the events are sorted by calc-grid node.  A workgroup's chunk then lies in one
    partition (all of its deposits are LDS deposits), every chunk carries the same share of
    depositing events, the table gathers keep their locality, and `lds_bank_order` can take
    the bank conflicts out as for small binnings.  Returns the permutation (device int64)."""
    n = obin.numel()
    n_win = max(1, -(-int(n_bins) // width))
    idx = torch.arange(n, device=obin.device)
    win = torch.where(obin >= 0, obin.long() // width, idx % n_win)
    key = win * (int(node.max().item()) + 3 if n else 1) + (node.long() + 1)
    return torch.argsort(key, stable=True)


def _aligned_partition_blocks(dep_blocks, total_blocks, n_wg):
    """blocks per partition such that every partition is a whole number of a workgroup's share (`chunk` =
    ceil(total / n_wg) blocks; the last partition takes what is left): [blocks], or None if it cannot be done
    (fewer workgroups than non-empty partitions, a partition's depositing blocks beyond its share)"""
    n_part = len(dep_blocks)
    if n_wg < 1 or total_blocks < 1:
        return None
    chunk = -(-total_blocks // n_wg)
    wg_used = -(-total_blocks // chunk)                 # workgroups that really receive events
    need = [-(-b // chunk) for b in dep_blocks]         # workgroups a partition needs at the very least
    last = max((p for p in range(n_part) if dep_blocks[p] > 0), default=None)
    if last is None or sum(need) > wg_used:
        return None
    g = list(need)
    spare = wg_used - sum(g)
    tot = max(1, sum(dep_blocks))
    # the spare workgroups in proportion to the depositing blocks (largest remainder)
    want = [wg_used * b / tot for b in dep_blocks]
    while spare > 0:
        p = max(range(n_part), key=lambda p: want[p] - g[p])
        g[p] += 1
        spare -= 1
    sizes = [g[p] * chunk for p in range(n_part)]
    # the partitions behind the last depositing one are empty; the last depositing one ends with the column
    for p in range(last + 1, n_part):
        sizes[p] = 0
    sizes[last] = total_blocks - sum(sizes[:last])
    if sizes[last] < dep_blocks[last] or any(sz < b for sz, b in zip(sizes, dep_blocks)):
        return None
    return sizes


def _partition_accounting(n, n_dep, n_idle, block, n_wg):
    """the host side of the partitioned order: from the number of depositing events of every partition and of idle events,
    (top, dep_blocks, share) -- idle events that complete a partition's last depositing block, its depositing blocks, the
    idle blocks it is interleaved with -- or None when the container has too few idle events to top the partitions up"""
    n_part = len(n_dep)
    top = [(-k) % block for k in n_dep]                      # idle events that complete the last depositing block
    spare = int(n_idle) - sum(top)
    n_blocks = n // block                                    # (the tail beyond whole blocks stays idle, see below)
    if spare < 0 or n_blocks == 0:
        return None
    dep_blocks = [(k + t) // block for k, t in zip(n_dep, top)]
    idle_blocks = (n - sum(k + t for k, t in zip(n_dep, top))) // block
    share = None
    if n_wg:
        # partitions of whole per-workgroup shares (`pisa_hip_hist_workgroups`): no workgroup's chunk reaches
        # into a second partition
        sizes = _aligned_partition_blocks(dep_blocks, -(-n // block), int(n_wg))   # (the padded column)
        if sizes is not None:
            share = [sz - b for sz, b in zip(sizes, dep_blocks)]
            if n % block:        # the partial last block is made of the idle tail, not of a whole idle block
                last = max(p for p in range(n_part) if sizes[p] > 0)
                share[last] -= 1
            if min(share) < 0 or sum(share) != idle_blocks:
                share = None
    if share is None:
        # idle blocks dealt to the partitions in proportion to their depositing blocks (largest remainder)
        tot = max(1, sum(dep_blocks))
        share = [idle_blocks * b // tot for b in dep_blocks]
        rem = idle_blocks - sum(share)
        for p in sorted(range(n_part), key=lambda p: -(idle_blocks * dep_blocks[p] % tot))[:rem]:
            share[p] += 1
    return top, dep_blocks, share


def window_partition_order(obin, node, n_bins, width, block=256, n_wg=None):
    """Resident order of the 16-bit index form for a binning beyond the LDS accumulators (`width` =
    `pisa_hip_hist_window_bins`): the events are cut into partitions, partition p holding every event that
    deposits into bins [p width, (p+1) width) -- sorted by calc-grid node and in the LDS-bank-aware order, as for
    small binnings -- topped up with events that deposit nothing to a whole number of `block`-event blocks
    (what one wavefront takes per sweep) and interleaved with further idle blocks so that every stretch of the
    resident order carries the same share of depositing events.  The fused kernel then works a chunk through
    partition by partition with its LDS window on that partition's bins (pisa_hip_container::d_part_start).
    Returns (permutation, part_start in blocks [n_part + 1]), or None when the container has too few idle
    events to top the partitions up (the caller keeps the general order).  This torch formulation (~25 launches
    per container) is the specification; the engine calls `window_partition_order_native`."""
    n = int(obin.numel())
    dev = obin.device
    n_part = max(1, -(-int(n_bins) // int(width)))
    dep = (obin >= 0) & (node >= 0)
    idle = torch.nonzero(~dep).reshape(-1)
    idle = idle[torch.argsort(node[idle], stable=True)]
    part_of = torch.where(dep, obin.long() // int(width), torch.full_like(obin, -1, dtype=torch.int64))
    deps = []
    for p in range(n_part):
        ip = torch.nonzero(part_of == p).reshape(-1)
        ip = ip[torch.argsort(node[ip], stable=True)]
        if ip.numel() >= 8192:
            ip = ip[lds_bank_order(obin[ip], window=4096, banks=32, per=4)]
        deps.append(ip)
    n_dep = [int(d.numel()) for d in deps]
    acc = _partition_accounting(n, n_dep, int(idle.numel()), block, n_wg)
    if acc is None:
        return None
    top, dep_blocks, share = acc
    pieces, starts, at = [], [0], 0
    for p in range(n_part):
        body = torch.cat((deps[p], idle[at:at + top[p]]))
        at += top[p]
        filler = idle[at:at + share[p] * block]
        at += share[p] * block
        nb, nf = dep_blocks[p], share[p]
        if nb and nf:
            # depositing blocks spread evenly among the idle ones
            t = nb + nf
            pos_b = (torch.arange(nb, device=dev) * t) // nb
            is_b = torch.zeros(t, dtype=torch.bool, device=dev)
            is_b[pos_b] = True
            src = torch.empty(t, dtype=torch.int64, device=dev)
            src[is_b] = torch.arange(nb, device=dev)
            src[~is_b] = torch.arange(nb, t, device=dev)
            body = torch.cat((body, filler)).view(t, block)[src].reshape(-1)
        else:
            body = torch.cat((body, filler))
        pieces.append(body)
        starts.append(starts[-1] + (nb + nf))
    pieces.append(idle[at:])            # fewer than `block` events: behind the last partition, padded by the engine
    perm = torch.cat(pieces)
    assert perm.numel() == n
    starts[-1] = -(-n // block)         # the last partition takes the (idle) tail and its padding
    return perm, starts


def window_partition_order_native(obin, node, n_bins, width, n_nodes, n_wg=None):
    """`window_partition_order` through `pisa_hip_partition_order_sort` / `_assemble` (csrc/order.hip, round 6): one key per
    event and ONE stable radix sort instead of a nonzero + argsort per partition, the bank order of the large partitions by
    the window kernel, every output position's source in closed form; the block accounting in between is
    `_partition_accounting`, on the host, as in the torch formulation.  The SAME (permutation, part_start)
    (tests/test_gpu_engine.py compares them element by element), or None under the same condition."""
    import ctypes as C

    lib = _lib.lib()
    n = int(obin.numel())
    n_part = max(1, -(-int(n_bins) // int(width)))
    if n == 0 or n_part > 255 or (n_part + 1) * (int(n_nodes) + 1) > 0xFFFFFFFF:
        return window_partition_order(obin, node, n_bins, width, n_wg=n_wg)
    need = int(lib.pisa_hip_partition_order_workspace(n))
    if need < 0:
        raise ValueError("too many events for one container's order: %d" % n)
    work = torch.empty(need, dtype=torch.uint8, device=obin.device)
    node32 = (node if node.dtype == torch.int32 else node.to(torch.int32)).contiguous()
    obin32 = (obin if obin.dtype == torch.int32 else obin.to(torch.int32)).contiguous()
    counts = (C.c_int64 * (n_part + 1))()
    _lib.check(lib.pisa_hip_partition_order_sort(C.c_void_p(node32.data_ptr()), C.c_void_p(obin32.data_ptr()), n, int(n_nodes),
                                                 int(width), n_part, counts, C.c_void_p(work.data_ptr()), need, K._stream()))
    n_dep = [int(counts[p]) for p in range(n_part)]
    acc = _partition_accounting(n, n_dep, int(counts[n_part]), 256, n_wg)
    if acc is None:
        return None
    _, dep_blocks, share = acc
    perm = torch.empty(n, dtype=torch.int64, device=obin.device)
    arr = C.c_int64 * n_part
    _lib.check(lib.pisa_hip_partition_order_assemble(C.c_void_p(obin32.data_ptr()), n, n_part, arr(*n_dep), arr(*dep_blocks),
                                                     arr(*share), C.c_void_p(perm.data_ptr()), C.c_void_p(work.data_ptr()), need,
                                                     K._stream()))
    starts = [0]
    for nb, nf in zip(dep_blocks, share):
        starts.append(starts[-1] + nb + nf)
    starts[-1] = -(-n // 256)           # the last partition takes the (idle) tail and its padding
    return perm, starts


def lds_bank_order(obin, window=4096, banks=32, per=2):
    """Second-level event order (applied on top of the node sort) that takes the LDS bank
    conflicts out of the deposits.

    An fp64 LDS accumulator occupies one of `banks` = 32 bank pairs, selected by the output
    bin modulo 32 (all other terms of the accumulator address are multiples of 32 elements).
    The 32 lanes of a half-wavefront deposit the k-th event of their pair in one `ds_add_f64`;
    if those 32 events have 32 different bins modulo 32 the instruction is conflict free.
    Inside windows of `window` events (so the node locality of the table gathers survives)
    the events are therefore dealt round-robin over the 32 residues, and the emitted sequence
    is laid out so that 32 consecutive emissions land in the same slot (first or second event)
    of 32 consecutive pairs (`per` = 2 events per thread and sweep; 4 for the 16-bit index
    form, whose threads take quads).  The accumulation is exact, so this is free to choose.
    Returns a permutation (device int64) of arange(n)."""
    n = obin.numel()
    dev = obin.device
    idx = torch.arange(n, device=dev)
    n_full = (n // window) * window
    if n_full == 0:
        return idx
    b = obin[:n_full].long()
    # events outside the binning deposit nothing: spread them evenly over the residues
    res = torch.where(b >= 0, b % banks, idx[:n_full] % banks)
    win = idx[:n_full] // window
    key1 = win * banks + res
    order1 = torch.argsort(key1, stable=True)
    k1 = key1[order1]
    start = torch.ones(n_full, dtype=torch.bool, device=dev)
    start[1:] = k1[1:] != k1[:-1]
    pos = torch.arange(n_full, device=dev)
    group_start = torch.cummax(torch.where(start, pos, torch.zeros_like(pos)), 0).values
    rank = pos - group_start                       # k-th event of its (window, residue) queue
    key2 = (win[order1] * (window + 1) + rank) * banks + res[order1]
    seq = order1[torch.argsort(key2, stable=True)]  # emission order: one event per residue in turn
    s = pos % window
    blk, t = s // (32 * per), s % (32 * per)
    slot = (pos // window) * window + per * (blk * 32 + (t % 32)) + (t // 32)
    perm = idx.clone()
    perm[slot] = seq
    return perm


def deposit_block_order(obin, node, block=256, window=4096, banks=32):
    """Resident order of the 16-bit index form: the events that can deposit (inside the output binning
    AND the calc grid) are gathered into whole blocks of `block` = 256 events -- what ONE wavefront
    takes per sweep (64 lanes x a quad) -- and those blocks are spread evenly among the blocks of events
    that deposit nothing.  A wavefront then either deposits with all of its lanes or skips the deposit
    code altogether (wave-uniform branch), instead of every wavefront running it with the third of its
    lanes that happen to be inside the binning: the fused kernel issues a third of the LDS atomics and
    of the split arithmetic, which is what bounds the multi-point kernel (`eval_many`).  Inside the
    depositing events: sorted by calc-grid node, then the LDS-bank-aware order (`lds_bank_order`);
    blocks move whole, so both survive.  The sums are exact, so the order is free to choose.
    Returns a permutation (device int64)."""
    n = obin.numel()
    dev = obin.device
    dep = (obin >= 0) & (node >= 0)
    idx_b = torch.nonzero(dep).reshape(-1)
    idx_u = torch.nonzero(~dep).reshape(-1)
    pb = idx_b[torch.argsort(node[idx_b], stable=True)]
    if pb.numel():
        pb = pb[lds_bank_order(obin[pb], window=window, banks=banks, per=4)]
    pu = idx_u[torch.argsort(node[idx_u], stable=True)]
    seq = torch.cat((pb, pu))       # the last depositing block is topped up with the first idle events
    t_full = n // block
    nbb = min(-(-int(pb.numel()) // block), t_full)
    if nbb == 0 or nbb == t_full:
        return seq
    pos_b = (torch.arange(nbb, device=dev) * t_full) // nbb      # distinct: t_full >= nbb
    is_b = torch.zeros(t_full, dtype=torch.bool, device=dev)
    is_b[pos_b] = True
    src = torch.empty(t_full, dtype=torch.int64, device=dev)
    src[is_b] = torch.arange(nbb, device=dev)
    src[~is_b] = torch.arange(nbb, t_full, device=dev)
    body = seq[: t_full * block].view(t_full, block)[src].reshape(-1)
    return torch.cat((body, seq[t_full * block:]))


def deposit_block_order_native(obin, node, n_nodes):
    """`deposit_block_order` in one native call (`pisa_hip_deposit_block_order`, csrc/order.hip): one key per event, one
    stable radix sort, one workgroup per 4 096-event window for the bank order, the block interleave in closed form -- the
    SAME permutation as the torch formulation above (tests/test_gpu_engine.py compares them element by element), without
    its ~40 launches, four sorts and two host synchronisations per container."""
    import ctypes as C

    lib = _lib.lib()
    n = int(obin.numel())
    perm = torch.empty(n, dtype=torch.int64, device=obin.device)
    if n == 0:
        return perm
    need = int(lib.pisa_hip_deposit_block_order_workspace(n))
    if need < 0:
        raise ValueError("too many events for one container's order: %d" % n)
    work = torch.empty(need, dtype=torch.uint8, device=obin.device)
    node32 = node if node.dtype == torch.int32 else node.to(torch.int32)
    obin32 = obin if obin.dtype == torch.int32 else obin.to(torch.int32)
    _lib.check(lib.pisa_hip_deposit_block_order(C.c_void_p(node32.contiguous().data_ptr()), C.c_void_p(obin32.contiguous().data_ptr()),
                                                n, int(n_nodes), C.c_void_p(perm.data_ptr()), C.c_void_p(work.data_ptr()),
                                                need, K._stream()))
    return perm


def pack_resident_columns(perm, gx, gy, flux, aeff, w0, cols, node, obin):
    """The resident copies of one container's columns in the order `perm` and the interleaved / folded forms of the
    16-bit index layout, in ONE native launch (`pisa_hip_pack_resident_columns`, csrc/order.hip) -- element for element what
    the tensor operations of `HotPathEngine.__init__` produce (tests/test_gpu_engine.py compares them bit for bit).
    Returns (gx, gy, flux, aeff, w0, cols, node, obin, node_bin [n, 2], aeff_w0 [n, 2], static_w [n], node_bin16 [n_pad])."""
    import ctypes as C

    n = int(perm.numel())
    n_pad = -(-n // 256) * 256
    dev = perm.device
    f8 = lambda *shape: torch.empty(shape, dtype=torch.float64, device=dev)      # noqa: E731
    i4 = lambda *shape: torch.empty(shape, dtype=torch.int32, device=dev)        # noqa: E731
    o_gx, o_gy, o_flux, o_aeff, o_w0 = f8(n), f8(n), f8(n, 2), f8(n), f8(n)
    o_cols = [f8(n) for _ in cols]
    o_node, o_obin, o_nb, o_aw, o_cst, o_nb16 = i4(n), i4(n), i4(n, 2), f8(n, 2), f8(n), i4(n_pad)
    ins = [perm, gx, gy, flux, aeff, w0, node, obin] + list(cols)
    ins = [t.contiguous() for t in ins]
    ps = _lib.PackSet()
    ps.n, ps.n_pad, ps.n_sample = n, n_pad, len(cols)
    ps.d_perm, ps.d_grid_x, ps.d_grid_y, ps.d_nu_flux = (ins[k].data_ptr() for k in range(4))
    ps.d_weighted_aeff, ps.d_initial_weights, ps.d_node, ps.d_bin = (ins[k].data_ptr() for k in range(4, 8))
    ps.o_grid_x, ps.o_grid_y, ps.o_nu_flux = o_gx.data_ptr(), o_gy.data_ptr(), o_flux.data_ptr()
    ps.o_weighted_aeff, ps.o_initial_weights = o_aeff.data_ptr(), o_w0.data_ptr()
    for k in range(len(cols)):
        ps.d_sample[k], ps.o_sample[k] = ins[8 + k].data_ptr(), o_cols[k].data_ptr()
    ps.o_node, ps.o_bin, ps.o_node_bin = o_node.data_ptr(), o_obin.data_ptr(), o_nb.data_ptr()
    ps.o_aeff_w0, ps.o_static_w, ps.o_node_bin16 = o_aw.data_ptr(), o_cst.data_ptr(), o_nb16.data_ptr()
    _lib.check(_lib.lib().pisa_hip_pack_resident_columns(C.byref(ps), K._stream()))
    return o_gx, o_gy, o_flux, o_aeff, o_w0, o_cols, o_node, o_obin, o_nb, o_aw, o_cst, o_nb16


def local_slices(sizes, rank, world_size):
    """[(lo, hi)] of this rank's shard of every container (`sizes` = events per container): the
    partition `HotPathEngine` uses -- contiguous, equal to within one event, disjoint, complete"""
    return [shard_bounds(int(n), rank, world_size) for n in sizes]


LIMB_BITS, LIMB_LSB, N_LIMBS = 32, 116, 6


def limbs_to_float(limbs):
    """Host-side decoder of one accumulator (6 int64 limbs, possibly
    un-normalised sums over workgroups / ranks): value = sum limb_j*2^(32j-116),
    rounded once to fp64.  Python integers are exact; used by tests and tools --
    the product path converts on the device (`hist_finalize_kernel`)."""
    from fractions import Fraction

    total = sum(int(v) << (LIMB_BITS * j) for j, v in enumerate(limbs))
    return float(Fraction(total, 1 << LIMB_LSB))


def float_to_limbs(x):
    """exact canonical limbs of a double (|x| < 2^76, resolution 2^-116)"""
    from fractions import Fraction

    f = Fraction(float(x)) * (1 << LIMB_LSB)
    total = int(f) if f >= 0 else -int(-f)  # truncate towards zero below 2^-116
    out = []
    for _ in range(N_LIMBS - 1):
        out.append(total & 0xFFFFFFFF)
        total >>= LIMB_BITS
    out.append(total)
    return out


class PointGroups:
    """Hybrid point x event parallelism of a fit loop (round 5): the W ranks of a torch.distributed world form
    G GROUPS of R = W / G SHARDS, world rank = group * R + shard.  A group holds the whole sample -- replicated on
    its one rank (R = 1; 200 MB at 1e7 events, 288 GB per GPU) or sharded over its R ranks with the int64 limb
    all-reduce INSIDE the group only --, and the K independent points of one `eval_many` call (the n + 1 points of a
    finite-difference gradient, `analysis.py:2493-2670` with scipy's l-bfgs-b / slsqp settings; a scan) are DEALT to the
    groups in contiguous blocks.  Every point is computed entirely inside one group, by the same kernels on the same
    integer limb sums as on one GPU: per point the metric has the single-GPU bits by construction.  One all-gather of
    K doubles over the world brings every rank the whole list (the minimiser runs replicated on every rank).

    Event sharding alone (G = 1) divides a 35 us kernel and adds a 12-35 us all-reduce: a flat strong-scaling curve
    (DESIGN section 6); point groups scale the stencil instead (G groups = G times the points per sweep time)."""

    def __init__(self, world_rank, world_size, n_groups, world_group=None, make_group=None):
        if n_groups < 1 or world_size % n_groups:
            raise ValueError("%d ranks do not form %d groups of equal size" % (world_size, n_groups))
        self.world_rank, self.world_size, self.n_groups = int(world_rank), int(world_size), int(n_groups)
        self.n_shards = self.world_size // self.n_groups
        self.group_id, self.shard = divmod(self.world_rank, self.n_shards)
        self.world_group = world_group
        self.shard_group = None
        if self.n_shards > 1 and self.n_groups > 1:
            # every rank creates every sub-group, in the same order (torch.distributed's rule)
            if make_group is None:
                import torch.distributed as dist

                make_group = lambda ranks: dist.new_group(ranks=ranks)   # noqa: E731
            groups = [make_group(list(range(g * self.n_shards, (g + 1) * self.n_shards))) for g in range(self.n_groups)]
            self.shard_group = groups[self.group_id]
        elif self.n_shards > 1:
            self.shard_group = world_group      # one group: the world itself

    @property
    def topology(self):
        return "%dx%d" % (self.n_groups, self.n_shards)

    def engine_kwargs(self):
        """rank / world_size / group of the engine of this rank: its coordinates INSIDE its group"""
        return dict(rank=self.shard, world_size=self.n_shards, group=self.shard_group)

    def block(self, n_points, group=None):
        """[lo, hi): the points of a list of `n_points` that group `group` (default: this rank's) evaluates"""
        g = self.group_id if group is None else group
        return (g * n_points) // self.n_groups, ((g + 1) * n_points) // self.n_groups

    def gather(self, mine, n_points, device):
        """the K values from the blocks the groups computed: one all-gather of ceil(K / G) doubles per rank over the
        world; the values of group g are taken from its shard 0 (every shard of a group holds the same bits)"""
        import torch.distributed as dist

        per = -(-n_points // self.n_groups)
        if dist.get_backend(self.world_group) != "nccl":
            device = "cpu"      # (gloo: the CPU tests, and the one-device GPU test whose ranks exchange over gloo)
        buf = torch.full((per,), float("nan"), dtype=torch.float64, device=device)
        if mine:
            buf[:len(mine)] = torch.tensor(mine, dtype=torch.float64, device=device)
        got = [torch.empty_like(buf) for _ in range(self.world_size)]
        dist.all_gather(got, buf, group=self.world_group)
        out = []
        for g in range(self.n_groups):
            lo, hi = self.block(n_points, g)
            out += got[g * self.n_shards][:hi - lo].tolist()
        return out


_POINT_GROUPS = {"n": 0, "obj": None}


def configure_point_groups(n_groups):
    """Topology of the engines that stages build for themselves (`stages/utils/hist.py`): 0 / 1 = event sharding over the
    whole torch.distributed world (the default, north star), G > 1 = G groups of world / G ranks with the points of
    `eval_many` -- the stencil of `Analysis.fit_hypo(batched_gradient=True)` -- dealt to the groups.  Call on every rank,
    after `init_process_group`, before the first evaluation."""
    _POINT_GROUPS["n"], _POINT_GROUPS["obj"] = int(n_groups), None


def configured_points():
    """the PointGroups of this rank under `configure_point_groups`, or None"""
    if _POINT_GROUPS["n"] <= 1:
        return None
    if _POINT_GROUPS["obj"] is None:
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()):
            return None
        _POINT_GROUPS["obj"] = PointGroups(dist.get_rank(), dist.get_world_size(), _POINT_GROUPS["n"])
    return _POINT_GROUPS["obj"]


class HotPathEngine:
    """See module docstring.  `containers` is a list of dicts with keys
    name, flav, nubar, true_energy, true_coszen, nu_flux[n,2], weighted_aeff,
    initial_weights, sample (list of D host columns, already regularised),
    scale."""

    def __init__(self, containers, grid, out_binning, earth, max_layers, rank=0, world_size=1,
                 group=None, indexed=True, planned=True, packed=True, sort_events=True,
                 external_tables=False, osc_mode="grid", drop_unbinned=False, compact=False, index16=True,
                 lds_order=True, node_flux=False, block_order=True, points=None, time_setup=False):
        self.dev = K.device()
        # `time_setup`: device-synchronised wall time of the phases of this constructor (host columns -> first evaluation
        # ready) in `self.setup_ms`: upload / digitise / order / pack / plan (bench.py reports it; off: no synchronisation)
        import time as _time

        # ("upload" = what the launch stream still waits for the copies: they run beside the previous container's phases)
        self.setup_ms = {"upload": 0.0, "digitise": 0.0, "order": 0.0, "pack": 0.0, "plan": 0.0} if time_setup else None
        _t = [_time.perf_counter()]

        def _phase(name):
            if self.setup_ms is not None:
                torch.cuda.synchronize()
                now = _time.perf_counter()
                self.setup_ms[name] += 1e3 * (now - _t[0])
                _t[0] = now

        # hybrid point x event parallelism: `points` (a PointGroups) carries this rank's coordinates; the engine itself
        # only sees its group (rank = shard, world_size = shards per group) and deals the points of `eval_many`
        self.points = points
        if points is not None:
            kw = points.engine_kwargs()
            rank, world_size, group = kw["rank"], kw["world_size"], kw["group"]
        assert osc_mode in ("grid", "events")
        # flux given on the calc grid (`nu_flux_nodes` [grid.size, 2] per container) instead of per
        # event: the per-node products flux x probability become each container's own gather table
        # (`pisa_hip_flux_prob_tables`) and the events keep only their static factor
        self.node_flux = bool(node_flux)
        if self.node_flux:
            assert osc_mode == "grid" and indexed and packed and compact, \
                "node_flux needs the indexed, packed, compact event columns"
            self._own_tables = torch.empty((len(containers), grid.size, 2), dtype=torch.float64,
                                           device=self.dev)
            self._node_flux_t = []
            self._flux_tab_args = None
        self.osc_events = osc_mode == "events"
        self._event_sets, self._event_tables = [], []
        if self.osc_events:
            indexed = packed = True
            external_tables = True
        self.grid = grid
        self.out_binning = out_binning
        self.n_bins = int(np.prod([out_binning.nbins[k] for k in range(out_binning.ndim)]))
        if sort_events is True:
            # LDS accumulators (96 B per bin, <= 64 KiB) take events in any order, so
            # order them for the table gathers; larger binnings need runs of equal bin
            sort_events = "node" if self.n_bins * 96 <= 65536 else "part"
        self.rank, self.world_size, self.group = rank, world_size, group
        self._rccl = None  # direct RCCL limb all-reduce, created at the first use
        self.names = [c["name"] for c in containers]
        self._keep = []  # device tensors referenced by raw pointer
        self.indexed, self.planned = indexed, planned
        self.cont = []
        self._perm, self._flux, self._slices = [], [], []
        self._static_w, self._wflux = [], []
        self.compact = bool(compact)
        import os
        # 16-bit index form of the compact columns (20 B/event) where it applies: grid mode, calc
        # grid below 65535 nodes, output binning below 65535 bins
        index16 = (bool(index16) and self.compact
                   and packed and indexed and not self.osc_events and grid.size < 0xFFFF
                   and self.n_bins < 0xFFFF)
        self.index16 = index16
        self.n_local = 0
        shards = local_slices([len(c["true_energy"]) for c in containers], rank, world_size)
        # workgroups the fused kernel will give each container (a layout hint for the partitioned order)
        import ctypes as C

        n_arr = (C.c_int64 * len(shards))(*[hi - lo for lo, hi in shards])
        hist_wgs = (C.c_int32 * len(shards))()
        if not (torch.cuda.is_available() and _lib.lib().pisa_hip_hist_workgroups(n_arr, len(shards), hist_wgs) == 0):
            hist_wgs = [0] * len(shards)

        def column(x, sl):
            """this rank's slice of an event column -- host array or device tensor (a workload generated in
            HBM: 1e8-event samples never exist on the host) -- as an fp64 device tensor of its own"""
            if torch.is_tensor(x):
                return x[sl].to(self.dev, torch.float64).clone()
            return K.to_device(np.asarray(x, dtype=np.float64)[sl])

        def upload(c, sl):
            """this rank's slice of a container's columns in HBM (the calc-grid coordinates as (ln E, coszen): the log is
            numpy's, on the host, so that the digitisation sees the reference's values)"""
            e_col = column(c["true_energy"], sl)
            lnE = torch.log(e_col) if torch.is_tensor(c["true_energy"]) else \
                K.to_device(np.log(np.asarray(c["true_energy"], dtype=np.float64)[sl]))
            cz = column(c["true_coszen"], sl)
            flux_d = None if self.node_flux else column(c["nu_flux"], sl)
            aeff_d = column(c["weighted_aeff"], sl)
            w0_d = column(c["initial_weights"], sl)
            cols = [column(col, sl) for col in c["sample"]]
            return e_col, lnE, cz, flux_d, aeff_d, w0_d, cols

        # The columns of container i + 1 cross PCIe (a second thread, a stream of its own) while container i is digitised
        # and ordered on the launch stream: at 1e7 events 22 ms of copies beside 30 ms of ordering and packing instead of
        # in front of them (round 5, bench.py `setup_ms`).
        # (PISA_HIP_UPLOAD_THREADS=0: everything on the calling thread and the launch stream -- for counter-collecting
        # profilers, which serialise dispatches and have been seen to stall when several host threads submit work)
        import os as _os

        n_up = int(_os.environ.get("PISA_HIP_UPLOAD_THREADS", "3"))
        prefetch = torch.cuda.is_available() and len(containers) > 1 and n_up > 0
        if prefetch:
            from concurrent.futures import ThreadPoolExecutor

            # three containers ahead: the host side of an upload (slicing, numpy's log of the energies: 3-5 ms per container,
            # it releases the GIL) runs on three cores at once, the copies themselves queue on the link
            AHEAD = max(1, min(n_up, 4))
            pool = ThreadPoolExecutor(max_workers=AHEAD)
            sides = [torch.cuda.Stream(device=self.dev) for _ in range(AHEAD)]

            def upload_async(k):
                torch.cuda.set_device(self.dev)
                side = sides[k % AHEAD]
                with torch.cuda.stream(side):
                    out = upload(containers[k], slice(*shards[k]))
                    ev = torch.cuda.Event()
                    ev.record(side)
                return out, ev

            for side in sides:
                side.wait_stream(torch.cuda.current_stream())     # (columns generated in HBM: their producer has run)
            pending = [pool.submit(upload_async, k) for k in range(min(AHEAD, len(containers)))]
        # (an exception while a container is loaded must not leave the upload threads and their side streams behind)
        try:
            for ci, (c, (lo, hi)) in enumerate(zip(containers, shards)):
                sl = slice(lo, hi)
                self._slices.append((lo, hi))
                d = _lib.Container()
                d.n_events = hi - lo
                self.n_local += hi - lo
                if prefetch:
                    (e_col, lnE, cz, flux_d, aeff_d, w0_d, cols), ev = pending.pop(0).result()
                    if ci + AHEAD < len(containers):
                        pending.append(pool.submit(upload_async, ci + AHEAD))
                    main = torch.cuda.current_stream()
                    main.wait_event(ev)
                    for t in [e_col, lnE, cz, flux_d, aeff_d, w0_d] + cols:
                        if t is not None:
                            t.record_stream(main)      # allocated on the side stream, used (and freed) on this one
                else:
                    e_col, lnE, cz, flux_d, aeff_d, w0_d, cols = upload(c, sl)
                gx, gy = (lnE, cz) if grid.energy_first else (cz, lnE)
                if self.node_flux:
                    fn = c["nu_flux_nodes"]
                    fn = (fn.to(self.dev, torch.float64) if torch.is_tensor(fn)
                          else K.to_device(np.asarray(fn, dtype=np.float64))).reshape(grid.size, 2).contiguous().clone()
                    self._node_flux_t.append(fn)
                    d.d_pepmu = self._own_tables[len(self.cont)].data_ptr()
                    flux_d = torch.ones((hi - lo, 2), dtype=torch.float64, device=self.dev)
                _phase("upload")
                node = obin = perm = None
                packed_forms = None          # (nb, aw, cst, nb16) from the native pack call, where it applies
                static_w = wflux = None
                part_starts, part_width = None, 0
                if self.osc_events:
                    # event-by-event oscillation: every event is its own "node"; the shard
                    # is stored sorted by coszen so that the lanes of a wavefront cross the
                    # same number of Earth layers (3.4x faster than random order: no
                    # divergence in the layer loop)
                    e_true = e_col
                    obin = K.event_indices(cols, out_binning)
                    if sort_events and hi - lo > 1:
                        perm = torch.argsort(cz, stable=True)
                        gx, gy, flux_d, aeff_d, w0_d, e_true, cz = (
                            t[perm].contiguous() for t in (gx, gy, flux_d, aeff_d, w0_d, e_true, cz))
                        cols = [t[perm].contiguous() for t in cols]
                        obin = obin[perm].contiguous()
                    node = torch.arange(hi - lo, dtype=torch.int32, device=self.dev)
                    own = torch.zeros((hi - lo, 2), dtype=torch.float64, device=self.dev)
                    es = _lib.EventSet()
                    es.n_events, es.d_energy, es.d_coszen = hi - lo, e_true.data_ptr(), cz.data_ptr()
                    es.d_probability, es.d_pepmu = None, own.data_ptr()
                    es.nubar, es.flav = int(c["nubar"]), int(c["flav"])
                    self._event_sets.append(es)
                    self._event_tables.append((e_true, cz, own))   # resident (coszen-sorted) order
                    self._keep += [e_true, cz, own]
                    d.d_pepmu = own.data_ptr()
                elif indexed:
                    # coordinates never change between evaluations: digitise once
                    node = K.event_indices([gx, gy], grid.binning)
                    obin = K.event_indices(cols, out_binning)
                    _phase("digitise")
                    if sort_events and hi - lo > 1:
                        # Event order inside a container is arbitrary and the exact
                        # accumulation makes the result independent of it, so the
                        # shard is stored sorted by calc-grid node: the (P_e, P_mu)
                        # gathers of a wavefront then hit a handful of cache lines
                        # instead of 64 different ones.
                        if sort_events == "bin":
                            perm = bin_window_order(obin, self.n_bins)
                        elif sort_events == "part":
                            # binning beyond the LDS accumulators: partitions = the kernel's LDS windows
                            width = _lib.lib().pisa_hip_hist_window_bins(self.n_bins) if index16 else 0
                            res = None
                            if width > 0 and lds_order and block_order and not drop_unbinned:
                                res = window_partition_order_native(obin, node, self.n_bins, width, grid.size,
                                                                    n_wg=hist_wgs[len(self.cont)])
                            if res is not None:
                                perm, part_starts = res
                                part_width = width
                            else:
                                perm = bin_partition_order(obin, node, self.n_bins, width=width if width > 0 else 672)
                        else:
                            perm = torch.argsort(node, stable=True)
                    blocked = (index16 and lds_order and block_order and perm is not None and sort_events == "node"
                               and self.n_bins * 96 <= 65536)
                    if blocked:
                        # (PISA_HIP_TORCH_ORDER=1: the torch formulation, the specification the native call is tested against)
                        perm = deposit_block_order(obin, node, window=4096, banks=32) \
                            if _os.environ.get("PISA_HIP_TORCH_ORDER") == "1" else deposit_block_order_native(obin, node, grid.size)
                    elif part_starts is not None:
                        pass                      # (bank order applied inside the partitions)
                    elif lds_order and perm is not None and (sort_events == "part" or (
                            sort_events != "bin" and self.n_bins * 96 <= 65536)):
                        perm = perm[lds_bank_order(obin[perm], window=4096, banks=32,
                                                   per=4 if index16 else 2)]
                    _phase("order")
                    if drop_unbinned:
                        # an event outside the output binning (or outside the calc grid: P = 0)
                        # adds nothing to any map, whatever the parameters: the coordinates are
                        # static, so such events need not be resident at all
                        keep = (obin >= 0) & (node >= 0)
                        perm = torch.nonzero(keep).reshape(-1) if perm is None else perm[keep[perm]]
                    # the columns in resident order: one native launch for the 16-bit index layout (round 5: the ~25 tensor
                    # operations per container below were 6 ms of a 24 ms set-up at 1e7 events, bound by their dispatch on the
                    # host; PISA_HIP_TORCH_ORDER=1 keeps them: they are the specification the call is tested against)
                    native_pack = (perm is not None and packed and compact and index16 and len(cols) <= 3
                                   and perm.dtype == torch.int64 and node.dtype == torch.int32 and obin.dtype == torch.int32
                                   and flux_d.dim() == 2 and flux_d.shape[1] == 2
                                   and all(t.dtype == torch.float64 for t in [gx, gy, flux_d, aeff_d, w0_d] + cols)
                                   and _os.environ.get("PISA_HIP_TORCH_ORDER") != "1")
                    if native_pack:
                        (gx, gy, flux_d, aeff_d, w0_d, cols, node, obin, *packed_forms) = pack_resident_columns(
                            perm, gx, gy, flux_d, aeff_d, w0_d, cols, node, obin)
                        self.n_local += int(perm.numel()) - d.n_events
                        d.n_events = int(perm.numel())
                    elif perm is not None:
                        gx, gy, flux_d, aeff_d, w0_d = (t[perm].contiguous() for t in
                                                       (gx, gy, flux_d, aeff_d, w0_d))
                        cols = [t[perm].contiguous() for t in cols]
                        node, obin = node[perm].contiguous(), obin[perm].contiguous()
                        self.n_local += int(perm.numel()) - d.n_events
                        d.n_events = int(perm.numel())
                if not indexed and not self.osc_events and sort_events and hi - lo > 1 and not self.node_flux:
                    # coordinate form (both digitisations in the kernel, 72 B/event): the resident ORDER is still
                    # free, so the events are stored sorted by the calc-grid node they will fall on -- digitised here once, for
                    # the order only, the indices are not kept -- and the kernel's 16-byte table gathers of a wavefront stay
                    # inside a few cache lines (random order: 11 % more HBM traffic than the columns, round-4 PMC pass)
                    perm = torch.argsort(K.event_indices([gx, gy], grid.binning), stable=True)
                    gx, gy, flux_d, aeff_d, w0_d = (t[perm].contiguous() for t in (gx, gy, flux_d, aeff_d, w0_d))
                    cols = [t[perm].contiguous() for t in cols]
                    _phase("order")
                self._keep += [gx, gy, flux_d, aeff_d, w0_d] + cols
                self._perm.append(perm)
                self._flux.append(flux_d)
                d.d_grid_x, d.d_grid_y = gx.data_ptr(), gy.data_ptr()
                d.d_nu_flux = flux_d.data_ptr()
                d.d_weighted_aeff, d.d_initial_weights = aeff_d.data_ptr(), w0_d.data_ptr()
                for k, col in enumerate(cols):
                    d.d_sample[k] = col.data_ptr()
                if indexed:
                    self._keep += [node, obin]
                    d.d_node, d.d_bin = node.data_ptr(), obin.data_ptr()
                    if packed:
                        # interleaved columns: every load of the fused kernel is 16 bytes
                        if packed_forms is not None:
                            nb, aw = packed_forms[0], packed_forms[1]
                        else:
                            nb = torch.stack([node, obin], dim=1).contiguous()
                            aw = torch.stack([aeff_d, w0_d], dim=1).contiguous()
                        self._keep += [nb, aw]
                        d.d_node_bin, d.d_aeff_w0 = nb.data_ptr(), aw.data_ptr()
                        if compact:
                            # the factors of the weight that no oscillation parameter touches,
                            # multiplied once: (w0*aeff) * (f_e, f_mu)
                            cst = packed_forms[2] if packed_forms is not None else (w0_d * aeff_d).contiguous()
                            self._keep.append(cst)
                            static_w = cst
                            if index16:
                                # 20 B/event: both indices in 16 bits, and the flux pairs of a quad of
                                # events stored lane-contiguously ([quad / 64][4][quad % 64]) so that
                                # every load of the kernel is 16 contiguous bytes per lane; columns
                                # padded to whole blocks of 64 quads with events outside the binning
                                n_ev = int(node.numel())
                                n_pad = -(-n_ev // 256) * 256
                                if packed_forms is not None:
                                    nb16 = packed_forms[3]
                                else:
                                    v = torch.full((n_pad,), 0xFFFFFFFF, dtype=torch.int64, device=self.dev)
                                    v[:n_ev] = (torch.where(node < 0, 0xFFFF, node.long())
                                                | (torch.where(obin < 0, 0xFFFF, obin.long()) << 16))
                                    nb16 = torch.where(v >= 2 ** 31, v - 2 ** 32, v).to(torch.int32).contiguous()
                                wq = torch.zeros((n_pad // 256, 4, 64, 2), dtype=torch.float64, device=self.dev)
                                self._keep += [nb16, wq]
                                d.d_node_bin16, d.d_weighted_flux_q = nb16.data_ptr(), wq.data_ptr()
                                if part_starts is not None:
                                    ps = torch.tensor(part_starts, dtype=torch.int32, device=self.dev)
                                    assert int(ps[-1]) * 256 == n_pad
                                    self._keep.append(ps)
                                    d.d_part_start, d.n_part, d.part_width = ps.data_ptr(), len(part_starts) - 1, part_width
                                wflux = wq
                                self._fill_wflux(wq, cst, flux_d)
                            else:
                                # 24 B/event with node_bin
                                wf = (cst[:, None] * flux_d).contiguous()
                                self._keep.append(wf)
                                d.d_weighted_flux = wf.data_ptr()
                                wflux = wf
                _phase("pack")
                d.flav, d.nubar, d.scale = int(c["flav"]), int(c["nubar"]), float(c["scale"])
                self._static_w.append(static_w)
                self._wflux.append(wflux)
                self.cont.append(d)
            self._cont_arr = (_lib.Container * len(self.cont))(*self.cont)
            self._cont_gen = 0      # bumped by every writer of `_cont_arr` (set_scale, containers_changed): part of the evaluator's key
            # Earth layers for the coszen nodes (prob3.setup_function, prob3.py:398-409)
            self.earth = earth
        finally:
            if prefetch:
                pool.shutdown(wait=True, cancel_futures=True)
        self.prob_nu = self.prob_nubar = self.pepmu = self.plan = None
        if not external_tables:
            self.energy_d = K.to_device(grid.energy)
            _, self.dens_d, self.dist_d = K.calc_layers(earth, K.to_device(grid.coszen), max_layers)
            self.prob_nu = torch.empty((grid.size, 3, 3), dtype=torch.float64, device=self.dev)
            self.prob_nubar = torch.empty((grid.size, 3, 3), dtype=torch.float64, device=self.dev)
            self.pepmu = torch.empty((2, 3, grid.size, 2), dtype=torch.float64, device=self.dev)
            self.plan = K.GridPlan(self.dens_d, self.dist_d) if planned else None
        self._event_arr = (_lib.EventSet * len(self._event_sets))(*self._event_sets) \
            if self._event_sets else None
        self.ws = K.HistWorkspace(len(self.cont), self.n_bins, self.dev)
        _phase("plan")
        if self.setup_ms is not None:
            self.setup_ms["total"] = sum(self.setup_ms.values())
        self.metric_out = torch.zeros(1, dtype=torch.float64, device=self.dev)
        self.metric_status = torch.zeros(1, dtype=torch.int32, device=self.dev)
        # [0]: the metric as the tail kernel leaves it; [0:4]: the four partial sums of its split form
        # (`pisa_hip_finalize_metric_parts`, 16 workgroups: joined along the kernel's own reduction tree, bit for bit)
        self.metric_host = torch.zeros(16, dtype=torch.float64).pin_memory()
        self._metric_host_np = self.metric_host.numpy()
        self.split_tail = True     # four tail workgroups per point (an attribute, not an environment switch)
        self.spin_wait = 50000  # polls of the pinned result (~7 ms) before falling back to a stream sync
        self.fused_tail = True
        self._limbs_zero = self._maps_valid = False
        self._lean = None
        self._evaluator = None     # pisa_hip_evaluator of the standard shape (`_evaluator_for`)
        self.one_call = True       # eval_host through it; False: the three separate C-ABI calls (tests compare both)
        self.data = None
        self._data_src = None
        self._out_block = None   # weak reference to the device-backed maps handed out last

    def _up(self, a):
        t = K.to_device(a)
        self._keep.append(t)
        return t

    def update_flux(self, i, flux):
        """new nu_flux column for container i (flux systematics changed): `flux` [n, 2] device
        tensor in the container's own event order.  One kernel gathers it into this engine's
        resident order and folds the static factors in (`pisa_hip_fold_flux`)."""
        lo, hi = self._slices[i]
        f = flux[lo:hi] if (lo != 0 or hi != flux.shape[0]) else flux
        f = f.contiguous()
        perm = self._perm[i]
        if self._wflux[i] is not None:
            out = self._wflux[i]
            _lib.check(_lib.lib().pisa_hip_fold_flux(
                K._ptr(f), None if perm is None else K._ptr(perm), K._ptr(self._static_w[i]),
                int(self._static_w[i].numel()), 0 if out.dim() == 2 else 1, K._ptr(out), K._stream()))
            # (the separate resident nu_flux column is not read by the compact kernels)
        else:
            self._flux[i].copy_(f if perm is None else f[perm])

    def update_flux_many(self, items):
        """`update_flux` for several containers -- `items` = [(i, flux tensor), ...] -- in ONE launch
        (`pisa_hip_fold_flux_multi`); the argument block is reused while the same tensors come back
        (a flux stage that rewrites its arrays in place)."""
        if not items:
            return
        if any(self._wflux[i] is None for i, _ in items):
            for i, f in items:
                self.update_flux(i, f)
            return
        prepared = []
        for i, flux in items:
            lo, hi = self._slices[i]
            f = flux[lo:hi] if (lo != 0 or hi != flux.shape[0]) else flux
            prepared.append((i, f.contiguous()))
        key = tuple((i, f.data_ptr()) for i, f in prepared)
        blk = getattr(self, "_fold_block", None)
        if blk is None or blk["key"] != key:
            arr = (_lib.FoldSet * len(prepared))()
            for d, (i, f) in zip(arr, prepared):
                out, perm = self._wflux[i], self._perm[i]
                d.n = int(self._static_w[i].numel())
                d.d_flux, d.d_static_w, d.d_out = f.data_ptr(), self._static_w[i].data_ptr(), out.data_ptr()
                d.d_perm = None if perm is None else perm.data_ptr()
                d.layout = 0 if out.dim() == 2 else 1
            blk = self._fold_block = dict(key=key, arr=arr, keep=[f for _, f in prepared])
        _lib.check(_lib.lib().pisa_hip_fold_flux_multi(blk["arr"], len(blk["arr"]), K._stream()))

    def enable_barr(self, columns):
        """Prepare the ONE-pass refresh of the folded flux columns for `flux.barr_simple` systematics
        (`update_flux_barr`).  `columns`: per container (true_energy[n], true_coszen[n],
        nu_flux_nominal[n, 2], nubar_flux_nominal[n, 2]) device tensors in the container's own event
        order.  The engine keeps copies in its resident order AND column layout (for the 20 B form the
        quad-blocked one), with the event's parameter-free factors of apply_sys_vectorized
        (`pisa_hip_barr_factors`: ten of its eleven transcendentals), so that a moved systematic costs one
        elementwise pass that writes the folded column directly -- no Barr output in container order, no
        gather into the resident order."""
        assert all(w is not None for w in self._wflux), "needs the compact (folded) event columns"
        lib = _lib.lib()
        arr = (_lib.BarrFoldSet * len(self.cont))()
        keep = []
        status = torch.zeros(1, dtype=torch.int32, device=self.dev)
        for i, (e, cz, nu, nub) in enumerate(columns):
            lo, hi = self._slices[i]
            perm, out, sw = self._perm[i], self._wflux[i], self._static_w[i]
            n_ev = int(sw.numel())
            if out.dim() == 2:       # plain [n][2] column: position = resident event
                n_pos = n_ev
                ev = torch.arange(n_pos, device=self.dev)
            else:                    # quad-blocked: event 4q + k at [q / 64][k][q % 64]
                n_pos = out.shape[0] * 256
                pos = torch.arange(n_pos, device=self.dev)
                q = (pos // 256) * 64 + pos % 64
                ev = 4 * q + (pos // 64) % 4
            valid = ev < n_ev
            evc = torch.where(valid, ev, torch.zeros_like(ev))
            src = evc if perm is None else perm[evc]

            def take(col, fill):
                t = col[lo:hi][src]
                t[~valid] = fill
                return t.contiguous()

            e_b, cz_b = take(e, 1.0), take(cz, 0.0)
            nu_b, nub_b = take(nu, 0.0), take(nub, 0.0)
            w_b = torch.where(valid, sw[evc], torch.zeros_like(sw[evc])).contiguous()
            fac = torch.empty((5, n_pos), dtype=torch.float64, device=self.dev)
            _lib.check(lib.pisa_hip_barr_factors(K._ptr(e_b), K._ptr(cz_b), n_pos, K._ptr(fac), K._ptr(status),
                                                 K._stream()))
            d = arr[i]
            d.n = n_pos
            d.d_nu_flux_nominal, d.d_nubar_flux_nominal = nu_b.data_ptr(), nub_b.data_ptr()
            d.d_factors, d.d_static_w, d.d_out = fac.data_ptr(), w_b.data_ptr(), out.data_ptr()
            d.nubar = int(self.cont[i].nubar)
            keep += [nu_b, nub_b, fac, w_b]
        if int(status.item()) != 0:
            raise ValueError("true_energy must be positive for the one-pass flux refresh")
        self._barr = dict(arr=arr, keep=keep, fn=lib.pisa_hip_barr_fold_multi)

    def update_flux_barr(self, nue_numu_ratio, nu_nubar_ratio, delta_index, Barr_uphor_ratio, Barr_nu_nubar_ratio):
        """new folded flux columns of ALL containers for these `flux.barr_simple` parameter values
        (barr_simple.py:83-104 + the fold of `update_flux`): one launch, one pass, the same bits"""
        b = self._barr
        _lib.check(b["fn"](b["arr"], len(b["arr"]), float(nue_numu_ratio), float(nu_nubar_ratio), float(delta_index),
                           float(Barr_uphor_ratio), float(Barr_nu_nubar_ratio), K._stream()))

    def update_flux_nodes(self, i, flux_nodes):
        """node_flux mode: new [grid.size, 2] flux of container i on the calc grid (device tensor).
        A contiguous fp64 tensor on this device is adopted as it is -- the table kernel reads it
        through its pointer at the next evaluation, so a flux stage that rewrites its own array in
        place costs nothing here -- anything else is copied."""
        t = flux_nodes.reshape(self.grid.size, 2)
        cur = self._node_flux_t[i]
        if t.dtype == torch.float64 and t.device == cur.device and t.is_contiguous():
            if t.data_ptr() != cur.data_ptr():
                self._node_flux_t[i] = t
                if self._flux_tab_args is not None:
                    self._flux_tab_args["ptrs"][i] = t.data_ptr()
        else:
            cur.copy_(t)

    def _flux_tables(self, pepmu):
        """per-container gather tables flux x probability from the shared (P_e, P_mu) tables"""
        import ctypes as C

        a = self._flux_tab_args
        if a is None:
            n = len(self.cont)
            a = self._flux_tab_args = dict(
                ptrs=(C.c_void_p * n)(*[t.data_ptr() for t in self._node_flux_t]),
                nubar=(C.c_int32 * n)(*[int(d.nubar) for d in self.cont]),
                flav=(C.c_int32 * n)(*[int(d.flav) for d in self.cont]), n=n,
                out=C.c_void_p(self._own_tables.data_ptr()), fn=_lib.lib().pisa_hip_flux_prob_tables)
        return a["fn"](a["ptrs"], a["nubar"], a["flav"], a["n"], C.c_void_p(pepmu.data_ptr()),
                       self.grid.size, a["out"], K._stream())

    @staticmethod
    def _fill_wflux(out, static_w, flux):
        """static_w * (f_e, f_mu) into the plain [n][2] column or the quad-blocked one (set-up)"""
        if out.dim() == 2:
            torch.mul(static_w[:, None], flux, out=out)
        else:
            n = flux.shape[0]
            tmp = torch.zeros((out.shape[0] * 256, 2), dtype=torch.float64, device=out.device)
            torch.mul(static_w[:, None], flux, out=tmp[:n])
            out.permute(0, 2, 1, 3).copy_(tmp.view(out.shape[0], 64, 4, 2))

    _evaluator = None

    def set_scale(self, name, scale):
        i = self.names.index(name)
        self.cont[i].scale = float(scale)
        self._cont_arr[i].scale = float(scale)
        ev = self._evaluator
        if ev is not None:
            # the evaluator's snapshot moves with the array: same generation on both sides, no new evaluator
            _lib.check(_lib.lib().pisa_hip_evaluator_set_scale(ev["handle"], i, float(scale)))

    def containers_changed(self):
        """to be called by whoever writes a field of `_cont_arr` other than through `set_scale` (a pointer swapped for
        another column, an event count): the next evaluation makes a new evaluator from the array as it then is.  (Round 5
        hashed the array's bytes on every evaluation instead -- a copy and a hash of 2.4 KB on a 67 us step.)"""
        self._cont_gen += 1

    # -- one evaluation per C-ABI call (`pisa_hip_evaluator_*`) --------------
    def _evaluator_for(self):
        """the evaluator of this engine's standard shape (planned grid oscillation, indexed columns, fused
        tail, the value polled from pinned memory), made at the first use and again when one of the
        buffers it was made from has been replaced; None where the shape does not apply (flux on the
        grid nodes: a launch of its own between prob3 and the accumulation; several ranks without the
        direct RCCL communicator: the all-reduce is torch.distributed's)"""
        import ctypes as C

        if self.node_flux or not self.spin_wait:
            return None
        # fast path (this sits between the LLH read-back and the first launch of the next point): the very tensor OBJECTS
        # the evaluator was made from are still the engine's (the evaluator keeps references, so an identity cannot be
        # recycled) and nobody has written the container array -- a dozen `data_ptr()` calls (~0.3 us each) otherwise
        ev = self._evaluator
        if ev is not None:
            o = ev["objects"]
            ws = self.ws
            if (o[0] is self.pepmu and o[1] is ws.limbs and o[2] is ws.hist and o[3] is ws.sumw2 and o[4] is ws.status
                    and o[5] is self.metric_status and o[6] is self.metric_host and o[7] is self.plan and o[8] is self.energy_d
                    and o[9] is self._rccl and ev["gen"] == self._cont_gen and ev["world"] == self.world_size):
                return ev
        fn = comm = None
        if self.world_size > 1:
            if self._rccl is None:
                self.allreduce_setup()
            if not self._rccl:
                return None
            fn = C.cast(self._rccl.lib.ncclAllReduce, C.c_void_p)
            comm = self._rccl.comm
        # every buffer the evaluator was made from, and the generation of the container array it SNAPSHOTS
        # (pisa_hip_evaluator_create copies it; `set_scale` updates array and copy together, any other writer calls
        # `containers_changed`): a changed array makes a new evaluator instead of running on stale pointers
        key = (self.pepmu.data_ptr(), self.ws.limbs.data_ptr(), self.ws.hist.data_ptr(), self.ws.sumw2.data_ptr(),
               self.ws.status.data_ptr(), self.metric_status.data_ptr(), self.metric_host.data_ptr(),
               self.plan.handle.value if hasattr(self.plan.handle, "value") else self.plan.handle, self.energy_d.data_ptr(),
               None if comm is None else comm.value, self.world_size, self._cont_gen)
        ev = self._evaluator
        if ev is not None and ev["key"] == key:
            return ev
        if ev is not None:
            _lib.lib().pisa_hip_evaluator_destroy(ev["handle"])
            self._evaluator = None
        d = _lib.EvaluatorDesc()
        d.h_containers = C.cast(self._cont_arr, C.c_void_p)
        d.n_containers, d.n_e, d.e_major = len(self._cont_arr), self.energy_d.numel(), 1 if self.grid.energy_first else 0
        d.h_calc_grid = C.cast(C.pointer(self.grid.binning), C.c_void_p)
        d.h_out_binning = C.cast(C.pointer(self.out_binning), C.c_void_p)
        d.plan = self.plan.handle
        d.d_energy, d.d_pepmu = self.energy_d.data_ptr(), self.pepmu.data_ptr()
        d.d_limbs, d.d_hist, d.d_sumw2 = self.ws.limbs.data_ptr(), self.ws.hist.data_ptr(), self.ws.sumw2.data_ptr()
        d.partial = self.metric_host.data_ptr()
        d.d_status, d.d_metric_status = self.ws.status.data_ptr(), self.metric_status.data_ptr()
        d.allreduce, d.comm = fn, comm
        h = C.c_void_p()
        _lib.check(_lib.lib().pisa_hip_evaluator_create(C.byref(d), C.byref(h)))
        value = C.c_double()
        ev = self._evaluator = dict(key=key, handle=h, value=value, value_ref=C.byref(value), gen=self._cont_gen, world=self.world_size,
                                    objects=(self.pepmu, self.ws.limbs, self.ws.hist, self.ws.sumw2, self.ws.status, self.metric_status,
                                             self.metric_host, self.plan, self.energy_d, self._rccl),
                                    call=_lib.lib().pisa_hip_evaluator_eval, keep=(fn, comm))
        return ev

    def _eval_one_call(self, ev, params, kind):
        """prob3 -> accumulate -> [all-reduce] -> tail, enqueued AND awaited inside one C-ABI call"""
        self._release_outputs()
        rc = ev["call"](ev["handle"], params, K.METRIC_KIND[kind], self.data.data_ptr(), 1 if self._limbs_zero else 0,
                        20000, ev["value_ref"], K._stream())
        self._limbs_zero = self._maps_valid = rc == 0
        if rc:
            _lib.check(rc)
        return ev["value"].value

    # -- per-eval steps ----------------------------------------------------
    def compute_probs(self, params):
        if self.osc_events:
            K.prob3_events_multi(params, self.earth, self._event_arr, self.ws.status)
        elif self.plan is not None:
            K.prob3_grid_planned(params, self.plan, self.energy_d, e_major=self.grid.energy_first,
                                 out_nu=self.prob_nu, out_nubar=self.prob_nubar,
                                 out_pepmu=self.pepmu)
        else:
            K.prob3_grid(params, self.energy_d, self.dens_d, self.dist_d,
                         e_major=self.grid.energy_first, out_nu=self.prob_nu,
                         out_nubar=self.prob_nubar, out_pepmu=self.pepmu)

    def _release_outputs(self):
        """device-backed maps of the previous evaluation that are still referenced are brought
        to the host before the next launch overwrites them"""
        if self._out_block is not None:
            b = self._out_block()
            if b is not None:
                b.detach()
            self._out_block = None

    def accumulate(self, params=None):
        self._release_outputs()
        if params is not None:
            self.compute_probs(params)
        if self.node_flux:
            _lib.check(self._flux_tables(self.pepmu))
        K.reweight_hist(self._cont_arr, self.grid.binning, self.prob_nu, self.prob_nubar,
                        self.pepmu if (self.indexed and not self.osc_events) else None,
                        self.out_binning, self.ws, clear=not self._limbs_zero)
        self._limbs_zero = self._maps_valid = False

    def allreduce(self):
        """int64 SUM of the limbs over the ranks: RCCL called directly on the launch stream where
        the group runs on RCCL (`pisa_amd/rccl.py`; all ranks agree on it at the first call),
        `torch.distributed` otherwise (gloo in the CPU tests, or PISA_HIP_DIRECT_RCCL=0)."""
        if self.world_size <= 1:
            return
        if self._rccl is None:
            self.allreduce_setup()
        if self._rccl:
            self._rccl.all_reduce_(self.ws.limbs, K._stream())
        else:
            allreduce_limbs(self.ws.limbs, self.world_size, self.group)

    def allreduce_setup(self):
        """direct RCCL communicator where the group runs on RCCL (all ranks agree), else False"""
        import os

        from . import rccl

        self._rccl = False
        if int(os.environ.get("PISA_HIP_DIRECT_RCCL", "1")):
            self._rccl = rccl.LimbAllReduce.create(self.dev, self.group) or False

    def __del__(self):
        ev = getattr(self, "_evaluator", None)
        if ev is not None:
            try:
                _lib.lib().pisa_hip_evaluator_destroy(ev["handle"])
            except Exception:  # interpreter shutdown
                pass
            self._evaluator = None

    def close(self):
        """release the direct RCCL communicator (before the process group is destroyed)"""
        if getattr(self, "_evaluator", None) is not None:
            _lib.lib().pisa_hip_evaluator_destroy(self._evaluator["handle"])
            self._evaluator = None
        if self._rccl:
            self._rccl.destroy()
        self._rccl = None

    def finalize(self):
        """limbs -> fp64 maps (no-op if the fused tail of `eval` already wrote them)"""
        if not self._maps_valid:
            K.hist_finalize(self.ws)
            self._maps_valid = True
        return self.ws.hist, self.ws.sumw2

    def set_data(self, data_hist):
        self.data = K.to_device(np.asarray(data_hist, dtype=np.float64).ravel())
        self._data_src = None

    def set_data_cached(self, data_hist):
        """`set_data` unless `data_hist` holds what was uploaded last (a fit loop compares every
        template with the same data map)"""
        src = self._data_src
        if src is not None and src.shape == data_hist.shape and np.array_equal(src, data_hist):
            return
        self.set_data(data_hist)
        self._data_src = np.array(data_hist, dtype=np.float64, copy=True)

    def upload_extra(self, extra):
        """[2, n_bins] host array (expectation and variance added to the template) -> device tensor;
        re-uploaded only when the values changed"""
        src = getattr(self, "_extra_src", None)
        if src is None or src.shape != extra.shape or not np.array_equal(src, extra):
            self._extra_d = K.to_device(np.ascontiguousarray(extra, dtype=np.float64))
            self._extra_src = np.array(extra, dtype=np.float64, copy=True)
        return self._extra_d

    # -- two-phase evaluation for callers that hand out maps before a metric is asked for ------
    def front(self, tables=None):
        """phase A: fused lookup + reweight + histogram (+ all-reduce) with the probability
        tables of the caller; asynchronous.  Device-backed maps of the previous evaluation that
        are still referenced are brought to the host first (the launch overwrites them)."""
        self._release_outputs()
        import ctypes as C

        a = self._lean
        if a is None or a.get("kind") != "front" or a.get("front_pepmu") is not tables:
            lib = _lib.lib()
            pep = tables if tables is not None else self.pepmu
            a = self._lean = dict(
                kind="front", front_pepmu=tables, tabs=None, lib=lib, cont=self._cont_arr, n_cont=len(self._cont_arr),
                nu=C.c_void_p(self.prob_nu.data_ptr()) if self.prob_nu is not None else None,
                nubar=C.c_void_p(self.prob_nubar.data_ptr()) if self.prob_nubar is not None else None,
                pepmu=C.c_void_p(pep.data_ptr()), grid=C.byref(self.grid.binning),
                outb=C.byref(self.out_binning), limbs=C.c_void_p(self.ws.limbs.data_ptr()),
                status=C.c_void_p(self.ws.status.data_ptr()), hist=C.c_void_p(self.ws.hist.data_ptr()),
                sumw2=C.c_void_p(self.ws.sumw2.data_ptr()), data=None, data_t=None,
                out=C.c_void_p(self.metric_host.data_ptr()),
                mstatus=C.c_void_p(self.metric_status.data_ptr()), keep=pep)
        lib = a["lib"]
        if self.node_flux:
            _lib.check(self._flux_tables(a["keep"]))
        fn = lib.pisa_hip_reweight_hist_acc if self._limbs_zero else lib.pisa_hip_reweight_hist
        rc = fn(a["cont"], a["n_cont"], a["grid"], a["nu"], a["nubar"], a["pepmu"], a["outb"],
                a["limbs"], a["status"], K._stream())
        self._limbs_zero = self._maps_valid = False
        _lib.check(rc)
        self.allreduce()

    def can_fuse_tail(self):
        return (self.fused_tail and not self._maps_valid
                and len(self.cont) * self.n_bins <= K.FINALIZE_METRIC_MAX)

    def _split_ok(self, kind):
        """the tail in its four-workgroup form (partial sums joined here): whenever the value is polled from
        pinned memory anyway; not for chi2 (its all-bins-equal rule, stats.py:160-161, needs every bin)"""
        return self.split_tail and self.spin_wait > 0 and kind != "chi2"

    TAIL_PARTS = 16   # workgroups of the split tail (`pisa_hip_finalize_metric_parts`)

    @staticmethod
    def _join_parts(p):
        """the metric kernel's reduction tree, its last levels: p[i] += p[i + w] for w = n/2 ... 1"""
        p = [float(v) for v in p]
        w = len(p) // 2
        while w >= 1:
            for i in range(w):
                p[i] = p[i] + p[i + w]
            w //= 2
        return p[0]

    def _poll_split(self):
        """the split tail's partial sums joined as the kernel's own tree joins them, as soon as all have arrived"""
        h = self._metric_host_np
        n = self.TAIL_PARTS
        hn = h[:n]
        add = np.add.reduce
        for _ in range(self.spin_wait):
            v = add(hn)          # NaN while one of them is missing (one numpy call per poll)
            if v == v:
                return self._join_parts(hn)
        torch.cuda.current_stream().synchronize()
        return self._join_parts(hn)

    def tail_host(self, kind, scale=None, extra=None):
        """phase B: maps + metric against `self.data` of the accumulated limbs, value on the host.
        `scale` [n_cont, n_bins] / `extra` [2, n_bins] (device tensors): per-bin factors of a stage after
        the histogram and maps of other pipelines added to the template
        (`pisa_hip_finalize_metric_scaled`); only with the fused tail (`can_fuse_tail()`)."""
        if self.can_fuse_tail():
            import ctypes as C

            a = self._lean
            if a is None:   # first use without `front` / `eval_host` having built the argument block
                lib = _lib.lib()
                a = object()   # matches no table: `front` / `_lean_eval` build their full block
                a = self._lean = dict(
                    kind="tail", front_pepmu=a, tabs=a, lib=lib, cont=self._cont_arr, n_cont=len(self._cont_arr),
                    limbs=C.c_void_p(self.ws.limbs.data_ptr()), status=C.c_void_p(self.ws.status.data_ptr()),
                    hist=C.c_void_p(self.ws.hist.data_ptr()), sumw2=C.c_void_p(self.ws.sumw2.data_ptr()),
                    data=None, data_t=None, out=C.c_void_p(self.metric_host.data_ptr()),
                    mstatus=C.c_void_p(self.metric_status.data_ptr()))
            if a["data_t"] is not self.data:
                a["data"], a["data_t"] = C.c_void_p(self.data.data_ptr()), self.data
            h = self._metric_host_np
            if self._split_ok(kind):
                h[:] = np.nan
                rc = a["lib"].pisa_hip_finalize_metric_parts(
                    a["limbs"], 1, a["n_cont"], self.n_bins, a["hist"], a["sumw2"], K.METRIC_KIND[kind], a["data"],
                    None if scale is None else C.c_void_p(scale.data_ptr()), 0,
                    None if extra is None else C.c_void_p(extra.data_ptr()),
                    a["out"], self.TAIL_PARTS, a["status"], a["mstatus"], 1, K._stream())
                self._limbs_zero = self._maps_valid = rc == 0
                _lib.check(rc)
                return self._poll_split()
            h[0] = np.nan
            if scale is None and extra is None:
                rc = a["lib"].pisa_hip_finalize_metric(a["limbs"], a["n_cont"], self.n_bins, a["hist"],
                                                       a["sumw2"], K.METRIC_KIND[kind], a["data"], a["out"],
                                                       a["status"], a["mstatus"], 1, K._stream())
            else:
                rc = a["lib"].pisa_hip_finalize_metric_scaled(
                    a["limbs"], a["n_cont"], self.n_bins, a["hist"], a["sumw2"], K.METRIC_KIND[kind], a["data"],
                    None if scale is None else C.c_void_p(scale.data_ptr()),
                    None if extra is None else C.c_void_p(extra.data_ptr()),
                    a["out"], a["status"], a["mstatus"], 1, K._stream())
            self._limbs_zero = self._maps_valid = rc == 0
            _lib.check(rc)
            for _ in range(self.spin_wait):
                v = h[0]
                if v == v:
                    return float(v)
        else:
            assert scale is None and extra is None, "scaled tail needs the fused tail kernel"
            self._tail(kind, self.metric_host)
        torch.cuda.current_stream().synchronize()
        return float(self.metric_host[0])

    def metric(self, kind="llh"):
        return K.metric(kind, self.data, self.ws.hist, self.ws.sumw2, total_out=self.metric_out,
                        status=self.metric_status)

    def _tail(self, kind, out):
        """maps + metric of the accumulated (and all-reduced) limbs into `out`.
        One launch (`pisa_hip_finalize_metric`, which also leaves the limbs zeroed
        for the next accumulate) when a single workgroup can hold the maps; the
        separate finalize and metric kernels otherwise.  Same bits either way."""
        if (self.fused_tail and not self._maps_valid
                and len(self.cont) * self.n_bins <= K.FINALIZE_METRIC_MAX):
            K.finalize_metric(self.ws, kind, self.data, out, self.metric_status, clear_limbs=True)
            self._limbs_zero = self._maps_valid = True
        else:
            self.finalize()
            K.metric(kind, self.data, self.ws.hist, self.ws.sumw2, total_out=out,
                     status=self.metric_status)
        return out

    def eval(self, params, kind="llh"):
        """One template evaluation + metric; returns a 1-element device tensor."""
        self.accumulate(params)
        self.allreduce()
        if self.data is None:
            self.finalize()
            return None
        return self._tail(kind, self.metric_out)

    def eval_host(self, params, kind="llh"):
        """`eval` for a fit loop that needs the number on the host: the tail kernel
        writes the metric into pinned, device-mapped host memory, so the host only
        waits for the stream (no device->host copy operation).  Returns a float.

        For the standard shape (planned grid, packed columns, fused tail) the four
        launches go through `_lean_eval`: the same C-ABI calls as the generic
        methods, minus their per-call Python (tensor -> pointer conversions,
        contiguity asserts, keyword handling) -- this sits between the LLH
        read-back and the first kernel of the next point."""
        if (self.plan is not None and self.indexed and not self.osc_events and self.fused_tail
                and self.data is not None
                and len(self.cont) * self.n_bins <= K.FINALIZE_METRIC_MAX):
            if self.spin_wait:
                ev = self._evaluator_for() if self.one_call else None
                if ev is not None:
                    return self._eval_one_call(ev, params, kind)
                # The tail kernel's store into pinned host memory is visible a few us before the
                # stream-completion signal has travelled through the runtime: poll it.  NaN is
                # the "not yet" marker (a genuine NaN result falls through to the stream sync).
                h = self._metric_host_np
                if self._split_ok(kind):
                    h[:] = np.nan
                    self._lean_eval(params, kind, split=True)
                    return self._poll_split()
                h[0] = np.nan
                self._lean_eval(params, kind)
                for _ in range(self.spin_wait):
                    v = h[0]
                    if v == v:
                        return float(v)
            else:
                self._lean_eval(params, kind)
        else:
            self.accumulate(params)
            self.allreduce()
            if self.spin_wait and self.data is not None and self.can_fuse_tail():
                # (event-by-event oscillations, unplanned grids: the same tail and the same polled value)
                return self.tail_host(kind)
            self._tail(kind, self.metric_host)
        torch.cuda.current_stream().synchronize()
        return float(self.metric_host[0])

    def _lean_eval(self, params, kind, split=False):
        import ctypes as C

        a = self._lean
        tabs = (self.prob_nu, self.prob_nubar, self.pepmu)
        if a is None or a.get("kind") != "eval" or a["tabs"] is not tabs[2]:
            lib = _lib.lib()
            a = self._lean = dict(
                kind="eval", tabs=self.pepmu, lib=lib, cont=self._cont_arr, n_cont=len(self._cont_arr),
                plan=self.plan.handle, energy=C.c_void_p(self.energy_d.data_ptr()),
                n_e=self.energy_d.numel(), e_major=1 if self.grid.energy_first else 0,
                nu=C.c_void_p(self.prob_nu.data_ptr()), nubar=C.c_void_p(self.prob_nubar.data_ptr()),
                pepmu=C.c_void_p(self.pepmu.data_ptr()), grid=C.byref(self.grid.binning),
                outb=C.byref(self.out_binning), limbs=C.c_void_p(self.ws.limbs.data_ptr()),
                status=C.c_void_p(self.ws.status.data_ptr()), hist=C.c_void_p(self.ws.hist.data_ptr()),
                sumw2=C.c_void_p(self.ws.sumw2.data_ptr()), data=C.c_void_p(self.data.data_ptr()),
                data_t=self.data, out=C.c_void_p(self.metric_host.data_ptr()),
                mstatus=C.c_void_p(self.metric_status.data_ptr()))
        if a["data_t"] is not self.data:  # new pseudo-data
            a["data"], a["data_t"] = C.c_void_p(self.data.data_ptr()), self.data
        self._release_outputs()
        lib, s = a["lib"], K._stream()
        # the fused kernel reads the (P_e, P_mu) gather tables only: the full P[3][3] tables (72 B per
        # node, poorly coalesced stores) are not written on this path (`compute_probs` writes them)
        rc = lib.pisa_hip_prob3_grid_planned(C.byref(params), a["plan"], a["energy"], a["n_e"],
                                             a["e_major"], None, None, a["pepmu"], s)
        if rc == 0 and self.node_flux:
            rc = self._flux_tables(self.pepmu)
        if rc == 0:
            fn = lib.pisa_hip_reweight_hist_acc if self._limbs_zero else lib.pisa_hip_reweight_hist
            rc = fn(a["cont"], a["n_cont"], a["grid"], a["nu"], a["nubar"], a["pepmu"], a["outb"],
                    a["limbs"], a["status"], s)
        self._limbs_zero = self._maps_valid = False
        if rc == 0:
            self.allreduce()
            if split:
                rc = lib.pisa_hip_finalize_metric_parts(a["limbs"], 1, a["n_cont"], self.n_bins, a["hist"],
                                                        a["sumw2"], K.METRIC_KIND[kind], a["data"], None, 0, None,
                                                        a["out"], self.TAIL_PARTS, a["status"], a["mstatus"], 1, s)
            else:
                rc = lib.pisa_hip_finalize_metric(a["limbs"], a["n_cont"], self.n_bins, a["hist"],
                                                  a["sumw2"], K.METRIC_KIND[kind], a["data"], a["out"],
                                                  a["status"], a["mstatus"], 1, s)
            self._limbs_zero = self._maps_valid = rc == 0
        _lib.check(rc)

    def eval_batch(self, params_list, kind="llh"):
        """Several INDEPENDENT parameter points (e.g. the 2n+1 points of a
        finite-difference gradient): the oscillation kernels of point k+1 run on
        a second HIP stream while the fused reweight+histogram kernel of point k
        streams the events, with double-buffered probability tables.  Results
        are identical to calling `eval` point by point; one host sync at the
        end.  Returns a device tensor with one metric value per point."""
        assert not self.osc_events and self.plan is not None and self.data is not None
        self._release_outputs()   # device-backed maps of the previous evaluation: home before the limbs are reused
        n = len(params_list)
        out = torch.empty(n, dtype=torch.float64, device=self.dev)
        if not hasattr(self, "_osc_stream"):
            # high priority: the small oscillation kernels must not queue behind the
            # chip-filling fused kernel of the previous point
            self._osc_stream = torch.cuda.Stream(device=self.dev, priority=-1)
            self._tables = [(self.prob_nu, self.prob_nubar, self.pepmu),
                            tuple(torch.empty_like(t) for t in (self.prob_nu, self.prob_nubar, self.pepmu))]
        main = torch.cuda.current_stream()
        osc_done = [torch.cuda.Event() for _ in range(n)]
        used = [None, None]  # event after which a table set may be overwritten
        self._osc_stream.wait_stream(main)
        for k, p in enumerate(params_list):
            tab = self._tables[k % 2]
            with torch.cuda.stream(self._osc_stream):
                if used[k % 2] is not None:
                    self._osc_stream.wait_event(used[k % 2])
                K.prob3_grid_planned(p, self.plan, self.energy_d, e_major=self.grid.energy_first,
                                     out_nu=tab[0], out_nubar=tab[1], out_pepmu=tab[2])
                osc_done[k].record(self._osc_stream)
            main.wait_event(osc_done[k])
            if self.node_flux:
                _lib.check(self._flux_tables(tab[2]))
            K.reweight_hist(self._cont_arr, self.grid.binning, tab[0], tab[1], tab[2],
                            self.out_binning, self.ws, clear=not self._limbs_zero)
            used[k % 2] = torch.cuda.Event()
            used[k % 2].record(main)
            self._limbs_zero = self._maps_valid = False
            self.allreduce()
            self._tail(kind, out[k:k + 1])
        self.prob_nu, self.prob_nubar, self.pepmu = self._tables[(n - 1) % 2] if n else self._tables[0]
        return out

    # -- several independent parameter points in one sweep of the events ---------------------------
    def multi_capable(self, plan=None):
        """whether `eval_many` can take its one-sweep path: planned grid oscillation (the engine's own
        plan or the caller's, e.g. the osc.prob3 stage's), 16-bit index columns read through the shared
        grid tables, all maps in one tail workgroup, LDS room for at least two points"""
        return ((plan or self.plan) is not None and self.indexed and self.index16 and not self.osc_events
                and not self.node_flux and self.fused_tail and self.data is not None
                and len(self.cont) * self.n_bins <= K.FINALIZE_METRIC_MAX
                and _lib.lib().pisa_hip_multi_points_per_pass(self.n_bins) >= 2)

    def _multi_ws(self, k):
        ws = getattr(self, "_multi", None)
        if ws is None:
            ws = self._multi = {}
        w = ws.get(k)
        if w is None:
            import ctypes as C

            n_c = len(self.cont)
            w = ws[k] = dict(
                tables=torch.empty((2, 3, self.grid.size, k, 2), dtype=torch.float64, device=self.dev),
                limbs=torch.zeros((k, n_c, self.n_bins, 2, _lib.ACC_LIMBS), dtype=torch.int64, device=self.dev),
                hist=torch.empty((k, n_c, self.n_bins), dtype=torch.float64, device=self.dev),
                sumw2=torch.empty((k, n_c, self.n_bins), dtype=torch.float64, device=self.dev),
                host=torch.zeros(4 * k, dtype=torch.float64).pin_memory(),   # k values, or 4 k partial sums (split tail)
                params=(_lib.Prob3Params * k)(), scales=(C.c_double * (k * n_c))(), zero=True)
            w["host_np"] = w["host"].numpy()
        return w

    def eval_many(self, params_list, kind="llh", scales=None, plan=None, energy=None):
        """K INDEPENDENT parameter points (the n + 1 points of a finite-difference gradient, a scan)
        in one sweep of the events: one pair of prob3 launches for all points
        (`pisa_hip_prob3_grid_planned_multi`), one fused launch that reads the event columns once
        and keeps K sets of accumulators (`pisa_hip_reweight_hist_multi`), one all-reduce of the K
        limb sets, one tail launch with a workgroup per point (`pisa_hip_finalize_metric_multi`);
        the K metric values arrive in pinned host memory.  Per point the limbs, maps and metric are
        bit-identical to `eval_host` at that point.  `scales` [K][n_containers] (optional): the
        containers' aeff scales per point.  `plan` / `energy`: grid plan and node energies of the caller when the
        oscillation tables are not the engine's own (a Pipeline: osc.prob3 holds them).  Returns a list of K floats; the maps of the points stay in
        `last_many` (device tensors hist / sumw2 [K, n_cont, n_bins])."""
        import ctypes as C

        n = len(params_list)
        if n == 0:
            return []
        pg = self.points
        if pg is not None and pg.n_groups > 1:
            # the points dealt to the groups (contiguous blocks), every point computed inside one group; one
            # all-gather of the K values.  A single point is evaluated by every group (nothing to deal).
            lo, hi = pg.block(n)
            self.points = None
            try:
                mine = self.eval_many(params_list[lo:hi], kind, None if scales is None else scales[lo:hi], plan, energy) \
                    if hi > lo else []
            finally:
                self.points = pg
            return pg.gather(mine, n, self.dev)
        plan = plan or self.plan
        energy = energy if energy is not None else getattr(self, "energy_d", None)
        # (a single point goes point by point where the engine has oscillation tables of its own; with the caller's plan
        # -- a stage-built engine whose group was dealt one point of a stencil -- it takes the sweep path with K = 1)
        if (n == 1 and self.plan is not None) or not self.multi_capable(plan):
            assert self.plan is not None, "point-by-point evaluation needs the engine's own oscillation tables"
            out = []
            for i, p in enumerate(params_list):
                if scales is not None:
                    for name, sc in zip(self.names, scales[i]):
                        self.set_scale(name, sc)
                out.append(self.eval_host(p, kind))
            return out
        if n > _lib.MAX_POINTS:
            out = []
            for i in range(0, n, _lib.MAX_POINTS):
                out += self.eval_many(params_list[i:i + _lib.MAX_POINTS], kind,
                                      None if scales is None else scales[i:i + _lib.MAX_POINTS], plan, energy)
            return out
        w = self._multi_ws(n)
        self._release_outputs()
        self._many_sweep(w, params_list, scales, plan, energy)
        if self.world_size > 1:
            if self._rccl is None:
                self.allreduce_setup()
            if self._rccl:
                self._rccl.all_reduce_(w["limbs"], K._stream())
            else:
                allreduce_limbs(w["limbs"], self.world_size, self.group)
        out = self._many_tail(w, n, kind)
        self.last_many = w
        return out

    def _many_sweep(self, w, params_list, scales, plan, energy):
        """prob3 of all points + ONE pass over the events into the points' limb sets (asynchronous)"""
        import ctypes as C

        n = len(params_list)
        lib, s = _lib.lib(), K._stream()
        arr = w["params"]
        for i, p in enumerate(params_list):
            arr[i] = p
        sc_ptr = None
        if scales is not None:
            flat = np.ascontiguousarray(scales, dtype=np.float64).reshape(n * len(self.cont))
            C.memmove(w["scales"], flat.ctypes.data, flat.nbytes)
            sc_ptr = C.cast(w["scales"], C.c_void_p)
        g = self.grid
        rc = lib.pisa_hip_prob3_grid_planned_multi(
            C.cast(arr, C.c_void_p), n, plan.handle, C.c_void_p(energy.data_ptr()),
            energy.numel(), 1 if g.energy_first else 0, C.c_void_p(w["tables"].data_ptr()), s)
        if rc == 0:
            rc = lib.pisa_hip_reweight_hist_multi(
                self._cont_arr, len(self._cont_arr), C.byref(g.binning), C.c_void_p(w["tables"].data_ptr()), n,
                sc_ptr, C.byref(self.out_binning), C.c_void_p(w["limbs"].data_ptr()), 0 if w["zero"] else 1,
                C.c_void_p(self.ws.status.data_ptr()), s)
        w["zero"] = False
        _lib.check(rc)

    def _many_tail(self, w, n, kind):
        """maps + metric of the (all-reduced) limb sets, one workgroup per point; the values arrive in
        pinned host memory"""
        import ctypes as C

        lib, s = _lib.lib(), K._stream()
        h = w["host_np"]
        h[:] = np.nan
        if self._split_ok(kind):
            _lib.check(lib.pisa_hip_finalize_metric_split(
                C.c_void_p(w["limbs"].data_ptr()), n, len(self.cont), self.n_bins, C.c_void_p(w["hist"].data_ptr()),
                C.c_void_p(w["sumw2"].data_ptr()), K.METRIC_KIND[kind], C.c_void_p(self.data.data_ptr()), None, 0, None,
                C.c_void_p(w["host"].data_ptr()), C.c_void_p(self.ws.status.data_ptr()),
                C.c_void_p(self.metric_status.data_ptr()), 1, s))
            w["zero"] = True
            h4 = h[:4 * n]
            for _ in range(self.spin_wait):
                if not np.isnan(h4).any():
                    break
            else:
                torch.cuda.current_stream().synchronize()
            p = h4.reshape(n, 4)
            return [(float(q[0]) + float(q[2])) + (float(q[1]) + float(q[3])) for q in p]
        h = h[:n]
        _lib.check(lib.pisa_hip_finalize_metric_multi(
            C.c_void_p(w["limbs"].data_ptr()), n, len(self.cont), self.n_bins, C.c_void_p(w["hist"].data_ptr()),
            C.c_void_p(w["sumw2"].data_ptr()), K.METRIC_KIND[kind], C.c_void_p(self.data.data_ptr()), None, 0, None,
            C.c_void_p(w["host"].data_ptr()), C.c_void_p(self.ws.status.data_ptr()),
            C.c_void_p(self.metric_status.data_ptr()), 1, s))
        w["zero"] = True      # the tail leaves the limbs zeroed for the next sweep
        for _ in range(self.spin_wait):
            if not np.isnan(h).any():
                return [float(v) for v in h]
        torch.cuda.current_stream().synchronize()
        return [float(v) for v in h]

    def metric_status_host(self):
        """status word of the metric kernels (negative inputs) -- one 4-byte read"""
        st = int(self.metric_status.item())
        if st != 0:
            self.metric_status.zero_()
        return st

    def check_status(self):
        st = int(self.ws.status.item())
        if st != 0:
            self.ws.status.zero_()
            if st & 2:
                raise ValueError("partition table of the resident order (pisa_hip_container::d_part_start) does not start at 0, "
                                 "decreases or stops short of the container: events would have been dropped")
            raise OverflowError("event weight not finite or outside the accumulator range")
        st = int(self.metric_status.item())
        if st != 0:
            self.metric_status.zero_()
            _lib.check(st)

    def maps(self):
        """host copies: (hist[n_cont, n_bins], sumw2[...])"""
        return self.ws.hist.cpu().numpy(), self.ws.sumw2.cpu().numpy()
