// order.hip -- the resident ORDER of a container's events for the 16-bit index form of the fused kernel
// (engine.deposit_block_order; the accumulation is exact, so the order is free to choose -- DESIGN section 3).
//
// What the order is (the torch formulation in pisa_amd/engine.py is the specification and stays the test's reference):
//   1. events that can deposit (inside the output binning AND the calc grid) first, sorted by calc-grid node (gather
//      locality), ties in input order; the others ("idle") behind them, sorted by node as well;
//   2. inside whole windows of 4 096 depositing events: dealt round-robin over the 32 LDS bank pairs (bin mod 32) --
//      sorted by (rank inside the (window, residue) queue, residue) -- and laid out so that 32 consecutive emissions land
//      in the same slot of 32 consecutive quads (lds_bank_order with per = 4);
//   3. the depositing blocks of 256 events (what one wavefront takes per sweep) spread evenly among the idle blocks.
//
// Round 5: the torch formulation costs ~40 launches, four sorts and two host synchronisations per container (2 ms per
// 8.3e5 events: 24 ms of a 44 ms set-up at 1e7 events).  Here: one key per event, ONE stable radix sort on 17-18 bits,
// one workgroup per window for step 2 (ballot ranks + a 32-entry table: no second and third sort), step 3 in closed form
// inside the final write.  No host synchronisation: the number of depositing events stays on the device.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "common.hpp"

namespace pisa {

constexpr int ORD_WINDOW = 4096, ORD_BANKS = 32, ORD_PER = 4, ORD_BLOCK = 256;

// key: depositing events by node, idle events behind them by node (node = -1 first).  A fixed number of workgroups walks
// the events (grid stride): the count of depositing events is ONE atomic per workgroup on one address, and such atomics go
// one after the other at ~17 ns each across the eight dies (3 255 of them per 8.3e5-event container cost more than the keys).
constexpr int ORD_KEY_BLOCKS = 512;
__global__ void __launch_bounds__(256)
order_key_kernel(const int32_t *__restrict__ node, const int32_t *__restrict__ bin, int64_t n, uint32_t n_nodes,
                 uint32_t *__restrict__ key, uint32_t *__restrict__ val, unsigned long long *__restrict__ n_dep) {
    unsigned int mine = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int32_t nd = node[i], b = bin[i];
        const bool dep = nd >= 0 && b >= 0;
        key[i] = dep ? (uint32_t)nd : n_nodes + 1u + (uint32_t)(nd + 1);
        val[i] = (uint32_t)i;
        mine += dep ? 1u : 0u;
    }
    __shared__ unsigned int cnt;
    if (threadIdx.x == 0) cnt = 0u;
    __syncthreads();
    if (mine) atomicAdd(&cnt, mine);                           // (integer counts: order-independent)
    __syncthreads();
    if (threadIdx.x == 0 && cnt) atomicAdd(n_dep, (unsigned long long)cnt);
}

// step 2: one workgroup per window of 4 096 sorted events
__global__ void __launch_bounds__(1024)
order_bank_kernel(const uint32_t *__restrict__ sorted, const int32_t *__restrict__ bin, int64_t n,
                  const unsigned long long *__restrict__ n_dep_p, uint32_t *__restrict__ out) {
    __shared__ unsigned short s_cnt[64 * ORD_BANKS];   // [chunk of 64 positions][residue]: count, then exclusive offset
    __shared__ unsigned short s_tot[ORD_BANKS];
    const int64_t base = (int64_t)blockIdx.x * ORD_WINDOW;
    const int64_t n_dep = (int64_t)*n_dep_p;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (base + ORD_WINDOW > n_dep) {   // not a whole window of depositing events: kept as sorted
        for (int p = threadIdx.x; p < ORD_WINDOW && base + p < n; p += 1024) out[base + p] = sorted[base + p];
        return;
    }
    uint32_t v[4];
    int res[4], rk[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int chunk = wave * 4 + c;
        v[c] = sorted[base + chunk * 64 + lane];
        res[c] = (int)((uint32_t)bin[v[c]] % ORD_BANKS);
        rk[c] = 0;
        for (int r = 0; r < ORD_BANKS; r++) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(res[c] == r);
            if (res[c] == r) rk[c] = __builtin_popcountll(m & ((1ull << lane) - 1ull));
            if (lane == r) s_cnt[chunk * ORD_BANKS + r] = (unsigned short)__builtin_popcountll(m);
        }
    }
    __syncthreads();
    if (threadIdx.x < ORD_BANKS) {     // exclusive prefix over the chunks, per residue
        unsigned int run = 0;
        for (int chunk = 0; chunk < 64; chunk++) {
            const unsigned int c = s_cnt[chunk * ORD_BANKS + threadIdx.x];
            s_cnt[chunk * ORD_BANKS + threadIdx.x] = (unsigned short)run;
            run += c;
        }
        s_tot[threadIdx.x] = (unsigned short)run;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int chunk = wave * 4 + c;
        const int k = (int)s_cnt[chunk * ORD_BANKS + res[c]] + rk[c];     // rank inside the (window, residue) queue
        // emission position: every residue's queue contributes its first min(len, k) events, the shorter residues of
        // rank k come before this one
        int s = 0;
        for (int r = 0; r < ORD_BANKS; r++) {
            const int len = (int)s_tot[r];
            s += len < k ? len : k;
            s += (r < res[c] && len > k) ? 1 : 0;
        }
        const int blk = s / (32 * ORD_PER), t = s % (32 * ORD_PER);
        const int slot = ORD_PER * (blk * 32 + (t % 32)) + t / 32;
        out[base + slot] = v[c];
    }
}

// step 3 + the int64 permutation torch gathers with
__global__ void __launch_bounds__(256)
order_interleave_kernel(const uint32_t *__restrict__ seq, int64_t n, const unsigned long long *__restrict__ n_dep_p,
                        int64_t *__restrict__ perm) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t t_full = n / ORD_BLOCK;
    const int64_t n_dep = (int64_t)*n_dep_p;
    int64_t nbb = (n_dep + ORD_BLOCK - 1) / ORD_BLOCK;
    if (nbb > t_full) nbb = t_full;
    const int64_t j = i / ORD_BLOCK, l = i % ORD_BLOCK;
    int64_t src = j;
    if (j < t_full && nbb > 0 && nbb < t_full) {
        // depositing block q sits at output block floor(q t_full / nbb); the idle blocks fill the rest in order
        const int64_t q = (j * nbb + t_full - 1) / t_full;          // depositing positions in front of block j
        const bool is_b = q < nbb && (q * t_full) / nbb == j;
        src = is_b ? q : nbb + (j - q);
    }
    perm[i] = (int64_t)seq[src * ORD_BLOCK + l];
}


// ------------------------------------------------------------------ partitioned order (binnings beyond the LDS accumulators)
// engine.window_partition_order in two native calls (round 6).  The torch formulation -- per partition a nonzero, a stable
// argsort by node, the bank order, concatenations and an index shuffle: ~25 launches per container -- stays the
// specification and the test's reference; the block accounting between the two calls (how many idle blocks each
// partition is topped up and interleaved with) stays on the host, in Python, as it was.
//   call 1 (`_sort`): key = (partition of the event's bin, node) for depositing events, (n_part, node + 1) for the idle
//                     ones; ONE stable radix sort; the number of events per partition (+ idle) in n_part + 1 counters;
//   call 2 (`_assemble`): the bank order inside whole 4 096-event windows of every partition with >= 8 192 events (the
//                     window kernel above, per segment), then every output position's source in closed form from the
//                     host's table (depositing blocks spread evenly among a partition's idle blocks).
constexpr int PORD_MAX_PART = 255;      // (PART_MAX of hist.hip)

__global__ void __launch_bounds__(256)
part_key_kernel(const int32_t *__restrict__ node, const int32_t *__restrict__ bin, int64_t n, uint32_t n_nodes, uint32_t width,
                uint32_t n_part, uint32_t *__restrict__ key, uint32_t *__restrict__ val, unsigned long long *__restrict__ counts) {
    __shared__ unsigned int cnt[PORD_MAX_PART + 1];
    for (int k = threadIdx.x; k <= (int)n_part; k += 256) cnt[k] = 0u;
    __syncthreads();
    const uint32_t K = n_nodes + 1u;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int32_t nd = node[i], b = bin[i];
        const bool dep = nd >= 0 && b >= 0;
        const uint32_t part = dep ? (uint32_t)b / width : n_part;
        key[i] = part * K + (dep ? (uint32_t)nd : (uint32_t)(nd + 1));      // (idle: node -1 first, as argsort(node) has it)
        val[i] = (uint32_t)i;
        atomicAdd(&cnt[part < n_part ? part : n_part], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k <= (int)n_part; k += 256)
        if (cnt[k]) atomicAdd(&counts[k], (unsigned long long)cnt[k]);        // (integer counts: order-independent)
}

struct PartOrderTable {
    int32_t n_part;
    int32_t pad;
    int64_t n, idle_off;                        // events; first idle event in the sorted sequence
    int64_t tail_out, tail_idle;                // output position / idle index of what is left behind the last partition
    int64_t out_block[PORD_MAX_PART + 1];       // first output block of partition p
    int64_t dep_off[PORD_MAX_PART + 1];         // first sorted event of partition p
    int32_t n_dep[PORD_MAX_PART + 1];           // its depositing events
    int32_t nb[PORD_MAX_PART + 1], nf[PORD_MAX_PART + 1];   // its depositing blocks (topped up) and idle filler blocks
    int64_t idle_top[PORD_MAX_PART + 1];        // idle index of its top-up events; the filler follows them
    int64_t win_base[PORD_MAX_PART + 2];        // bank-order windows of the partitions in front of p (whole windows of the >= 8 192-event ones)
};

// step 2 per segment: window w of the launch belongs to the partition whose [win_base[p], win_base[p + 1]) holds it
__global__ void __launch_bounds__(1024)
part_bank_kernel(const uint32_t *__restrict__ sorted, const int32_t *__restrict__ bin, const PartOrderTable *__restrict__ tp,
                 uint32_t *__restrict__ out) {
    __shared__ unsigned short s_cnt[64 * ORD_BANKS];
    __shared__ unsigned short s_tot[ORD_BANKS];
    const PartOrderTable &T = *tp;
    int p = 0;
    while (p + 1 < T.n_part && (int64_t)blockIdx.x >= T.win_base[p + 1]) p++;
    const int64_t base = T.dep_off[p] + ((int64_t)blockIdx.x - T.win_base[p]) * ORD_WINDOW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t v[4];
    int res[4], rk[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int chunk = wave * 4 + c;
        v[c] = sorted[base + chunk * 64 + lane];
        res[c] = (int)((uint32_t)bin[v[c]] % ORD_BANKS);
        rk[c] = 0;
        for (int r = 0; r < ORD_BANKS; r++) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(res[c] == r);
            if (res[c] == r) rk[c] = __builtin_popcountll(m & ((1ull << lane) - 1ull));
            if (lane == r) s_cnt[chunk * ORD_BANKS + r] = (unsigned short)__builtin_popcountll(m);
        }
    }
    __syncthreads();
    if (threadIdx.x < ORD_BANKS) {
        unsigned int run = 0;
        for (int chunk = 0; chunk < 64; chunk++) {
            const unsigned int c = s_cnt[chunk * ORD_BANKS + threadIdx.x];
            s_cnt[chunk * ORD_BANKS + threadIdx.x] = (unsigned short)run;
            run += c;
        }
        s_tot[threadIdx.x] = (unsigned short)run;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int chunk = wave * 4 + c;
        const int k = (int)s_cnt[chunk * ORD_BANKS + res[c]] + rk[c];
        int s = 0;
        for (int r = 0; r < ORD_BANKS; r++) {
            const int len = (int)s_tot[r];
            s += len < k ? len : k;
            s += (r < res[c] && len > k) ? 1 : 0;
        }
        const int blk = s / (32 * ORD_PER), t = s % (32 * ORD_PER);
        const int slot = ORD_PER * (blk * 32 + (t % 32)) + t / 32;
        out[base + slot] = v[c];
    }
}

// every output position's source: `banked` = the sorted sequence with the bank order applied where it applies
__global__ void __launch_bounds__(256)
part_assemble_kernel(const uint32_t *__restrict__ banked, const PartOrderTable *__restrict__ tp, int64_t *__restrict__ perm) {
    const PartOrderTable &T = *tp;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= T.n) return;
    if (i >= T.tail_out) {          // what is left of the idle events behind the last partition
        perm[i] = (int64_t)banked[T.idle_off + T.tail_idle + (i - T.tail_out)];
        return;
    }
    const int64_t jg = i / ORD_BLOCK, l = i % ORD_BLOCK;
    int p = 0;
    while (p + 1 < T.n_part && jg >= T.out_block[p + 1]) p++;     // (workgroup-uniform: a block lies in one partition)
    const int64_t j = jg - T.out_block[p], nb = T.nb[p], nf = T.nf[p], t = nb + nf;
    int64_t sb = j;
    if (nb > 0 && nf > 0) {
        // depositing block q sits at block floor(q t / nb) of the partition; the filler blocks take the rest in order
        const int64_t q = (j * nb + t - 1) / t;
        const bool is_b = q < nb && (q * t) / nb == j;
        sb = is_b ? q : nb + (j - q);
    }
    int64_t src;
    if (sb < nb) {
        const int64_t e = sb * ORD_BLOCK + l;
        src = e < T.n_dep[p] ? T.dep_off[p] + e : T.idle_off + T.idle_top[p] + (e - T.n_dep[p]);
    } else {
        const int64_t top = nb * ORD_BLOCK - T.n_dep[p];
        src = T.idle_off + T.idle_top[p] + top + (sb - nb) * ORD_BLOCK + l;
    }
    perm[i] = (int64_t)banked[src];
}

// the resident copies of a container's columns in the order `perm` and their interleaved / folded forms, one thread per event
__global__ void __launch_bounds__(256)
pack_columns_kernel(const pisa_hip_pack_set a) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= a.n_pad) return;
    if (k >= a.n) {
        a.o_node_bin16[k] = (int32_t)0xFFFFFFFFu;
        return;
    }
    const int64_t i = a.d_perm[k];
    const double aeff = a.d_weighted_aeff[i], w0 = a.d_initial_weights[i];
    const int32_t nd = a.d_node[i], b = a.d_bin[i];
    a.o_grid_x[k] = a.d_grid_x[i];
    a.o_grid_y[k] = a.d_grid_y[i];
    a.o_nu_flux[2 * k] = a.d_nu_flux[2 * i];
    a.o_nu_flux[2 * k + 1] = a.d_nu_flux[2 * i + 1];
    a.o_weighted_aeff[k] = aeff;
    a.o_initial_weights[k] = w0;
    for (int c = 0; c < a.n_sample; c++) a.o_sample[c][k] = a.d_sample[c][i];
    a.o_node[k] = nd;
    a.o_bin[k] = b;
    a.o_node_bin[2 * k] = nd;
    a.o_node_bin[2 * k + 1] = b;
    a.o_aeff_w0[2 * k] = aeff;
    a.o_aeff_w0[2 * k + 1] = w0;
    a.o_static_w[k] = w0 * aeff;
    const uint32_t lo = nd < 0 ? 0xFFFFu : (uint32_t)nd, hi = b < 0 ? 0xFFFFu : (uint32_t)b;
    a.o_node_bin16[k] = (int32_t)(lo | (hi << 16));
}

// bytes of one of the five index arrays, a multiple of 256: the sort's temporary storage behind them holds 64-bit look-back
// states and must not start at an odd multiple of four bytes (an odd event count did that: the sort of 3.3e6 events then hung
// under a counter-collecting profiler)
static size_t order_stride(int64_t n) { return (((size_t)(n + ORD_WINDOW) * 4 + 255) / 256) * 256; }

static int order_key_bits(uint32_t n_nodes) {
    int bits = 1;
    while (bits < 32 && (1ull << bits) <= 2ull * n_nodes + 2ull) bits++;
    return bits;
}

}  // namespace pisa

using namespace pisa;

PISA_API int64_t pisa_hip_deposit_block_order_workspace(int64_t n) {
    if (n < 0 || n > 0x7FFFFFF0LL) return -1;
    size_t temp = 0;
    (void)rocprim::radix_sort_pairs(nullptr, temp, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                    (uint32_t *)nullptr, (size_t)n, 0u, 32u);
    return (int64_t)(5 * order_stride(n) + temp + 1024);
}

PISA_API int pisa_hip_deposit_block_order(const int32_t *d_node, const int32_t *d_bin, int64_t n, int64_t n_nodes,
                                          int64_t *d_perm, void *d_work, int64_t work_bytes, void *stream) {
    if (n < 0 || n > 0x7FFFFFF0LL || n_nodes < 1 || n_nodes > 0x3FFFFFFFLL) return PISA_HIP_ERR_INVALID;
    if (n == 0) return PISA_HIP_OK;
    if (!d_node || !d_bin || !d_perm || !d_work) return PISA_HIP_ERR_INVALID;
    const int64_t need = pisa_hip_deposit_block_order_workspace(n);
    if (need < 0 || work_bytes < need) return PISA_HIP_ERR_NOMEM;
    hipStream_t s = as_stream(stream);
    char *w = (char *)d_work;
    unsigned long long *n_dep = (unsigned long long *)w;
    const size_t stride = order_stride(n);
    uint32_t *key_a = (uint32_t *)(w + 256), *key_b = (uint32_t *)(w + 256 + stride);
    uint32_t *val_a = (uint32_t *)(w + 256 + 2 * stride), *val_b = (uint32_t *)(w + 256 + 3 * stride);
    uint32_t *seq = (uint32_t *)(w + 256 + 4 * stride);
    char *temp = w + 256 + 5 * stride;
    size_t temp_bytes = (size_t)work_bytes - (256 + 5 * stride);
    PISA_TRY_HIP(hipMemsetAsync(n_dep, 0, 8, s));
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(order_key_kernel, dim3(nb < (unsigned)ORD_KEY_BLOCKS ? nb : (unsigned)ORD_KEY_BLOCKS), dim3(256), 0, s, d_node, d_bin, n, (uint32_t)n_nodes, key_a, val_a, n_dep);
    PISA_TRY_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, key_a, key_b, val_a, val_b, (size_t)n, 0u,
                                           (unsigned)order_key_bits((uint32_t)n_nodes), s));
    hipLaunchKernelGGL(order_bank_kernel, dim3((unsigned)((n + ORD_WINDOW - 1) / ORD_WINDOW)), dim3(1024), 0, s, val_b,
                       d_bin, n, n_dep, seq);
    hipLaunchKernelGGL(order_interleave_kernel, dim3(nb), dim3(256), 0, s, seq, n, n_dep, d_perm);
    PISA_CHECK_LAUNCH("deposit block order kernels");
    return PISA_HIP_OK;
}


PISA_API int64_t pisa_hip_partition_order_workspace(int64_t n) {
    if (n < 0 || n > 0x7FFFFFF0LL) return -1;
    size_t temp = 0;
    (void)rocprim::radix_sort_pairs(nullptr, temp, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                    (uint32_t *)nullptr, (size_t)n, 0u, 32u);
    return (int64_t)(4096 + sizeof(PartOrderTable) + 256 + 5 * order_stride(n) + temp + 1024);
}

namespace {
struct PartWork {
    unsigned long long *counts;
    PartOrderTable *table;
    uint32_t *key_a, *key_b, *val_a, *val_b, *banked;
    char *temp;
    size_t temp_bytes;
};
PartWork part_work(void *d_work, int64_t work_bytes, int64_t n) {
    char *w = (char *)d_work;
    PartWork W;
    W.counts = (unsigned long long *)w;                                   // PORD_MAX_PART + 1 counters (2 KB)
    const size_t tab = 4096, tab_bytes = (sizeof(PartOrderTable) + 255) / 256 * 256;
    W.table = (PartOrderTable *)(w + tab);
    const size_t stride = order_stride(n), base = tab + tab_bytes;
    W.key_a = (uint32_t *)(w + base);
    W.key_b = (uint32_t *)(w + base + stride);
    W.val_a = (uint32_t *)(w + base + 2 * stride);
    W.val_b = (uint32_t *)(w + base + 3 * stride);
    W.banked = (uint32_t *)(w + base + 4 * stride);
    W.temp = w + base + 5 * stride;
    W.temp_bytes = (size_t)work_bytes - (base + 5 * stride);
    return W;
}
}  // namespace

PISA_API int pisa_hip_partition_order_sort(const int32_t *d_node, const int32_t *d_bin, int64_t n, int64_t n_nodes,
                                           int32_t width, int32_t n_part, int64_t *h_counts, void *d_work,
                                           int64_t work_bytes, void *stream) {
    if (n < 1 || n > 0x7FFFFFF0LL || n_nodes < 1 || width < 1 || n_part < 1 || n_part > PORD_MAX_PART ||
        (uint64_t)(n_part + 1) * (uint64_t)(n_nodes + 1) > 0xFFFFFFFFull)
        return PISA_HIP_ERR_INVALID;
    if (!d_node || !d_bin || !h_counts || !d_work) return PISA_HIP_ERR_INVALID;
    const int64_t need = pisa_hip_partition_order_workspace(n);
    if (need < 0 || work_bytes < need) return PISA_HIP_ERR_NOMEM;
    hipStream_t s = as_stream(stream);
    PartWork W = part_work(d_work, work_bytes, n);
    PISA_TRY_HIP(hipMemsetAsync(W.counts, 0, (PORD_MAX_PART + 1) * sizeof(unsigned long long), s));
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(part_key_kernel, dim3(nb < (unsigned)ORD_KEY_BLOCKS ? nb : (unsigned)ORD_KEY_BLOCKS), dim3(256), 0, s, d_node,
                       d_bin, n, (uint32_t)n_nodes, (uint32_t)width, (uint32_t)n_part, W.key_a, W.val_a, W.counts);
    PISA_CHECK_LAUNCH("part_key_kernel");
    int bits = 1;
    while (bits < 32 && (1ull << bits) < (uint64_t)(n_part + 1) * (uint64_t)(n_nodes + 1)) bits++;
    PISA_TRY_HIP(rocprim::radix_sort_pairs(W.temp, W.temp_bytes, W.key_a, W.key_b, W.val_a, W.val_b, (size_t)n, 0u, (unsigned)bits, s));
    unsigned long long h[PORD_MAX_PART + 1];
    PISA_TRY_HIP(hipMemcpyAsync(h, W.counts, (size_t)(n_part + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    PISA_TRY_HIP(hipStreamSynchronize(s));
    for (int p = 0; p <= n_part; p++) h_counts[p] = (int64_t)h[p];
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_partition_order_assemble(const int32_t *d_bin, int64_t n, int32_t n_part, const int64_t *h_n_dep,
                                               const int64_t *h_dep_blocks, const int64_t *h_idle_blocks, int64_t *d_perm,
                                               void *d_work, int64_t work_bytes, void *stream) {
    if (n < 1 || n > 0x7FFFFFF0LL || n_part < 1 || n_part > PORD_MAX_PART || !d_bin || !h_n_dep || !h_dep_blocks ||
        !h_idle_blocks || !d_perm || !d_work)
        return PISA_HIP_ERR_INVALID;
    const int64_t need = pisa_hip_partition_order_workspace(n);
    if (need < 0 || work_bytes < need) return PISA_HIP_ERR_NOMEM;
    hipStream_t s = as_stream(stream);
    PartWork W = part_work(d_work, work_bytes, n);
    PartOrderTable T;
    memset(&T, 0, sizeof(T));
    T.n_part = n_part;
    T.n = n;
    int64_t dep_off = 0, out_block = 0, idle_at = 0, wins = 0;
    for (int p = 0; p < n_part; p++) {
        const int64_t nd = h_n_dep[p], nb = h_dep_blocks[p], nf = h_idle_blocks[p];
        if (nd < 0 || nd > 0x7FFFFFF0LL || nb < 0 || nf < 0 || nb * ORD_BLOCK < nd || nb * ORD_BLOCK - nd >= ORD_BLOCK)
            return PISA_HIP_ERR_INVALID;      // (a partition's depositing blocks are its events topped up to whole blocks)
        T.out_block[p] = out_block;
        T.dep_off[p] = dep_off;
        T.n_dep[p] = (int32_t)nd;
        T.nb[p] = (int32_t)nb;
        T.nf[p] = (int32_t)nf;
        T.idle_top[p] = idle_at;
        T.win_base[p] = wins;
        if (nd >= 2 * ORD_WINDOW) wins += nd / ORD_WINDOW;       // (engine.window_partition_order: bank order from 8 192 events on)
        idle_at += (nb * ORD_BLOCK - nd) + nf * ORD_BLOCK;
        dep_off += nd;
        out_block += nb + nf;
    }
    T.out_block[n_part] = out_block;
    T.win_base[n_part] = wins;
    T.idle_off = dep_off;
    T.tail_out = out_block * ORD_BLOCK;
    T.tail_idle = idle_at;
    if (T.tail_out > n || dep_off + idle_at + (n - T.tail_out) != n) return PISA_HIP_ERR_INVALID;   // the host's accounting does not add up
    PISA_TRY_HIP(hipMemcpyAsync(W.table, &T, sizeof(T), hipMemcpyHostToDevice, s));
    // the sorted sequence as it is, then the whole windows of the large partitions in the bank order
    PISA_TRY_HIP(hipMemcpyAsync(W.banked, W.val_b, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    if (wins > 0)
        hipLaunchKernelGGL(part_bank_kernel, dim3((unsigned)wins), dim3(1024), 0, s, W.val_b, d_bin, W.table, W.banked);
    hipLaunchKernelGGL(part_assemble_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W.banked, W.table, d_perm);
    PISA_CHECK_LAUNCH("partition order kernels");
    PISA_TRY_HIP(hipStreamSynchronize(s));      // (the table was uploaded from this frame)
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_pack_resident_columns(const pisa_hip_pack_set *set, void *stream) {
    if (!set || set->n < 0 || set->n_pad < set->n || set->n_pad % 256 != 0 || set->n_pad > 0x7FFFFFF0LL || set->n_sample < 0 || set->n_sample > 3)
        return PISA_HIP_ERR_INVALID;
    if (set->n_pad == 0) return PISA_HIP_OK;
    const pisa_hip_pack_set &a = *set;
    if (!a.o_node_bin16) return PISA_HIP_ERR_INVALID;
    if (a.n > 0) {
        if (!a.d_perm || !a.d_grid_x || !a.d_grid_y || !a.d_nu_flux || !a.d_weighted_aeff || !a.d_initial_weights || !a.d_node || !a.d_bin ||
            !a.o_grid_x || !a.o_grid_y || !a.o_nu_flux || !a.o_weighted_aeff || !a.o_initial_weights || !a.o_node || !a.o_bin ||
            !a.o_node_bin || !a.o_aeff_w0 || !a.o_static_w)
            return PISA_HIP_ERR_INVALID;
        for (int c = 0; c < a.n_sample; c++)
            if (!a.d_sample[c] || !a.o_sample[c]) return PISA_HIP_ERR_INVALID;
    }
    hipLaunchKernelGGL(pack_columns_kernel, dim3((unsigned)(a.n_pad / 256)), dim3(256), 0, as_stream(stream), a);
    PISA_CHECK_LAUNCH("pack_columns_kernel");
    return PISA_HIP_OK;
}
