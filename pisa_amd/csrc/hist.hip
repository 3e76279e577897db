// hist.hip -- HBM-bound half of the hot path for gfx950:
//   event_indices           digitise coordinates once (they never change
//                           between evaluations): flat bin / grid-node index
//   lookup_regular          grid -> event gather          (translation.py:417-501)
//   histogram_regular       weighted N-D histogram        (translation.py:90-205)
//   reweight_hist           fused prob3.apply + aeff.apply + hist.apply(sumw2)
//   hist_finalize           fixed point -> fp64 maps
//
// ORDER-INDEPENDENT ACCUMULATION.  Every summand x = +-m 2^e is cut EXACTLY into the (at most
// three) 32-bit digits it occupies in a fixed-point number of NL 32-bit slabs with LSB 2^-116
// (`deposit_units`); the digits are added with integer LDS atomics (ds_add_u64) into slab
// accumulators [slab][quantity][bin], which give the same sum whatever the event order.  At
// the end of a workgroup the slab accumulators are added to the global limb array with integer
// atomics (associative => the result is independent of workgroup scheduling, workgroup count
// and GPU count).  The limbs [container][bin][quantity][6] can be SUM-all-reduced across ranks
// as int64 and are rounded ONCE (round-to-nearest-even) to fp64 by hist_finalize.
// Range: |x| < 2^76, resolution 2^-116; outside -> status flag.
//
// Roofline: HBM.  Per event the indexed fused kernel reads
//   node i32 + bin i32 + nu_flux 2xf64 + weighted_aeff f64 + initial_weights f64
//   = 40 B      (coordinates digitised once at setup, like the reference's own
//                pre-digitised irregular dimensions, utils/hist.py:100-113),
//   24 B in the compact form (static factors folded into the flux pair once),
//   20 B with the two indices in 16 bits each (MODE 7, the default where it applies),
// and the coordinate form reads 8 B x (2 lookup coords + D sample coords) more
// = 72 B (D=3).  The (P_e, P_mu) gather tables (<= 3.8 MB) stay in L2.
#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include <string.h>

#include "common.hpp"
#include "metric_device.hpp"

namespace pisa {

constexpr int NL = PISA_HIP_ACC_LIMBS;  // slabs ("limbs") per accumulator
constexpr int FX_LSB = 116;             // value = sum limb_j * 2^(32j - 116)
constexpr int MAX_CONT = 16;            // containers per launch (kernarg budget)
constexpr int PART_MAX = 255;           // partitions of a container's resident order (pisa_hip_container::d_part_start)
constexpr int HIST_THREADS = 1024;
constexpr int64_t LDS_ACC_BYTES_MAX = 64 * 1024;

// ---------------------------------------------------------------------------
// (Non-temporal streaming loads were measured A/B in one session: this kernel got 5 us SLOWER,
// 92.3 vs 87.4 us, and the whole evaluation did not change; plain loads are used.)

// The exact decomposition in integer arithmetic: x = +-m * 2^(ex-1075), m the 53-bit significand,
// is truncated to a multiple of 2^-116 and cut into the (at most three) 32-bit digits it occupies,
//   digit[jj] = (|x| / 2^-116 >> 32 jj) & 0xffffffff,  jj = j, j-1, j-2,  j = slab of the leading bit,
// each handed to add(jj, +-digit) as a signed 64-bit count of units 2^(32jj-116).  Integer additions are
// associative, so sums of these counts are exact in any order as long as they fit 64 bits (2^31 events
// per accumulator).  Two 64-bit shifts instead of three rounded add/subtract pairs: a third of the
// instructions of `deposit`, which is what bounds the multi-point kernel.  |x| < 2^-116 deposits nothing;
// digits below slab 0 are dropped (truncation towards zero; `deposit` rounds to nearest there -- the two
// differ by less than 2^-116 per event).
// LDS and global integer atomics with their own memory scopes: apart from being what is meant, the
// different scopes keep the optimiser from merging an `in LDS ? ... : ...` pair of atomics into one
// atomic on a generic pointer (which this compiler then fails to select).
__device__ __forceinline__ void lds_add(unsigned long long *p, unsigned long long v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void glb_add(unsigned long long *p, unsigned long long v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <class F>
__device__ __forceinline__ bool deposit_units_general(double x, F &&add) {
    const unsigned hi = (unsigned)__double2hiint(x);
    const int ex = (hi >> 20) & 0x7ff;
    if (ex == 0x7ff) return false;        // Inf / NaN
    const int t = ex - 1023 + FX_LSB;     // position of the leading bit above the LSB of the format
    if (t < 0) return true;               // below 2^-116 (incl. zero / subnormals)
    const int j = t >> 5;
    if (j >= NL) return false;            // |x| >= 2^76
    const int sh = t & 31;                // position of the leading bit inside digit j
    const unsigned long long m =
        ((unsigned long long)((hi & 0xfffffu) | 0x100000u) << 32) | (unsigned)__double2loint(x);
    const unsigned long long low = m << (sh + 12);     // digits j-1 (high word) and j-2 (low word)
    long long d0 = (long long)(m >> (52 - sh));
    long long d1 = (long long)(low >> 32);
    long long d2 = (long long)(low & 0xffffffffull);
    if ((int)hi < 0) { d0 = -d0; d1 = -d1; d2 = -d2; }
    add(j, d0);
    if (j >= 1 && d1 != 0) add(j - 1, d1);
    if (j >= 2 && d2 != 0) add(j - 2, d2);
    return true;
}

// The case that occurs: x positive, 2^-52 <= x < 2^76 (leading bit in digit 2 or above), where the three
// digits exist and none needs a sign -- no branch inside, `add3(j, d0, d1, d2)` receives the digits of
// slabs j, j-1, j-2 at once.  Anything else (negative, tiny, non-finite, too large) takes the general
// form above; the wavefront decides once (`__any`), so the rare path costs nothing when nobody needs it.
template <class F3, class F>
__device__ __forceinline__ bool deposit_units(double x, F3 &&add3, F &&add) {
    const unsigned hi = (unsigned)__double2hiint(x);
    const unsigned t = (hi >> 20) - (1023u - FX_LSB);          // sign bit set: huge -> not `fast`
    const bool fast = (t - 64u) < (unsigned)(NL * 32 - 64);
    bool ok = true;
    if (__any(!fast)) {
        if (!fast) ok = deposit_units_general(x, add);
    }
    if (fast) {
        const unsigned sh = t & 31u;
        const unsigned long long m =
            ((unsigned long long)((hi & 0xfffffu) | 0x100000u) << 32) | (unsigned)__double2loint(x);
        const unsigned long long low = m << (sh + 12u);
        add3((int)(t >> 5), m >> (52u - sh), low >> 32, low & 0xffffffffull);
    }
    return ok;
}

__device__ __forceinline__ bool bin_index(const DevBinning &b, double x, double y, double z,
                                          int64_t &flat) {
    // half-open [min, max) per dimension, bin = (int)((x - min) * norm)
    // (fast_histogram rule; translation.py:417-456 for lookups)
    if (!(x >= b.mins[0] && x < b.maxs[0])) return false;
    int ix = (int)((x - b.mins[0]) * b.norm[0]);
    ix = ix < b.nb[0] ? ix : b.nb[0] - 1;
    flat = ix;
    if (b.ndim > 1) {
        if (!(y >= b.mins[1] && y < b.maxs[1])) return false;
        int iy = (int)((y - b.mins[1]) * b.norm[1]);
        iy = iy < b.nb[1] ? iy : b.nb[1] - 1;
        flat = flat * b.nb[1] + iy;
    }
    if (b.ndim > 2) {
        if (!(z >= b.mins[2] && z < b.maxs[2])) return false;
        int iz = (int)((z - b.mins[2]) * b.norm[2]);
        iz = iz < b.nb[2] ? iz : b.nb[2] - 1;
        flat = flat * b.nb[2] + iz;
    }
    return true;
}

// bin_index without early exits and with the number of dimensions known at compile time: `flat` is
// always a valid index (0 for a point outside), the return value says whether the point is inside --
// same comparisons, same arithmetic (the flat index fits 32 bits: checked by the launcher)
template <int ND>
__device__ __forceinline__ bool bin_index_flat(const DevBinning &b, double x, double y, double z,
                                               int &flat) {
    bool ok = x >= b.mins[0] && x < b.maxs[0];
    int ix = (int)((x - b.mins[0]) * b.norm[0]);
    ix = ix < b.nb[0] ? ix : b.nb[0] - 1;
    int f = ix;
    if (ND > 1) {
        ok = ok && (y >= b.mins[1] && y < b.maxs[1]);
        int iy = (int)((y - b.mins[1]) * b.norm[1]);
        iy = iy < b.nb[1] ? iy : b.nb[1] - 1;
        f = f * b.nb[1] + iy;
    }
    if (ND > 2) {
        ok = ok && (z >= b.mins[2] && z < b.maxs[2]);
        int iz = (int)((z - b.mins[2]) * b.norm[2]);
        iz = iz < b.nb[2] ? iz : b.nb[2] - 1;
        f = f * b.nb[2] + iz;
    }
    flat = ok ? f : 0;
    return ok;
}

__global__ void __launch_bounds__(256)
event_indices_kernel(const DevBinning b, const double *__restrict__ x,
                     const double *__restrict__ y, const double *__restrict__ z, int64_t n,
                     int32_t *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t flat;
    bool in = bin_index(b, x[i], b.ndim > 1 ? y[i] : 0.0, b.ndim > 2 ? z[i] : 0.0, flat);
    out[i] = in ? (int32_t)flat : -1;
}

// ---------------------------------------------------------------------------
struct ContDev {
    int64_t n;
    const double *gx, *gy, *flux, *aeff, *w0;
    const double *s[3];
    const int32_t *node, *bin;  // optional pre-digitised indices
    const int2 *node_bin;       // optional packed (node, bin) per event
    const double2 *aeff_w0;     // optional packed (weighted_aeff, initial_weights) per event
    const double2 *pepmu_own;   // optional per-container (P_e, P_mu) table (event-mode prob3)
    const double2 *wflux;       // optional static-weighted flux (w0*aeff*f_e, w0*aeff*f_mu) per event
    const uint32_t *idx16;      // optional 16-bit packed (node | bin << 16) per event, 0xffff = outside
    const double2 *wflux_q;     // wflux in quad-blocked order [q / 64][4][q % 64], padded to 256 events
    double scale;
    int32_t flav, side;
    // optional, 16-bit index form with a binning beyond the LDS accumulators: the resident order is cut into
    // n_part partitions, partition p = events [256 part_start[p], 256 part_start[p+1]) holding only deposits into
    // bins [p W, (p+1) W), W = the LDS window (HistArgs::window)
    const int32_t *part_start;
    int32_t n_part, part_width;
};

struct HistArgs {
    int32_t n_cont;
    int32_t cont_base;    // index of cont[0] in the output limb array
    int64_t n_bins;
    int64_t n_nodes;      // calc-grid nodes (stride of the (P_e,P_mu) tables)
    int64_t chunk;        // events per workgroup (multiple of 2*HIST_THREADS)
    DevBinning grid;      // calc grid (lookup)
    DevBinning outb;      // output binning
    const double *prob[2];   // P[node][3][3] for nu / nubar
    const double2 *pepmu;    // optional compact tables [side][flav][node] = (P_e->f, P_mu->f)
    ContDev cont[MAX_CONT];
    int32_t blk_start[MAX_CONT + 1];
    int32_t cont_chunk[MAX_CONT];   // > 0: events per workgroup of this container / 256 (instead of `chunk`)
    int32_t copies;       // LDS replicas of the accumulators (power of two), lane-interleaved
    int32_t opts;         // 1: the caller has no use for the second quantity (plain histogram without counts).
                          // Builds with -DPISA_DEV_PROBES only (PISA_HIP_HIST_DBG): 2 no deposits, 4 no flush
    int32_t window;       // > 0: LDS holds this many bins starting at the chunk's lowest bin
    // development builds (common.hpp, HandOver): the tables this launch gathers from are being written by a kernel that
    // runs BESIDE it; poll these counters before the first gather
    const unsigned long long *wait_flags;
    unsigned long long wait_epoch;
    int32_t wait_wgs;
};

// MODE 0: generic histogram (weights or counts; quantities (w, 1))
// MODE 1: fused reweight chain from coordinates (quantities (w, w^2))
// MODE 2: fused reweight chain from pre-digitised indices, 2 events / thread / sweep
// MODE 3: same from the packed 16-byte columns (node,bin) / (aeff,w0) / flux: 40 B per event
// MODE 5: compact form, (node,bin) + the flux pair pre-multiplied by the static per-event
//         factor initial_weights*weighted_aeff: 24 B per event, two loads fewer per pair.
//         w = ((g_e*P_e) + (g_mu*P_mu)) * scale  -- the reference's product with the static
//         factors associated first; differs from MODE 3 by rounding only (<= 3 ulp per weight)
// MODE 7: MODE 5 with the two indices in 16 bits each (grids and binnings below 65535 entries):
//         20 B per event, four events per thread and sweep so that every load stays 16 bytes
// DIMS (MODE 0, 1) = 4 * (dimensions of the calc grid, MODE 1) + (dimensions of the output binning)
// Development build (make EXTRA=-DPISA_HIST_STAMPS, scripts/dev/hist_stamps.py): wall-clock stamps of
// thread 0 of every workgroup at six points of the kernel.
#ifdef PISA_HIST_STAMPS
__device__ unsigned long long g_hist_stamps[8 * 1024];
#define STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_hist_stamps[8 * blockIdx.x + (k)] = wall_clock64(); } while (0)
#else
#define STAMP(k) do {} while (0)
#endif
template <int MODE, bool LDS_ACC, int DIMS = 0>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8)))
hist_accumulate_kernel(const HistArgs a, unsigned long long *__restrict__ g_limbs,
                       int32_t *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_acc[];  // [slab][quantity][bin], integer units
    STAMP(0);
    const int nthreads = blockDim.x;
    int c = 0;
    const int bid = blockIdx.x;
    while (c + 1 < a.n_cont && bid >= a.blk_start[c + 1]) c++;  // workgroup-uniform
    const ContDev &C = a.cont[c];
    const int64_t lb = bid - a.blk_start[c];
    // events of this workgroup where the workgroups of a container do not sweep together: the container's own
    // share (partitioned window order: multiples of 256 events), else the launch's largest share
    const int64_t my_chunk = a.cont_chunk[c] > 0 ? (int64_t)a.cont_chunk[c] * 256 : a.chunk;
    const int64_t start = lb * my_chunk;
    int64_t end = start + my_chunk;
    if (end > C.n) end = C.n;
    // bins held in LDS: all of them, or (a.window > 0, MODE 3 only) the window
    // [bin_lo, bin_lo + window) that starts at the smallest bin of this workgroup's
    // chunk -- a binning too large for LDS still gets LDS accumulation when the
    // events are stored in an order that keeps a chunk inside a few hundred
    // neighbouring bins; deposits outside the window go to the global limbs directly
    const int n_bins = a.window > 0 ? a.window : (int)a.n_bins;
    const int n_acc = NL * 2 * n_bins;
    // replica used by this lane: neighbouring lanes (neighbouring, i.e. correlated,
    // events) add into different copies, which cuts same-address serialisation
    unsigned long long *my_acc = s_acc + (LDS_ACC ? (int)(threadIdx.x & (a.copies - 1)) * n_acc : 0);
    int32_t *s_part = reinterpret_cast<int32_t *>(s_acc + 2 * (size_t)n_acc);   // partitioned order only (host: LDS size)
    unsigned long long *g_out = g_limbs + (int64_t)(a.cont_base + c) * a.n_bins * 2 * NL;
    int bin_lo = 0;

    // MODE 3 sweep order: the workgroups of a container advance through its columns
    // TOGETHER (workgroup j takes pairs j*T + t, stepping by n_wg*T), so that at any
    // moment the chip reads a few contiguous windows of HBM; the LDS window of a large
    // binning needs bin-contiguous chunks and keeps them.  The loads of the first sweep
    // are issued here, before the LDS accumulators are cleared, so that the clearing
    // overlaps the first HBM round trip.
    const bool together = a.window == 0;
    const int64_t n_wg = a.blk_start[c + 1] - a.blk_start[c];
    const int64_t step = together ? n_wg * nthreads : nthreads;
    const int64_t p_end = together ? (C.n >> 1) : (end >> 1);
    int64_t p = (together ? lb * nthreads : (start >> 1)) + threadIdx.x;
    constexpr bool QUAD = MODE == 7;
    bool have = !QUAD && p < p_end;
    int4 ix = make_int4(-1, -1, -1, -1);  // node0, bin0, node1, bin1
    double2 awa = make_double2(0.0, 0.0), awb = awa, fa = awa, fb = awa;
    constexpr bool PACKED = MODE == 3 || MODE == 5;
    constexpr bool COMPACT = MODE == 5;
    // MODE 7: quads of events (columns padded to whole blocks of 64 quads)
    const int64_t q_end = ((together ? C.n : end) + 3) >> 2;
    int64_t q = (together ? lb * nthreads : (start >> 2)) + threadIdx.x;
    bool qhave = QUAD && q < q_end;
    uint4 qx = make_uint4(~0u, ~0u, ~0u, ~0u);
    double2 g0 = awa, g1 = awa;
    if (QUAD && qhave) {
        qx = reinterpret_cast<const uint4 *>(C.idx16)[q];
        const double2 *gq = C.wflux_q + ((q >> 6) * 256 + (q & 63));
        g0 = gq[0]; g1 = gq[64];
    }
    if (PACKED && have) {
        const double2 *col = COMPACT ? C.wflux : C.aeff_w0;
        ix = reinterpret_cast<const int4 *>(C.node_bin)[p];
        awa = col[2 * p]; awb = col[2 * p + 1];
        if (!COMPACT) {
            fa = reinterpret_cast<const double2 *>(C.flux)[2 * p];
            fb = reinterpret_cast<const double2 *>(C.flux)[2 * p + 1];
        }
    }

    STAMP(1);
    if (LDS_ACC) {
        // (partitioned order: two windows -- a chunk may reach into a second partition -- and the partition
        // table in LDS: looking a partition up in global memory is a chain of dependent scalar loads, ~1 us each)
        const bool parts = QUAD && a.window > 0 && C.part_start;
        if (parts)   // behind the two windows (a strided fill: the table has up to PART_MAX + 1 entries whatever the block size)
            for (int k = threadIdx.x; k <= C.n_part; k += nthreads) s_part[k] = C.part_start[k];
        for (int k = threadIdx.x; k < n_acc * (parts ? 2 : a.copies); k += nthreads) s_acc[k] = 0ull;
        if ((PACKED || QUAD) && a.window > 0 && !(QUAD && C.part_start)) {
            __shared__ int s_lo;
            if (threadIdx.x == 0) s_lo = 0x7fffffff;
            __syncthreads();
            int m = 0x7fffffff;
            if (QUAD) {
                const uint4 *idxq = reinterpret_cast<const uint4 *>(C.idx16);
                for (int64_t qq = (start >> 2) + threadIdx.x; qq < q_end; qq += nthreads) {
                    const uint4 v = idxq[qq];
                    const int b[4] = {(int)(v.x >> 16), (int)(v.y >> 16), (int)(v.z >> 16), (int)(v.w >> 16)};
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (b[k] != 0xffff && b[k] < m) m = b[k];
                }
            } else {
            const int4 *idx4 = reinterpret_cast<const int4 *>(C.node_bin);
            for (int64_t p = (start >> 1) + threadIdx.x; p < (end >> 1); p += nthreads) {
                const int4 ix = idx4[p];
                if (ix.y >= 0 && ix.y < m) m = ix.y;
                if (ix.w >= 0 && ix.w < m) m = ix.w;
            }
            }
            for (int o = 32; o > 0; o >>= 1) {
                const int other = __shfl_xor(m, o);
                m = other < m ? other : m;
            }
            if ((threadIdx.x & 63) == 0 && m != 0x7fffffff) atomicMin(&s_lo, m);
            __syncthreads();
            bin_lo = s_lo == 0x7fffffff ? 0 : s_lo;
        }
        __syncthreads();
    }
    bool bad = false;
#ifdef PISA_DEV_PROBES
    if (a.wait_flags) {
        // consumer side of the hand-over (guide, "valid forms"): relaxed agent-scope polls by the first wavefront (one counter
        // per lane), ONE agent acquire, its wait, workgroup barrier, then plain loads of the tables
        if (threadIdx.x < HANDOVER_SLOTS) {
            const int k = (int)threadIdx.x;
            const unsigned long long need = a.wait_epoch * (unsigned long long)(a.wait_wgs / HANDOVER_SLOTS + (k < a.wait_wgs % HANDOVER_SLOTS ? 1 : 0));
            const unsigned long long t0 = wall_clock64();
            while (__hip_atomic_load(a.wait_flags + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                if (wall_clock64() - t0 > 100000000ull) {   // ~1 s of the 100 MHz counter: the producer is not coming
                    if (status) atomicOr(status, 8);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
#endif
    STAMP(2);

    auto accumulate = [&](int bin, double w, double w2) {
        const int rel = bin - bin_lo;
        const bool in_lds = LDS_ACC && (unsigned)rel < (unsigned)n_bins;
        // digits of slabs j, j-1, j-2 of one weight (deposit_units): LDS accumulators of this workgroup,
        // or -- bins outside the LDS window, binnings without LDS accumulators -- the global limbs
        auto put = [&](int qn, int j, unsigned long long d) {
            if (in_lds) lds_add(&my_acc[__mul24(j * 2 + qn, n_bins) + rel], d);
            else glb_add(&g_out[((int64_t)bin * 2 + qn) * NL + j], d);
        };
        auto add0 = [&](int j, long long d) { put(0, j, (unsigned long long)d); };
        auto add1 = [&](int j, long long d) { put(1, j, (unsigned long long)d); };
        auto add30 = [&](int j, unsigned long long d0, unsigned long long d1, unsigned long long d2) {
            if (in_lds) {
                const int i0 = __mul24(j * 2 + 0, n_bins) + rel;
                lds_add(&my_acc[i0], d0); lds_add(&my_acc[i0 - 2 * n_bins], d1); lds_add(&my_acc[i0 - 4 * n_bins], d2);
            } else {
                const int64_t i0 = ((int64_t)bin * 2 + 0) * NL + j;
                glb_add(&g_out[i0], d0); glb_add(&g_out[i0 - 1], d1); glb_add(&g_out[i0 - 2], d2);
            }
        };
        auto add31 = [&](int j, unsigned long long d0, unsigned long long d1, unsigned long long d2) {
            if (in_lds) {
                const int i0 = __mul24(j * 2 + 1, n_bins) + rel;
                lds_add(&my_acc[i0], d0); lds_add(&my_acc[i0 - 2 * n_bins], d1); lds_add(&my_acc[i0 - 4 * n_bins], d2);
            } else {
                const int64_t i0 = ((int64_t)bin * 2 + 1) * NL + j;
                glb_add(&g_out[i0], d0); glb_add(&g_out[i0 - 1], d1); glb_add(&g_out[i0 - 2], d2);
            }
        };
#ifdef PISA_DEV_PROBES
        if (a.opts & 2) {  // probe: keep the loads and the weight chain alive, no atomics
            if (w == 1.2345e-300 || w2 == 1.2345e-300) bad = true;
            return;
        }
#endif
        bool ok = deposit_units(w, add30, add0);
        if (!(a.opts & 1)) ok = deposit_units(w2, add31, add1) && ok;
        if (!ok) bad = true;
    };

    // slab accumulators -> integer units, added to the global limbs.  The loop runs in
    // GLOBAL order (limb fastest): a wave's 64 atomics fall into 512 contiguous bytes,
    // which the L2 atomic units take at full rate (scattered 96 B apart they do not).
    // Workgroups of a container finish together; each starts at a different offset so
    // that they do not all queue on the same limbs at the same moment.
    auto flush_win = [&](const unsigned long long *win, int lo, int ncopies) {
        const int rot = (int)((lb * 7 * 64) % n_acc);
        for (int g0 = threadIdx.x; g0 < n_acc; g0 += nthreads) {
            int g = g0 + rot;
            if (g >= n_acc) g -= n_acc;
            const int bin = g / (2 * NL);
            const int rem = g - bin * 2 * NL;
            const int q = rem / NL;
            const int j = rem - q * NL;
            const int k = (j * 2 + q) * n_bins + bin;
            unsigned long long v = win[k];
            for (int r = 1; r < ncopies; r++) v += win[r * n_acc + k];  // integer: exact
            if (v != 0ull && lo + bin < (int)a.n_bins) atomicAdd(&g_out[(int64_t)lo * 2 * NL + g], v);
        }
    };
    // partitioned order: the other LDS window (see the QUAD sweep) and the first bin of what it holds
    int other_lo = -1;

    if (QUAD) {
        const double2 *tab = C.pepmu_own ? C.pepmu_own
                                         : a.pepmu + ((int64_t)C.side * 3 + C.flav) * a.n_nodes;
        const double scale = C.scale;
        const uint4 *idxq = reinterpret_cast<const uint4 *>(C.idx16);
        const double2 *aw = C.wflux_q;
        int64_t qstep = together ? n_wg * nthreads : nthreads;
        const double2 zero2 = make_double2(0.0, 0.0);
        int64_t q_stop = q_end;
        // Partitioned resident order (binning beyond the LDS accumulators, ContDev::part_start): the chunk is
        // worked through partition by partition, each with the LDS window on that partition's bins -- every
        // deposit is an LDS deposit, nothing is scanned for the window's position, and only a chunk that
        // straddles a partition boundary flushes twice.  Partition boundaries are multiples of 256 events =
        // one wavefront's sweep.
        // A chunk that straddles a partition boundary keeps BOTH partitions' windows in LDS (2 x 63 KiB of the
        // CU's 160): its wavefronts move from one to the other on their own, no barrier, no intermediate flush
        // (in-kernel stamps: flushing, clearing and restarting in between left the straddling workgroups ~10 us
        // behind the others); a third partition in one chunk recycles the older window behind a barrier.
        int part = -1, phase = 0;
        if (LDS_ACC && a.window > 0 && C.part_start) {
            // the caller's table is taken on trust by the sweep below (an entry that decreases, or a last entry short of
            // the container, would end the loop early and DROP events without a trace): checked here, status bit 2
            if (threadIdx.x == 0 && lb == 0 && status) {
                bool ok = s_part[0] == 0 && (int64_t)s_part[C.n_part] * 256 >= C.n;
                for (int k = 0; k < C.n_part; k++) ok = ok && s_part[k] <= s_part[k + 1];
                if (!ok) atomicOr(status, 2);
            }
            part = 0;
            while (part + 1 < C.n_part && (int64_t)s_part[part + 1] * 256 <= start) part++;
        }
      for (;;) {
        if (part >= 0) {
            const int64_t ps = (int64_t)s_part[part] * 256, pe = (int64_t)s_part[part + 1] * 256;
            const int64_t lo = ps > start ? ps : start, hi = pe < end ? pe : end;
            if (phase >= 1) {
                unsigned long long *win = s_acc + (phase & 1) * n_acc;
                if (phase >= 2) {     // (workgroup-uniform: the chunk's geometry)
                    __syncthreads();
                    flush_win(win, other_lo, 1);
                    __syncthreads();
                    for (int k = threadIdx.x; k < n_acc; k += nthreads) win[k] = 0ull;
                    __syncthreads();
                }
                other_lo = bin_lo;
                my_acc = win;
            }
            bin_lo = part * a.window;
            q_stop = (hi + 3) >> 2;
            if (lo > start) {     // a later partition of this chunk: its own first loads
                q = (lo >> 2) + threadIdx.x;
                qhave = q < q_stop;
                if (qhave) {
                    qx = idxq[q];
                    const double2 *gq0 = aw + ((q >> 6) * 256 + (q & 63));
                    g0 = gq0[0]; g1 = gq0[64];
                }
            } else {
                qhave = q < q_stop;
            }
        }
        const int64_t q_end = q_stop;   // (shadows the chunk's end inside the sweep below)
        // The pair loop's software pipeline (see below), two half-sweeps per quad: while
        // events 0,1 of the quad are consumed the flux of events 2,3 is in flight, while 2,3
        // are consumed the next quad's indices and the flux of its events 0,1.
        while (qhave) {
            const double2 *gq = aw + ((q >> 6) * 256 + (q & 63));  // lane-contiguous 16-B loads
            {
                const unsigned n0 = qx.x & 0xffffu, n1 = qx.y & 0xffffu;
                const unsigned b0 = qx.x >> 16, b1 = qx.y >> 16;
                double2 p0 = tab[n0 == 0xffffu ? 0 : n0];
                double2 p1 = tab[n1 == 0xffffu ? 0 : n1];
                const double2 g2 = gq[128], g3 = gq[192];
                if (n0 == 0xffffu) p0 = zero2;
                if (n1 == 0xffffu) p1 = zero2;
                double w0 = ((g0.x * p0.x) + (g0.y * p0.y)) * scale;
                double w1 = ((g1.x * p1.x) + (g1.y * p1.y)) * scale;
                if (b0 == 0xffffu) w0 = 0.0;
                if (b1 == 0xffffu) w1 = 0.0;
                accumulate(b0 == 0xffffu ? 0 : (int)b0, w0, w0 * w0);
                accumulate(b1 == 0xffffu ? 0 : (int)b1, w1, w1 * w1);
                g0 = g2; g1 = g3;
            }
            const unsigned n2 = qx.z & 0xffffu, n3 = qx.w & 0xffffu;
            const unsigned b2 = qx.z >> 16, b3 = qx.w >> 16;
            double2 p2 = tab[n2 == 0xffffu ? 0 : n2];
            double2 p3 = tab[n3 == 0xffffu ? 0 : n3];
            const int64_t qn = q + qstep;
            const bool have_n = qn < q_end;
            const int64_t ql = have_n ? qn : q;  // unconditional loads (the last sweep re-reads its own quad)
            const uint4 qxn = idxq[ql];
            const double2 *gn = aw + ((ql >> 6) * 256 + (ql & 63));
            const double2 g0n = gn[0], g1n = gn[64];
            if (n2 == 0xffffu) p2 = zero2;
            if (n3 == 0xffffu) p3 = zero2;
            double w2 = ((g0.x * p2.x) + (g0.y * p2.y)) * scale;
            double w3 = ((g1.x * p3.x) + (g1.y * p3.y)) * scale;
            if (b2 == 0xffffu) w2 = 0.0;
            if (b3 == 0xffffu) w3 = 0.0;
            accumulate(b2 == 0xffffu ? 0 : (int)b2, w2, w2 * w2);
            accumulate(b3 == 0xffffu ? 0 : (int)b3, w3, w3 * w3);
            qx = qxn; g0 = g0n; g1 = g1n;
            q = qn;
            qhave = have_n;
        }
        if (part < 0) break;
        // next partition of this chunk, if any
        const int64_t pe = (int64_t)s_part[part + 1] * 256;
        if (pe >= end || part + 1 >= C.n_part) break;
        part++;
        phase++;
      }
    } else if (PACKED) {
        // packed columns: (node, bin) int2 and (aeff, w0) double2 per event; every load is 16 B
        const double2 *tab = C.pepmu_own ? C.pepmu_own
                                         : a.pepmu + ((int64_t)C.side * 3 + C.flav) * a.n_nodes;
        const double scale = C.scale;
        const int4 *idx4 = reinterpret_cast<const int4 *>(C.node_bin);
        const double2 *aw = COMPACT ? C.wflux : C.aeff_w0;
        const double2 *flux2 = reinterpret_cast<const double2 *>(C.flux);
        // Software-pipelined: the five streaming loads of the NEXT pair of events are
        // issued before the current pair is consumed, and the two dependent table
        // gathers of the current pair before them, so a wave always has a full
        // iteration of HBM requests in flight (a thread only runs ~6-10 iterations;
        // without this the idx -> gather -> use chain is exposed every time).
        while (have) {
            // node < 0 (outside the calc grid): read entry 0, then force P = 0
            double2 pa = tab[ix.x < 0 ? 0 : ix.x];
            double2 pb = tab[ix.z < 0 ? 0 : ix.z];
            const int64_t pn = p + step;
            const bool have_n = pn < p_end;
            // unconditional (the last sweep re-reads its own pair, never used): a
            // branch here would make the compiler wait for these loads as well
            const int64_t pl = have_n ? pn : p;
            const int4 ixn = idx4[pl];
            const double2 awan = aw[2 * pl], awbn = aw[2 * pl + 1];
            double2 fan = make_double2(0.0, 0.0), fbn = fan;
            if (!COMPACT) {
                fan = flux2[2 * pl];
                fbn = flux2[2 * pl + 1];
            }
            if (ix.x < 0) pa = make_double2(0.0, 0.0);
            if (ix.z < 0) pb = make_double2(0.0, 0.0);
            // branch-free on purpose (an event outside the binning deposits w = 0,
            // i.e. nothing): a conditional here lets the compiler sink the gathers
            // below the prefetch and wait for all of it
            double wa, wb;
            if (COMPACT) {
                wa = ((awa.x * pa.x) + (awa.y * pa.y)) * scale;
                wb = ((awb.x * pb.x) + (awb.y * pb.y)) * scale;
            } else {
                wa = awa.y * ((fa.x * pa.x) + (fa.y * pa.y));  // prob3.py:622
                wa = wa * (awa.x * scale);                     // aeff.py:87
                wb = awb.y * ((fb.x * pb.x) + (fb.y * pb.y));
                wb = wb * (awb.x * scale);
            }
            if (ix.y < 0) wa = 0.0;
            if (ix.w < 0) wb = 0.0;
            accumulate(ix.y < 0 ? bin_lo : ix.y, wa, wa * wa);
            accumulate(ix.w < 0 ? bin_lo : ix.w, wb, wb * wb);
            ix = ixn; awa = awan; awb = awbn; fa = fan; fb = fbn;
            p = pn;
            have = have_n;
        }
        if ((end & 1) && threadIdx.x == 0 && end > start) {  // odd tail of the container
            const int64_t i = end - 1;
            const int2 ix1 = C.node_bin[i];
            if (ix1.y >= 0) {
                double2 pp = ix1.x >= 0 ? tab[ix1.x] : make_double2(0.0, 0.0);
                double2 x = aw[i];
                double w;
                if (COMPACT) {
                    w = ((x.x * pp.x) + (x.y * pp.y)) * scale;
                } else {
                    double2 f = flux2[i];
                    w = x.y * ((f.x * pp.x) + (f.y * pp.y));
                    w = w * (x.x * scale);
                }
                accumulate(ix1.y, w, w * w);
            }
        }
    } else if (MODE == 2) {
        // (P_e, P_mu) of this container's class, one 16-B gather per event
        const double2 *tab = a.pepmu + ((int64_t)C.side * 3 + C.flav) * a.n_nodes;
        const double scale = C.scale;
        const int64_t p0 = start >> 1, p1 = end >> 1;  // whole pairs (start is even)
        const int2 *node2 = reinterpret_cast<const int2 *>(C.node);
        const int2 *bin2 = reinterpret_cast<const int2 *>(C.bin);
        const double2 *aeff2 = reinterpret_cast<const double2 *>(C.aeff);
        const double2 *w02 = reinterpret_cast<const double2 *>(C.w0);
        const double2 *flux2 = reinterpret_cast<const double2 *>(C.flux);
        for (int64_t p = p0 + threadIdx.x; p < p1; p += nthreads) {
            const int2 nd = node2[p];
            const int2 bn = bin2[p];
            const double2 ae = aeff2[p];
            const double2 w0 = w02[p];
            const double2 fa = flux2[2 * p], fb = flux2[2 * p + 1];
            double2 pa = make_double2(0.0, 0.0), pb = make_double2(0.0, 0.0);
            if (nd.x >= 0) pa = tab[nd.x];
            if (nd.y >= 0) pb = tab[nd.y];
            if (bn.x >= 0) {
                double w = w0.x * ((fa.x * pa.x) + (fa.y * pa.y));  // prob3.py:622
                w = w * (ae.x * scale);                             // aeff.py:87
                accumulate(bn.x, w, w * w);
            }
            if (bn.y >= 0) {
                double w = w0.y * ((fb.x * pb.x) + (fb.y * pb.y));
                w = w * (ae.y * scale);
                accumulate(bn.y, w, w * w);
            }
        }
        if ((end & 1) && threadIdx.x == 0 && end > start) {  // odd tail of the container
            const int64_t i = end - 1;
            const int nd = C.node[i], bn = C.bin[i];
            if (bn >= 0) {
                double2 pp = nd >= 0 ? tab[nd] : make_double2(0.0, 0.0);
                double2 f = flux2[i];
                double w = C.w0[i] * ((f.x * pp.x) + (f.y * pp.y));
                w = w * (C.aeff[i] * scale);
                accumulate(bn, w, w * w);
            }
        }
    } else {
        const double *prob = MODE == 1 ? a.prob[C.side] : nullptr;
        const int po_e = 0 * 3 + C.flav;   // P[e  -> flav]
        const int po_mu = 1 * 3 + C.flav;  // P[mu -> flav]
        if (MODE == 1) {
            // SURVEY section 8(d)'s unit of work: 72 B per event in eight column streams, both
            // digitisations in the kernel.  All streaming loads of an event are issued before anything
            // depends on one of them and nothing branches in between (an event outside the calc grid
            // gathers node 0 and multiplies by 0): one HBM round trip + one gather per sweep.  With the
            // lookup's early exits between the loads a sweep was three to four dependent round trips
            // (204 us for 10^7 events).
            constexpr int GD = DIMS >> 2, OD = DIMS & 3;
            constexpr bool g2 = GD > 1;
            const double2 *__restrict__ flux2 = reinterpret_cast<const double2 *>(C.flux);
            const double cscale = C.scale;
            // The gather is what this kernel waits for when it goes to the full matrices: 2.9 MB of
            // 72-byte records that the 720 MB column stream keeps pushing out of the L2 (all gathers
            // forced to one node: 202 -> 139 us for 10^7 events); the compact tables, 320 KB per class
            // and one 16-byte gather per event, are used whenever the caller provides them.
            const double2 *__restrict__ tab =
                a.pepmu ? a.pepmu + ((int64_t)C.side * 3 + C.flav) * a.n_nodes : nullptr;
            auto event = [&](double gx, double gy, double2 f, double w0, double ae, double x, double y, double z) {
                // grid -> event lookup of prob_e, prob_mu (container.py:981-1012,
                // translation.py:427-438): 0 outside the grid
                int node;
                const bool in = bin_index_flat<GD>(a.grid, gx, gy, 0.0, node);
                double pe, pmu;
                if (tab) {   // compact (P_e, P_mu) table of this container's class: one 16-byte gather
                    const double2 pp = tab[node];
                    pe = pp.x;
                    pmu = pp.y;
                } else {     // two 8-byte gathers from the 72-byte-stride P[node][3][3] table
                    pe = prob[9 * node + po_e];
                    pmu = prob[9 * node + po_mu];
                }
                pe = in ? pe : 0.0;
                pmu = in ? pmu : 0.0;
                double w = w0 * ((f.x * pe) + (f.y * pmu));   // prob3.py:622
                w = w * (ae * cscale);                        // aeff.py:87
                int bin;
                if (bin_index_flat<OD>(a.outb, x, y, z, bin)) accumulate(bin, w, w * w);
            };
            // every column 16-byte aligned (checked per workgroup, uniform): two consecutive events per
            // thread and sweep with 16-byte loads, the workgroups of a container sweeping its columns
            // together like the packed forms
            uintptr_t bits = (uintptr_t)C.gx | (uintptr_t)C.flux | (uintptr_t)C.w0 | (uintptr_t)C.aeff | (uintptr_t)C.s[0];
            if (g2) bits |= (uintptr_t)C.gy;
            if (OD > 1) bits |= (uintptr_t)C.s[1];
            if (OD > 2) bits |= (uintptr_t)C.s[2];
            if ((bits & 15) == 0) {
                const double2 *__restrict__ gx2 = reinterpret_cast<const double2 *>(C.gx);
                const double2 *__restrict__ gy2 = reinterpret_cast<const double2 *>(C.gy);
                const double2 *__restrict__ w02 = reinterpret_cast<const double2 *>(C.w0);
                const double2 *__restrict__ ae2 = reinterpret_cast<const double2 *>(C.aeff);
                const double2 *__restrict__ x2 = reinterpret_cast<const double2 *>(C.s[0]);
                const double2 *__restrict__ y2 = reinterpret_cast<const double2 *>(C.s[1]);
                const double2 *__restrict__ z2 = reinterpret_cast<const double2 *>(C.s[2]);
                const double2 zero = make_double2(0.0, 0.0);
                const int64_t n_wg1 = a.blk_start[c + 1] - a.blk_start[c];
                for (int64_t q = lb * nthreads + threadIdx.x; q < (C.n >> 1); q += n_wg1 * nthreads) {
                    const double2 gx = gx2[q];
                    const double2 gy = g2 ? gy2[q] : zero;
                    const double2 fa = flux2[2 * q], fb = flux2[2 * q + 1];
                    const double2 w0 = w02[q], ae = ae2[q];
                    const double2 x = x2[q];
                    const double2 y = OD > 1 ? y2[q] : zero;
                    const double2 z = OD > 2 ? z2[q] : zero;
                    event(gx.x, gy.x, fa, w0.x, ae.x, x.x, y.x, z.x);
                    event(gx.y, gy.y, fb, w0.y, ae.y, x.y, y.y, z.y);
                }
                if ((C.n & 1) && lb == 0 && threadIdx.x == 0) {  // odd tail of the container
                    const int64_t i = C.n - 1;
                    event(C.gx[i], g2 ? C.gy[i] : 0.0, flux2[i], C.w0[i], C.aeff[i], C.s[0][i],
                          OD > 1 ? C.s[1][i] : 0.0, OD > 2 ? C.s[2][i] : 0.0);
                }
            } else {
                for (int64_t i = start + threadIdx.x; i < end; i += nthreads)
                    event(C.gx[i], g2 ? C.gy[i] : 0.0, flux2[i], C.w0[i], C.aeff[i], C.s[0][i],
                          OD > 1 ? C.s[1][i] : 0.0, OD > 2 ? C.s[2][i] : 0.0);
            }
        } else {
            // MODE 0, generic histogram (DIMS = dimensions of the binning): like the coordinate form,
            // pairs of events with 16-byte loads in the together sweep when the columns allow it
            constexpr int OD = DIMS & 3;
            const bool weighted = C.w0 != nullptr;
            auto event = [&](double w, double x, double y, double z) {
                int bin;
                if (bin_index_flat<OD>(a.outb, x, y, z, bin)) accumulate(bin, w, 1.0);
            };
            uintptr_t bits = (uintptr_t)C.w0 | (uintptr_t)C.s[0];
            if (OD > 1) bits |= (uintptr_t)C.s[1];
            if (OD > 2) bits |= (uintptr_t)C.s[2];
            if ((bits & 15) == 0) {
                const double2 *__restrict__ w2 = reinterpret_cast<const double2 *>(C.w0);
                const double2 *__restrict__ x2 = reinterpret_cast<const double2 *>(C.s[0]);
                const double2 *__restrict__ y2 = reinterpret_cast<const double2 *>(C.s[1]);
                const double2 *__restrict__ z2 = reinterpret_cast<const double2 *>(C.s[2]);
                const double2 zero = make_double2(0.0, 0.0), one = make_double2(1.0, 1.0);
                const int64_t n_wg1 = a.blk_start[c + 1] - a.blk_start[c];
                for (int64_t q = lb * nthreads + threadIdx.x; q < (C.n >> 1); q += n_wg1 * nthreads) {
                    const double2 w = weighted ? w2[q] : one;
                    const double2 x = x2[q];
                    const double2 y = OD > 1 ? y2[q] : zero;
                    const double2 z = OD > 2 ? z2[q] : zero;
                    event(w.x, x.x, y.x, z.x);
                    event(w.y, x.y, y.y, z.y);
                }
                if ((C.n & 1) && lb == 0 && threadIdx.x == 0) {  // odd tail
                    const int64_t i = C.n - 1;
                    event(weighted ? C.w0[i] : 1.0, C.s[0][i], OD > 1 ? C.s[1][i] : 0.0, OD > 2 ? C.s[2][i] : 0.0);
                }
            } else {
                for (int64_t i = start + threadIdx.x; i < end; i += nthreads)
                    event(weighted ? C.w0[i] : 1.0, C.s[0][i], OD > 1 ? C.s[1][i] : 0.0, OD > 2 ? C.s[2][i] : 0.0);
            }
        }
    }
    if (bad && status) atomicOr(status, 1);
    STAMP(3);

#ifdef PISA_DEV_PROBES
    if (a.opts & 4) return;
#endif
    if (LDS_ACC) {
        __syncthreads();
        STAMP(4);
        flush_win(my_acc == s_acc || a.copies > 1 ? s_acc : s_acc + n_acc, bin_lo, a.copies);
        if (other_lo >= 0) flush_win(my_acc == s_acc ? s_acc + n_acc : s_acc, other_lo, 1);
    }
    STAMP(5);
}


// ---------------------------------------------------------------------------
// SEVERAL PARAMETER POINTS IN ONE SWEEP OF THE EVENTS (pisa_hip_reweight_hist_multi).
// A fit with a finite-difference minimiser asks for n + 1 independent points per gradient
// (pisa/analysis/analysis.py:2493-2670 driven by settings/minimizer/l-bfgs-b_*; the reference's own
// benchmark protocol, pisa/scripts/benchmark_pipeline_performance.py:196-223, times independent points
// as well).  The 20 B per event of the 16-bit index form do not depend on the point: this kernel reads
// them ONCE, fetches the event's (P_e, P_mu) pairs of all KP points in one contiguous run of the
// interleaved tables [sign][flavour][node][point], and deposits KP weights into KP sets of LDS
// accumulators.  Per point the weight and its exact three-piece deposit are the single-point kernel's
// (`deposit`, `slab_to_units`), so the limbs of every point are bit-identical to a
// pisa_hip_reweight_hist call at that point (tests/test_gpu_multipoint.py).
constexpr int MULTI_KP_MAX = 8;   // points per launch (a larger batch is split into passes)

struct MultiCont {
    int64_t n;
    const uint32_t *idx16;
    const double2 *wflux_q;
    int32_t flav, side;
};

struct MultiArgs {
    int32_t n_cont, cont_base;
    int32_t n_bins, k_stride;   // k_stride: points interleaved in the tables
    int64_t n_nodes;
    int64_t limb_stride;        // limbs of one point (in 8-byte words)
    const double2 *pepmu;       // tables + index of this launch's first point
    MultiCont cont[MAX_CONT];
    int32_t blk_start[MAX_CONT + 1];
    int32_t pad;
    double scale[MULTI_KP_MAX][MAX_CONT];   // aeff scale per (point, container)
};

template <int KP>
__global__ void __launch_bounds__(1024)
hist_accumulate_multi_kernel(const MultiArgs a, unsigned long long *__restrict__ g_limbs,
                             int32_t *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_units[];  // [point][slab][quantity][bin], integer units
    const int nthreads = blockDim.x;
    int c = 0;
    const int bid = blockIdx.x;
    while (c + 1 < a.n_cont && bid >= a.blk_start[c + 1]) c++;  // workgroup-uniform
    const MultiCont &C = a.cont[c];
    const int64_t lb = bid - a.blk_start[c];
    const int n_bins = a.n_bins;
    const int n_acc = NL * 2 * n_bins;
    // the workgroups of a container sweep its columns together (see hist_accumulate_kernel)
    const int64_t n_wg = a.blk_start[c + 1] - a.blk_start[c];
    const int64_t qstep = n_wg * nthreads;
    const int64_t q_end = (C.n + 3) >> 2;
    int64_t q = lb * nthreads + threadIdx.x;
    const uint4 *idxq = reinterpret_cast<const uint4 *>(C.idx16);
    const uint4 none = make_uint4(~0u, ~0u, ~0u, ~0u);
    const double2 zero2 = make_double2(0.0, 0.0);
    const double2 *tab = a.pepmu + ((int64_t)C.side * 3 + C.flav) * a.n_nodes * a.k_stride;
    const int ks = a.k_stride;
    // A wavefront takes 256 consecutive events per sweep (a quad per lane).  In the engine's resident
    // order (engine.deposit_block_order) such a block either holds depositing events or none at all, so
    // "does any lane of this wavefront have an event inside the binning" is the wave-uniform question
    // that decides whether the block's flux pairs and table entries are needed at all.
    auto any_in = [](const uint4 &v) {
        const bool in = ((v.x >> 16) != 0xffffu) | ((v.y >> 16) != 0xffffu) | ((v.z >> 16) != 0xffffu) |
                        ((v.w >> 16) != 0xffffu);
        return __any(in) != 0;
    };
    auto gather = [&](unsigned word, double2 (&p)[KP]) {
        // node 0xffff (outside the calc grid): entry 0 is read and replaced by P = 0 when it is used
        const unsigned node = word & 0xffffu;
        const double2 *tp = tab + (int64_t)(node == 0xffffu ? 0u : node) * ks;
#pragma unroll
        for (int k = 0; k < KP; k++) p[k] = tp[k];
    };
    auto load_flux = [&](int64_t qq, double2 (&gg)[4]) {
        const double2 *gq = C.wflux_q + ((qq >> 6) * 256 + (qq & 63));
        gg[0] = gq[0]; gg[1] = gq[64]; gg[2] = gq[128]; gg[3] = gq[192];
    };
    // Software pipeline over the sweeps i = 0, 1, ... of this thread (what a wavefront waits for is
    // memory -- 16 wavefronts per CU, nothing else to switch to): while sweep i is deposited, the index
    // words of sweep i + 2, the flux pairs of sweep i + 1 and the table entries of the NEXT event are in
    // flight.  qx / g / p: sweep i (p: its first event); qx1: sweep i + 1.
    uint4 qx = none, qx1 = none;
    double2 g[4] = {zero2, zero2, zero2, zero2};
    double2 p[KP];
#pragma unroll
    for (int k = 0; k < KP; k++) p[k] = zero2;
    bool have = q < q_end;
    if (have) qx = idxq[q];
    if (q + qstep < q_end) qx1 = idxq[q + qstep];
    bool in0 = any_in(qx);
    if (in0) {
        load_flux(q, g);
        gather(qx.x, p);
    }
    for (int k = threadIdx.x; k < n_acc * KP; k += nthreads) s_units[k] = 0ull;
    __syncthreads();
    double sc[KP];
#pragma unroll
    for (int k = 0; k < KP; k++) sc[k] = a.scale[k][c];
    bool bad = false;
    while (have) {
        const int64_t q1 = q + qstep, q2 = q1 + qstep;
        uint4 qx2 = none;
        if (q2 < q_end) qx2 = idxq[q2];
        const bool in1 = any_in(qx1);      // (lanes beyond the end hold `none`)
        double2 gn[4] = {zero2, zero2, zero2, zero2};
        if (in1) load_flux(q1, gn);
        if (in0) {
            const unsigned w4[5] = {qx.x, qx.y, qx.z, qx.w, qx1.x};
            double2 pn[KP];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                // table entries of the next event (for e = 3: the first event of the next sweep)
                if (e < 3 || in1) gather(w4[e + 1], pn);
                const unsigned node = w4[e] & 0xffffu, bin = w4[e] >> 16;
                if (bin != 0xffffu) {     // outside the binning: w = 0 for every point, nothing to deposit
#pragma unroll
                    for (int k = 0; k < KP; k++) {
                        // outside the calc grid: P = 0; the products are still formed, as in the single-point
                        // kernel (a non-finite flux is flagged there too)
                        const double2 pk = node != 0xffffu ? p[k] : zero2;
                        const double w = ((g[e].x * pk.x) + (g[e].y * pk.y)) * sc[k];
                        unsigned long long *acc = s_units + k * n_acc + (int)bin;
                        auto add0 = [&](int j, long long d) { atomicAdd(&acc[(j * 2 + 0) * n_bins], (unsigned long long)d); };
                        auto add1 = [&](int j, long long d) { atomicAdd(&acc[(j * 2 + 1) * n_bins], (unsigned long long)d); };
                        auto add30 = [&](int j, unsigned long long d0, unsigned long long d1, unsigned long long d2) {
                            unsigned long long *t0 = acc + __mul24(j * 2 + 0, n_bins);
                            atomicAdd(t0, d0);
                            atomicAdd(t0 - 2 * n_bins, d1);
                            atomicAdd(t0 - 4 * n_bins, d2);
                        };
                        auto add31 = [&](int j, unsigned long long d0, unsigned long long d1, unsigned long long d2) {
                            unsigned long long *t0 = acc + __mul24(j * 2 + 1, n_bins);
                            atomicAdd(t0, d0);
                            atomicAdd(t0 - 2 * n_bins, d1);
                            atomicAdd(t0 - 4 * n_bins, d2);
                        };
                        bool ok = deposit_units(w, add30, add0);
                        ok = deposit_units(w * w, add31, add1) && ok;
                        if (!ok) bad = true;
                    }
                }
#pragma unroll
                for (int k = 0; k < KP; k++) p[k] = pn[k];
            }
        } else if (in1) {
            gather(qx1.x, p);
        }
        qx = qx1; qx1 = qx2;
#pragma unroll
        for (int e = 0; e < 4; e++) g[e] = gn[e];
        in0 = in1;
        q = q1;
        have = q < q_end;
    }
    if (bad && status) atomicOr(status, 1);
    __syncthreads();
    // slab accumulators -> integer units, added to the points' global limbs (global order, rotated
    // start: see hist_accumulate_kernel)
    const int rot = (int)((lb * 7 * 64) % n_acc);
#pragma unroll 1
    for (int k = 0; k < KP; k++) {
        unsigned long long *g_out = g_limbs + (int64_t)k * a.limb_stride +
                                    (int64_t)(a.cont_base + c) * n_bins * 2 * NL;
        const unsigned long long *acc = s_units + k * n_acc;
        for (int g0 = threadIdx.x; g0 < n_acc; g0 += nthreads) {
            int gi = g0 + rot;
            if (gi >= n_acc) gi -= n_acc;
            const int bin = gi / (2 * NL);
            const int rem = gi - bin * 2 * NL;
            const int qq = rem / NL;
            const int j = rem - qq * NL;
            const unsigned long long v = acc[(j * 2 + qq) * n_bins + bin];
            if (v != 0ull) atomicAdd(&g_out[gi], v);
        }
    }
}

#ifdef PISA_HIST_STAMPS
PISA_API int pisa_hip_debug_hist_stamps(unsigned long long *h_out) {
    return check_hip(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_hist_stamps), sizeof(g_hist_stamps)), "stamps");
}
#endif

// limbs (possibly summed over workgroups and ranks, un-normalised) -> fp64,
// rounded once (RNE).  Sets *ovf if the value does not fit the format.
__device__ __forceinline__ double limbs_to_double(const long long *in, bool &ovf) {
    long long L[NL + 1];
    long long carry = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        long long v = in[k] + carry;
        carry = v >> 32;  // arithmetic shift = floor division
        L[k] = v & 0xffffffffLL;
    }
    L[NL] = carry;  // signed remainder
    bool neg = L[NL] < 0;
    if (neg) {
        long long c2 = 1;  // two's complement negate of the (NL+1)-limb number
#pragma unroll
        for (int k = 0; k < NL; k++) {
            long long v = (0xffffffffLL - L[k]) + c2;
            c2 = v >> 32;
            L[k] = v & 0xffffffffLL;
        }
        L[NL] = ~L[NL] + c2;
    }
    if (L[NL] != 0) ovf = true;
    // the NL digits as three 64-bit words; the highest non-zero word and the two below it, shifted so
    // that the value's leading bit is bit 63 of `top`: 64 significant bits and a sticky bit for the rest
    // (no indexing of the digit array with a run-time index: that costs a select chain per access)
    static_assert(NL == 6, "three 64-bit words");
    const unsigned long long w0 = (unsigned long long)L[0] | ((unsigned long long)L[1] << 32);
    const unsigned long long w1 = (unsigned long long)L[2] | ((unsigned long long)L[3] << 32);
    const unsigned long long w2 = (unsigned long long)L[4] | ((unsigned long long)L[5] << 32);
    if ((w0 | w1 | w2) == 0ULL) return 0.0;
    unsigned long long a_, b_, c_;
    int base;
    if (w2) { a_ = w2; b_ = w1; c_ = w0; base = 128; }
    else if (w1) { a_ = w1; b_ = w0; c_ = 0ULL; base = 64; }
    else { a_ = w0; b_ = 0ULL; c_ = 0ULL; base = 0; }
    const int s_ = __clzll(a_);   // 0 .. 63
    unsigned long long top = a_, lost = b_ | c_;
    if (s_) {
        top = (a_ << s_) | (b_ >> (64 - s_));
        lost = (b_ << s_) | c_;   // (only whether anything is left matters)
    }
    if (lost != 0ULL) top |= 1ULL;   // sticky bit: sits below the rounding position (bit 11 of `top`)
    double d = (double)top;           // u64 -> f64 is round-to-nearest-even
    d = ldexp(d, base - s_ - FX_LSB);
    return neg ? -d : d;
}

__global__ void __launch_bounds__(256)
hist_finalize_kernel(const long long *__restrict__ limbs, int64_t n_total_bins,
                     double *__restrict__ hist, double *__restrict__ q1,
                     int32_t *__restrict__ status) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_total_bins) return;
    bool ovf = false;
    if (hist) hist[b] = limbs_to_double(limbs + (b * 2 + 0) * NL, ovf);
    if (q1) q1[b] = limbs_to_double(limbs + (b * 2 + 1) * NL, ovf);
    if (ovf && status) atomicOr(status, 1);
}

// hist_finalize_kernel + metric_kernel (metric_flux.hip) in one workgroup: the
// tail of a template evaluation is launch-bound, not work-bound.  The metric part
// repeats metric_kernel's loop and reduction tree exactly (first 256 threads).
// One instantiation per metric: with a run-time `kind` the lgamma of poisson_llh sets the
// register need of every variant and the 128-VGPR budget of a 1024-thread workgroup spills.
// SPLIT = 4: FOUR workgroups per point, workgroup k takes the bins b = k mod 4 of every container (a quarter of the
// conversions -- what the one-workgroup form spends its time on: 3 per thread at the headline size) and leaves the
// partial sum u_k of the metric over ITS bins in total[4 * point + k].  The reduction tree below adds thread t to
// t + 128, t + 64, lane to lane + 32 .. + 4 -- all multiples of 4 -- before it joins the four residue classes with
// lane + 2 and lane + 1: u_k is exactly what lane k holds in the one-workgroup form before those last two steps, and
//       total = (u_0 + u_2) + (u_1 + u_3)
// added by the caller is the one-workgroup result bit for bit.  Not for chi2 (its all-bins-equal rule needs every bin).
template <int KIND, int SPLIT = 1>
__global__ void __launch_bounds__(1024)
finalize_metric_kernel(long long *__restrict__ limbs, int n_cont, int n_bins,
                       double *__restrict__ hist, double *__restrict__ q1,
                       const double *__restrict__ actual, double *__restrict__ total,
                       int32_t *__restrict__ status, int32_t *__restrict__ mstatus, int clear,
                       const double *__restrict__ scale, const double *__restrict__ extra,
                       int64_t limb_stride, int64_t scale_stride) {
    extern __shared__ __attribute__((aligned(16))) double s_map[];  // [2][n_cont][n_bins]
    __shared__ double s_sum[256];
    __shared__ int s_flag[2];
    constexpr int kind = KIND;
    const int n_tot = n_cont * n_bins;
    const int part = SPLIT > 1 ? (int)(blockIdx.x % SPLIT) : 0;   // this workgroup's bins: b = part mod SPLIT
    {
        // one workgroup (SPLIT of them) per parameter point (pisa_hip_finalize_metric_multi; a single point: 0)
        const int64_t pt = blockIdx.x / SPLIT;
        limbs += pt * limb_stride;
        hist += pt * n_tot;
        q1 += pt * n_tot;
        total += pt * SPLIT;
        if (scale) scale += pt * scale_stride;
    }
    if (threadIdx.x < 2) s_flag[threadIdx.x] = 0;
    // first bin's observed count: requested before anything else, used after the barrier
    double k_first = 0.0;
    if (threadIdx.x < 256 && (int)threadIdx.x < n_bins) k_first = actual[threadIdx.x];
    bool ovf = false;
    // One item = the NL limbs of one (bin, quantity) sum; item `it` sits at limbs + it*NL, so
    // a wave reads one contiguous run.  The limbs were produced by device-scope atomics and
    // come from beyond the L2: every load of (up to) four items per thread is issued before
    // the first conversion, so the kernel pays that latency once.
    // SPLIT > 1: the workgroup's items are numbered l = 2 (container * nb + j) + quantity over ITS nb bins j * SPLIT + part
    const int nb = SPLIT > 1 ? (n_bins - part + SPLIT - 1) / SPLIT : n_bins;
    const int n_items = 2 * n_cont * (nb > 0 ? nb : 0);
    auto item_of = [&](int l) {
        if (SPLIT == 1) return l;
        const int jq = l >> 1, cont = jq / nb, j = jq - cont * nb;
        return 2 * (cont * n_bins + j * SPLIT + part) + (l & 1);
    };
    constexpr int UNR = 4;
    for (int base = 0; base < n_items; base += UNR * (int)blockDim.x) {
        long long w[UNR][NL];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const int l = base + u * (int)blockDim.x + (int)threadIdx.x;
            const long long *L = limbs + (int64_t)(l < n_items ? item_of(l) : 0) * NL;
#pragma unroll
            for (int k = 0; k < NL; k++) w[u][k] = L[k];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const int l = base + u * (int)blockDim.x + (int)threadIdx.x;
            const int it = l < n_items ? item_of(l) : 0;
            if (l < n_items) {
                if (clear) {
                    long long *L = limbs + (int64_t)it * NL;
#pragma unroll
                    for (int k = 0; k < NL; k++) L[k] = 0;
                }
                const int i = it >> 1, q = it & 1;
                const double d = limbs_to_double(w[u], ovf);
                (q ? q1 : hist)[i] = d;   // the maps are written as histogrammed
                double v = d;
                if (scale) {
                    // per-bin scale factors of a stage that follows the histogram
                    // (discr_sys.hypersurfaces: weights = clip(weights * s, 0, inf), errors *= s;
                    // hypersurfaces.py:251-259) enter the metric's expectation only
                    const double sc = scale[i];
                    const double e = sqrt(d) * sc;   // errors = sqrt(sumw2) * s, variance = errors^2
                    v = q ? e * e : fmax(d * sc, 0.0);
                }
                s_map[q * n_tot + i] = v;
            }
            // conversions one after the other: interleaved they exceed the 128-VGPR budget
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (ovf && status) atomicOr(status, 1);
    __syncthreads();
    const double *expected = s_map, *sigma2 = s_map + n_tot;
    double acc = 0.0;
    if (threadIdx.x < 256) {
        // (SPLIT > 1: b mod SPLIT = thread mod SPLIT -- the threads of the other residue classes have no bin here)
        for (int b = threadIdx.x; b < n_bins && (SPLIT == 1 || (int)(threadIdx.x % SPLIT) == part); b += 256) {
            const double k = (b == (int)threadIdx.x) ? k_first : actual[b];
            double lam = 0.0, s2 = 0.0;
            for (int m = 0; m < n_cont; m++) {
                lam = (m == 0) ? expected[b] : lam + expected[m * n_bins + b];
                s2 = (m == 0) ? sigma2[b] : s2 + sigma2[m * n_bins + b];
            }
            if (extra) {   // maps of other pipelines added to the template (distribution_maker.py:274-281)
                lam += extra[b];
                s2 += extra[n_bins + b];
            }
            double v;
            const bool finite = (k == k) && (lam == lam) && !isinf(k) && !isinf(lam);
            if (!finite) {
                v = __longlong_as_double(0x7ff8000000000000LL);
            } else {
                if (k < 0.0 || lam < 0.0) atomicOr(&s_flag[0], 1);
                if (kind == PISA_HIP_METRIC_CHI2) {
                    const double lc = lam < SMALL_POS ? SMALL_POS : lam;
                    if (!(fabs(k - lc) < 5 * FTYPE_PREC)) atomicOr(&s_flag[1], 1);
                }
                v = metric_bin(kind, k, lam, s2);
            }
            if (v == v) acc += v;  // np.nansum
        }
        s_sum[threadIdx.x] = acc;
    }
    __syncthreads();
    // metric_kernel's reduction tree (s_sum[t] += s_sum[t + off], off = 128 .. 1): the same
    // additions in the same pairing, the last six levels inside wave 0 with lane shuffles
    // instead of LDS round trips and workgroup barriers
    if (threadIdx.x < 128) s_sum[threadIdx.x] += s_sum[threadIdx.x + 128];
    __syncthreads();
    if (threadIdx.x < 64) {
        double v = s_sum[threadIdx.x] + s_sum[threadIdx.x + 64];
#pragma unroll
        for (int off = 32; off >= SPLIT; off >>= 1) v += __shfl_down(v, off);
        if (SPLIT > 1) {
            if ((int)threadIdx.x == part) {
                const bool negative = mstatus && s_flag[0];
                total[part] = negative ? __builtin_nan("") : v;
                if (negative) mstatus[0] = PISA_HIP_ERR_NEGATIVE;
            }
        } else if (threadIdx.x == 0) {
            const bool chi2_zero = (kind == PISA_HIP_METRIC_CHI2) && s_flag[1] == 0;  // stats.py:160-161
            // negative input (stats.py:231-240 raises): the value is NaN as well, so that a host
            // that polls `total` in pinned memory needs to read the status word only then
            const bool negative = mstatus && s_flag[0];
            total[0] = negative ? __builtin_nan("") : (chi2_zero ? 0.0 : v);
            if (negative) mstatus[0] = PISA_HIP_ERR_NEGATIVE;
        }
    }
}

__global__ void __launch_bounds__(256)
hist_average_kernel(int64_t n_bins, double *__restrict__ hist, const double *__restrict__ cnt) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_bins) return;
    double v = hist[b] / cnt[b];  // translation.py:118-127: x/0 -> nan -> 0
    if (!(v == v)) v = 0.0;
    if (isinf(v)) v = v > 0 ? 1.7976931348623157e308 : -1.7976931348623157e308;
    hist[b] = v;
}

__global__ void __launch_bounds__(256)
lookup_regular_kernel(const DevBinning b, const double *__restrict__ x,
                      const double *__restrict__ y, const double *__restrict__ z, int64_t n,
                      const double *__restrict__ flat_hist, int width, double *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t node;
    bool in = bin_index(b, x[i], b.ndim > 1 ? y[i] : 0.0, b.ndim > 2 ? z[i] : 0.0, node);
    for (int w = 0; w < width; w++) out[i * width + w] = in ? flat_hist[node * width + w] : 0.0;
}

__global__ void __launch_bounds__(256)
apply_osc_weights_kernel(const double *__restrict__ flux, const double *__restrict__ pe,
                         const double *__restrict__ pmu, int64_t n, double *__restrict__ w) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double2 f = reinterpret_cast<const double2 *>(flux)[i];
    w[i] = w[i] * ((f.x * pe[i]) + (f.y * pmu[i]));
}

// the same with the two probabilities read at an element stride (columns of one table, e.g. the (P_e, P_mu) pairs
// of the gather tables: stride 2)
__global__ void __launch_bounds__(256)
apply_osc_weights_strided_kernel(const double *__restrict__ flux, const double *__restrict__ pe,
                                 const double *__restrict__ pmu, int64_t stride, int64_t n, double *__restrict__ w) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double2 f = reinterpret_cast<const double2 *>(flux)[i];
    w[i] = w[i] * ((f.x * pe[i * stride]) + (f.y * pmu[i * stride]));
}

// The reference's event-wise (or bin-wise) weight chain of several containers in ONE launch:
//   weights = copy(initial_weights)                        toy_event_generator.py:101-104 / the loaders
//   weights *= flux[:,0]*prob_e + flux[:,1]*prob_mu        prob3.py:621-622
//   weights *= weighted_aeff * scale                       aeff.py:87
// each step rounded to fp64 exactly as the one-step kernels above do (this file is compiled with contraction off).
constexpr int CHAIN_MAX_SETS = 24;
struct ChainArgs {
    pisa_hip_chain_set set[CHAIN_MAX_SETS];
};

__global__ void __launch_bounds__(256)
weight_chain_multi_kernel(const ChainArgs a) {
    const pisa_hip_chain_set &S = a.set[blockIdx.y];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S.n) return;
    double w = S.d_initial_weights[i];
    if (S.d_nu_flux) {
        const double2 f = reinterpret_cast<const double2 *>(S.d_nu_flux)[i];
        w = w * ((f.x * S.d_prob_e[i * S.prob_stride]) + (f.y * S.d_prob_mu[i * S.prob_stride]));
    }
    if (S.d_weighted_aeff) w = w * (S.d_weighted_aeff[i] * S.aeff_scale);
    S.d_weights[i] = w;
}

__global__ void __launch_bounds__(256)
apply_aeff_kernel(const double *__restrict__ aeff, double scale, int64_t n,
                  double *__restrict__ w) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    w[i] = w[i] * (aeff[i] * scale);
}

// ---------------------------------------------------------------- host side
static int64_t lds_acc_bytes(int64_t n_bins) { return n_bins * 2 * NL * 8; }

// Workgroups of an accumulate launch.  ONE 1024-thread workgroup per CU (`target` = the device's CU count unless
// PISA_HIP_HIST_BLOCKS says otherwise), never one more: with two per CU the SIMDs serve the older one first, the
// younger ones finish 6-8 us later on their own at half the chip's request rate (10^7 events: 40.1 -> 37.0 us by
// HIP events; 3 / 4 * 10^7 events beyond the Infinity Cache: 137 -> 111 / 157 -> 144 us), and a handful of workgroups
// beyond what is resident run as a second round after everything else (264 workgroups: 50 us).  The workgroups are
// dealt to the containers so that the largest share of a workgroup is as small as possible (greedy: the next
// workgroup goes to the container with the most events per workgroup), a workgroup gets at least one sweep
// (4 * threads events) and at most 2^28 events (the 64-bit accumulators take 2^31 32-bit digits).
// chunk = the largest share, a whole number of sweeps: the forms that do not sweep together cut the columns by it.
static int device_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = 256;
        return v;
    }();
    return n;
}

// LDS a workgroup of the current device may ask for (opt-in limit; 160 KiB on gfx950, 64 KiB on older parts)
static int64_t device_lds_bytes() {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 64 * 1024;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeSharedMemPerBlockOptin, dev) != hipSuccess || v <= 0)
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0)
            v = 64 * 1024;
    return v;
}

static int plan_blocks_balanced(const int64_t *n_events, int n_cont, int threads, int64_t target, int64_t &chunk,
                                int32_t *blk_start) {
    const int64_t sweep = 4 * (int64_t)threads, wg_max = 1LL << 28;
    int64_t nwg[MAX_CONT], cap[MAX_CONT], used = 0;
    for (int c = 0; c < n_cont; c++) {
        const int64_t n = n_events[c];
        cap[c] = n > 0 ? (n + sweep - 1) / sweep : 0;
        nwg[c] = n > 0 ? (n + wg_max - 1) / wg_max : 0;
        used += nwg[c];
    }
    while (used < target) {
        int best = -1;
        double most = 0.0;
        for (int c = 0; c < n_cont; c++) {
            if (nwg[c] >= cap[c]) continue;
            const double per = (double)n_events[c] / (double)nwg[c];
            if (per > most) { most = per; best = c; }
        }
        if (best < 0) break;
        nwg[best]++;
        used++;
    }
    chunk = sweep;
    blk_start[0] = 0;
    for (int c = 0; c < n_cont; c++) {
        if (nwg[c] > 0) {
            const int64_t per = (n_events[c] + nwg[c] - 1) / nwg[c];
            chunk = std::max(chunk, ((per + sweep - 1) / sweep) * sweep);
        }
        blk_start[c + 1] = blk_start[c] + (int32_t)nwg[c];
    }
    return blk_start[n_cont];
}

static int plan_blocks(const int64_t *n_events, int n_cont, int threads, int64_t &chunk,
                       int32_t *blk_start) {
    const int env = PISA_DEV_INT("HIST_BLOCKS", 0);
    const int64_t target = env > 0 ? env : (int64_t)device_cus() * std::max(1, 1024 / threads);
    return plan_blocks_balanced(n_events, n_cont, threads, target, chunk, blk_start);
}

// optional hipEvent pair recorded around the accumulate kernel of the next
// hist launch (bench.py measures the dominant kernel with them)
static thread_local hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;

static int run_hist(const ContDev *conts, int n_cont, int mode, const DevBinning *grid,
                    int64_t n_nodes, const double *prob_nu, const double *prob_nubar,
                    const double *pepmu, const DevBinning &outb, int64_t n_bins,
                    long long *d_limbs, int32_t *d_status, hipStream_t s, bool clear_first = true,
                    bool second_quantity = true) {
    if (n_bins > (1 << 28)) return PISA_HIP_ERR_INVALID;
    int64_t lds_bytes = lds_acc_bytes(n_bins);
    bool lds = lds_bytes <= LDS_ACC_BYTES_MAX;
    int window = 0;
    if (!lds && (mode == 3 || mode == 5 || mode == 7) && !PISA_DEV_INT("HIST_NO_WINDOW", 0)) {
        // binning too large for LDS: accumulate a window of it (see the kernel)
        // a multiple of 32 bins: the LDS bank pair of an accumulator is then (bin - bin_lo) mod 32
        // whatever the slab and quantity, which the bank-aware event order relies on
        window = (int)(LDS_ACC_BYTES_MAX / lds_acc_bytes(1)) / 32 * 32;
        lds_bytes = lds_acc_bytes(window);
        lds = true;
    }
    // two replicas: level with one for events in the LDS-bank-aware order (no same-address
    // deposits left to spread) and with four for node-sorted events (49.6 / 53 / 49 us)
    // events in arbitrary order (modes 0 and 1: no bank-aware order, neighbouring lanes often in the
    // same bin): up to eight replicas, as LDS allows
    int copies = window ? 1 : PISA_DEV_INT("HIST_COPIES", mode <= 1 ? 8 : 2);
    while (copies > 1 && (copies & (copies - 1))) copies--;
    while (copies > 1 && lds_bytes * copies > LDS_ACC_BYTES_MAX) copies >>= 1;
    if (copies < 1) copies = 1;
    if (clear_first)
        PISA_TRY_HIP(hipMemsetAsync(d_limbs, 0, (size_t)n_cont * n_bins * 2 * NL * 8, s));
    for (int base = 0; base < n_cont; base += MAX_CONT) {
        int nc = n_cont - base < MAX_CONT ? n_cont - base : MAX_CONT;
        HistArgs a;
        a.n_cont = nc;
        a.cont_base = base;
        a.n_bins = n_bins;
        a.n_nodes = n_nodes;
        if (grid) a.grid = *grid; else a.grid = outb;
        a.outb = outb;
        a.prob[0] = prob_nu;
        a.prob[1] = prob_nubar;
        a.pepmu = reinterpret_cast<const double2 *>(pepmu);
        a.opts = (PISA_DEV_INT("HIST_DBG", 0) & ~1) | (second_quantity ? 0 : 1);
        a.wait_flags = nullptr;
        a.wait_epoch = 0;
        a.wait_wgs = 0;
#ifdef PISA_DEV_PROBES
        if (g_hist_wait.flags) {
            a.wait_flags = g_hist_wait.flags;
            a.wait_epoch = g_hist_wait.epoch;
            a.wait_wgs = g_hist_wait.n_wg;
        }
#endif
        int64_t nev[MAX_CONT];
        for (int c = 0; c < nc; c++) {
            a.cont[c] = conts[base + c];
            nev[c] = conts[base + c].n;
            // the partitioned order is used where its partitions are this launch's LDS windows
            if (!(window > 0 && mode == 7 && a.cont[c].part_width == window)) a.cont[c].part_start = nullptr;
        }
        int threads = PISA_DEV_INT("HIST_THREADS", 1024);
        if (threads < 64 || threads > 1024 || (threads & 63)) threads = HIST_THREADS;
        int nblocks = plan_blocks(nev, nc, threads, a.chunk, a.blk_start);
        if (nblocks <= 0) continue;
        for (int c = 0; c < nc; c++) {
            a.cont_chunk[c] = 0;
            const int64_t nwg_c = a.blk_start[c + 1] - a.blk_start[c];
            // partitioned window order: equal shares of whole 256-event blocks per workgroup of the container
            if (a.cont[c].part_start && nwg_c > 0) {
                const int64_t blocks = (nev[c] + 255) / 256;
                a.cont_chunk[c] = (int32_t)((blocks + nwg_c - 1) / nwg_c);
            }
        }
        dim3 grid_dim((unsigned)nblocks), block(threads);
        size_t shmem = lds ? (size_t)lds_bytes * copies : 0;
        a.copies = lds ? copies : 1;
        a.window = window;
        bool parts = false;
        for (int c = 0; c < nc; c++) parts = parts || a.cont[c].part_start != nullptr;
        if (parts) {
            // partitioned order: two LDS windows per workgroup (a chunk may reach into a second partition);
            // beyond the default 64 KiB of dynamic LDS, asked for once per device
            const size_t want = (size_t)lds_bytes * 2 + (PART_MAX + 1) * sizeof(int32_t);   // + the partition table
            static std::atomic<uint64_t> attr_set{0};
            int dev = 0;
            PISA_TRY_HIP(hipGetDevice(&dev));
            const uint64_t bit = 1ull << (dev & 63);
            bool ok = (int64_t)want <= device_lds_bytes() - 256;
            if (ok && !(attr_set.load(std::memory_order_acquire) & bit)) {
                ok = hipFuncSetAttribute(reinterpret_cast<const void *>(&hist_accumulate_kernel<7, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(device_lds_bytes() - 256)) == hipSuccess;
                if (ok) attr_set.fetch_or(bit, std::memory_order_release);
            }
            if (ok) {
                shmem = want;
            } else {      // no room for two windows on this part: the general window path
                for (int c = 0; c < nc; c++) { a.cont[c].part_start = nullptr; a.cont_chunk[c] = 0; }
            }
        }
        unsigned long long *out = reinterpret_cast<unsigned long long *>(d_limbs);
        if (g_prof_start) PISA_TRY_HIP(hipEventRecord(g_prof_start, s));
#define LAUNCH(M, L) hipLaunchKernelGGL((hist_accumulate_kernel<M, L>), grid_dim, block, shmem, s, a, out, d_status)
        if (mode == 7) LAUNCH(7, true);
        else if (mode == 5) { if (lds) LAUNCH(5, true); else LAUNCH(5, false); }
        else if (mode == 3) { if (lds) LAUNCH(3, true); else LAUNCH(3, false); }
        else if (mode == 2) { if (lds) LAUNCH(2, true); else LAUNCH(2, false); }
        else if (mode == 1) {
            // dimensions as template parameters (uniform branches around the loads and 64-bit index
            // arithmetic otherwise: 1 080 instructions per pair of events)
            const int gd = a.grid.ndim, od = a.outb.ndim;
            if (gd < 1 || gd > 2 || od < 1 || od > 3 || n_nodes >= (1LL << 31) / 9) return PISA_HIP_ERR_INVALID;
#define LAUNCH1(G, O) do { if (lds) hipLaunchKernelGGL((hist_accumulate_kernel<1, true, 4 * G + O>), grid_dim, block, shmem, s, a, out, d_status); \
                           else hipLaunchKernelGGL((hist_accumulate_kernel<1, false, 4 * G + O>), grid_dim, block, shmem, s, a, out, d_status); } while (0)
            if (gd == 1) { if (od == 1) LAUNCH1(1, 1); else if (od == 2) LAUNCH1(1, 2); else LAUNCH1(1, 3); }
            else { if (od == 1) LAUNCH1(2, 1); else if (od == 2) LAUNCH1(2, 2); else LAUNCH1(2, 3); }
#undef LAUNCH1
        }
        else {
            const int od = a.outb.ndim;
            if (od < 1 || od > 3) return PISA_HIP_ERR_INVALID;
#define LAUNCH0(O) do { if (lds) hipLaunchKernelGGL((hist_accumulate_kernel<0, true, O>), grid_dim, block, shmem, s, a, out, d_status); \
                        else hipLaunchKernelGGL((hist_accumulate_kernel<0, false, O>), grid_dim, block, shmem, s, a, out, d_status); } while (0)
            if (od == 1) LAUNCH0(1); else if (od == 2) LAUNCH0(2); else LAUNCH0(3);
#undef LAUNCH0
        }
#undef LAUNCH
        PISA_CHECK_LAUNCH("hist_accumulate_kernel");
        if (g_prof_stop) PISA_TRY_HIP(hipEventRecord(g_prof_stop, s));
    }
    return PISA_HIP_OK;
}

}  // namespace pisa

using namespace pisa;

PISA_API int pisa_hip_hist_workgroups(const int64_t *h_n_events, int32_t n_containers, int32_t *h_workgroups) {
    if (!h_n_events || !h_workgroups || n_containers < 1) return PISA_HIP_ERR_INVALID;
    for (int base = 0; base < n_containers; base += MAX_CONT) {
        const int nc = n_containers - base < MAX_CONT ? n_containers - base : MAX_CONT;
        int64_t chunk = 0;
        int32_t blk_start[MAX_CONT + 1];
        plan_blocks(h_n_events + base, nc, HIST_THREADS, chunk, blk_start);
        for (int c = 0; c < nc; c++) h_workgroups[base + c] = blk_start[c + 1] - blk_start[c];
    }
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_hist_window_bins(int64_t n_bins) {
    if (n_bins < 1) return -1;
    if (lds_acc_bytes(n_bins) <= LDS_ACC_BYTES_MAX) return 0;
    return (int)(LDS_ACC_BYTES_MAX / lds_acc_bytes(1)) / 32 * 32;
}

PISA_API int pisa_hip_profile_events(void *start_event, void *stop_event) {
    g_prof_start = reinterpret_cast<hipEvent_t>(start_event);
    g_prof_stop = reinterpret_cast<hipEvent_t>(stop_event);
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_event_indices(const pisa_hip_binning *h_binning,
                                    const double *const *h_d_sample, int64_t n,
                                    int32_t *d_index, void *stream) {
    DevBinning b;
    int64_t n_bins;
    int rc = make_dev_binning(h_binning, b, n_bins);
    if (rc) return rc;
    if (n < 0 || !h_d_sample || n_bins > 0x7fffffffLL) return PISA_HIP_ERR_INVALID;
    if (n == 0) return PISA_HIP_OK;
    if (!d_index) return PISA_HIP_ERR_INVALID;
    for (int k = 0; k < b.ndim; k++)
        if (!h_d_sample[k]) return PISA_HIP_ERR_INVALID;
    dim3 block(256), grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(event_indices_kernel, grid, block, 0, as_stream(stream), b, h_d_sample[0],
                       b.ndim > 1 ? h_d_sample[1] : nullptr, b.ndim > 2 ? h_d_sample[2] : nullptr,
                       n, d_index);
    PISA_CHECK_LAUNCH("event_indices_kernel");
    return PISA_HIP_OK;
}

static int reweight_hist_impl(const pisa_hip_container *h_containers, int32_t n_containers,
                              const pisa_hip_binning *h_calc_grid, const double *d_prob_nu,
                              const double *d_prob_nubar, const double *d_pepmu,
                              const pisa_hip_binning *h_out_binning, int64_t *d_limbs,
                              int32_t *d_status, void *stream, bool clear_first) {
    if (!h_containers || n_containers < 1 || n_containers > 1024 || !d_limbs)
        return PISA_HIP_ERR_INVALID;
    DevBinning grid, outb;
    int64_t n_nodes, n_bins;
    int rc = make_dev_binning(h_calc_grid, grid, n_nodes);
    if (rc) return rc;
    if (grid.ndim > 2) return PISA_HIP_ERR_INVALID;
    if ((rc = make_dev_binning(h_out_binning, outb, n_bins))) return rc;
    ContDev *conts = new ContDev[n_containers];
    bool any_table = d_pepmu != nullptr;
    for (int c = 0; c < n_containers; c++) any_table = any_table || h_containers[c].d_pepmu;
    bool all_indexed = d_pepmu != nullptr;
    bool all_packed = any_table;
    bool all_compact = any_table;
    // the 16-bit index form needs 16-bit node / bin numbers; where it does not apply its columns
    // are ignored (the other forms, if given, are used)
    const bool ok16 = n_nodes < 0xffff && n_bins < 0xffff &&
                      (lds_acc_bytes(n_bins) <= LDS_ACC_BYTES_MAX || !PISA_DEV_INT("HIST_NO_WINDOW", 0));
    bool all_idx16 = any_table && ok16;
    for (int c = 0; c < n_containers; c++) {
        const pisa_hip_container &h = h_containers[c];
        ContDev &d = conts[c];
        const bool compact = h.d_node_bin && h.d_weighted_flux;
        const bool c16 = ok16 && h.d_node_bin16 && h.d_weighted_flux_q;  // stands alone
        bool packed = h.d_node_bin && (h.d_aeff_w0 || h.d_weighted_flux);
        bool indexed = packed || (h.d_node && h.d_bin);
        bool bad = h.n_events < 0 || h.flav < 0 || h.flav > 2 || (h.nubar != 1 && h.nubar != -1);
        if (h.n_events > 0) {
            const bool has_tab = d_pepmu || h.d_pepmu;
            bad = bad || (!h.d_nu_flux && !((compact || c16) && has_tab));
            if (!((packed || c16) && has_tab)) bad = bad || !h.d_weighted_aeff || !h.d_initial_weights;
            if (!((indexed || c16) && has_tab)) {
                bad = bad || !h.d_grid_x || (grid.ndim > 1 && !h.d_grid_y);
                for (int k = 0; k < outb.ndim; k++) bad = bad || !h.d_sample[k];
                bad = bad || (h.nubar > 0 ? !d_prob_nu : !d_prob_nubar);
            }
            all_indexed = all_indexed && indexed && (h.d_node && h.d_bin);
            all_packed = all_packed && packed && has_tab && h.d_aeff_w0 && h.d_nu_flux;
            all_compact = all_compact && compact && has_tab;
            all_idx16 = all_idx16 && c16 && has_tab;
        }
        if (bad) { delete[] conts; return PISA_HIP_ERR_INVALID; }
        d.n = h.n_events;
        d.gx = h.d_grid_x; d.gy = h.d_grid_y; d.flux = h.d_nu_flux; d.aeff = h.d_weighted_aeff;
        d.w0 = h.d_initial_weights;
        for (int k = 0; k < 3; k++) d.s[k] = h.d_sample[k];
        d.node = h.d_node; d.bin = h.d_bin;
        d.node_bin = reinterpret_cast<const int2 *>(h.d_node_bin);
        d.aeff_w0 = reinterpret_cast<const double2 *>(h.d_aeff_w0);
        d.pepmu_own = reinterpret_cast<const double2 *>(h.d_pepmu);
        d.wflux = reinterpret_cast<const double2 *>(h.d_weighted_flux);
        d.idx16 = h.d_node_bin16;
        d.wflux_q = reinterpret_cast<const double2 *>(h.d_weighted_flux_q);
        d.part_start = (h.d_part_start && h.n_part >= 1 && h.n_part <= PART_MAX && h.part_width > 0) ? h.d_part_start : nullptr;
        d.n_part = h.n_part;
        d.part_width = h.part_width;
        d.scale = h.scale;
        d.flav = h.flav;
        d.side = h.nubar > 0 ? 0 : 1;
    }
    rc = run_hist(conts, n_containers, all_idx16 ? 7 : all_compact ? 5 : (all_packed ? 3 : (all_indexed ? 2 : 1)), &grid, n_nodes, d_prob_nu, d_prob_nubar,
                  d_pepmu, outb, n_bins, (long long *)d_limbs, d_status, as_stream(stream), clear_first);
    delete[] conts;
    return rc;
}

PISA_API int pisa_hip_reweight_hist(const pisa_hip_container *h_containers, int32_t n_containers,
                                    const pisa_hip_binning *h_calc_grid, const double *d_prob_nu,
                                    const double *d_prob_nubar, const double *d_pepmu,
                                    const pisa_hip_binning *h_out_binning, int64_t *d_limbs,
                                    int32_t *d_status, void *stream) {
    return reweight_hist_impl(h_containers, n_containers, h_calc_grid, d_prob_nu, d_prob_nubar,
                              d_pepmu, h_out_binning, d_limbs, d_status, stream, true);
}

PISA_API int pisa_hip_reweight_hist_acc(const pisa_hip_container *h_containers,
                                        int32_t n_containers, const pisa_hip_binning *h_calc_grid,
                                        const double *d_prob_nu, const double *d_prob_nubar,
                                        const double *d_pepmu,
                                        const pisa_hip_binning *h_out_binning, int64_t *d_limbs,
                                        int32_t *d_status, void *stream) {
    return reweight_hist_impl(h_containers, n_containers, h_calc_grid, d_prob_nu, d_prob_nubar,
                              d_pepmu, h_out_binning, d_limbs, d_status, stream, false);
}


template <int KP>
static int launch_multi(const MultiArgs &a, int nblocks, size_t shmem, unsigned long long *out,
                        int32_t *d_status, hipStream_t s) {
    // LDS beyond the default 64 KiB has to be asked for once per kernel AND device
    static std::atomic<uint64_t> attr_set{0};
    int dev = 0;
    PISA_TRY_HIP(hipGetDevice(&dev));
    const uint64_t bit = 1ull << (dev & 63);
    if (shmem > 64 * 1024 && !(attr_set.load(std::memory_order_acquire) & bit)) {
        const int64_t limit = device_lds_bytes() - 256;
        if ((int64_t)shmem > limit) return PISA_HIP_ERR_INVALID;
        PISA_TRY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&hist_accumulate_multi_kernel<KP>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)limit));
        attr_set.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((hist_accumulate_multi_kernel<KP>), dim3((unsigned)nblocks), dim3(HIST_THREADS), shmem, s,
                       a, out, d_status);
    PISA_CHECK_LAUNCH("hist_accumulate_multi_kernel");
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_multi_points_per_pass(int64_t n_bins) {
    if (n_bins < 1) return 0;
    int64_t lds_max = (int64_t)PISA_DEV_INT("MULTI_LDS_KB", 128) * 1024;
    if (lds_max > device_lds_bytes() - 256) lds_max = device_lds_bytes() - 256;   // a part with 64 KiB LDS: fewer points per pass
    int64_t kp = lds_max / lds_acc_bytes(n_bins);
    if (kp > MULTI_KP_MAX) kp = MULTI_KP_MAX;
    return kp < 2 ? 0 : (int)kp;   // one point per pass is the single-point path's job
}

PISA_API int pisa_hip_reweight_hist_multi(const pisa_hip_container *h_containers, int32_t n_containers,
                                          const pisa_hip_binning *h_calc_grid, const double *d_pepmu_points,
                                          int32_t n_points, const double *h_scales,
                                          const pisa_hip_binning *h_out_binning, int64_t *d_limbs,
                                          int32_t clear_first, int32_t *d_status, void *stream) {
    if (!h_containers || n_containers < 1 || n_containers > 1024 || !d_limbs || !d_pepmu_points ||
        n_points < 1 || n_points > PISA_HIP_MAX_POINTS)
        return PISA_HIP_ERR_INVALID;
    DevBinning grid, outb;
    int64_t n_nodes, n_bins;
    int rc = make_dev_binning(h_calc_grid, grid, n_nodes);
    if (rc) return rc;
    if ((rc = make_dev_binning(h_out_binning, outb, n_bins))) return rc;
    if (n_nodes >= 0xffff || n_bins >= 0xffff) return PISA_HIP_ERR_INVALID;
    const int kp_max = pisa_hip_multi_points_per_pass(n_bins);
    if (kp_max < 1) return PISA_HIP_ERR_INVALID;   // binning beyond the LDS accumulators: point by point
    for (int c = 0; c < n_containers; c++) {
        const pisa_hip_container &h = h_containers[c];
        if (h.n_events < 0 || h.flav < 0 || h.flav > 2 || (h.nubar != 1 && h.nubar != -1)) return PISA_HIP_ERR_INVALID;
        // the 16-bit index form with the shared grid tables only
        if (h.n_events > 0 && (!h.d_node_bin16 || !h.d_weighted_flux_q || h.d_pepmu)) return PISA_HIP_ERR_INVALID;
    }
    hipStream_t s = as_stream(stream);
    const int64_t limb_stride = (int64_t)n_containers * n_bins * 2 * NL;
    if (clear_first) PISA_TRY_HIP(hipMemsetAsync(d_limbs, 0, (size_t)n_points * limb_stride * 8, s));
    const int n_pass = (n_points + kp_max - 1) / kp_max;
    for (int pass = 0, k0 = 0; pass < n_pass; pass++) {
        const int kp = (n_points - k0 + (n_pass - pass) - 1) / (n_pass - pass);   // passes of equal size
        // workgroups: what is resident at once.  From two points on the kernel needs more than 64 VGPRs,
        // a CU holds ONE 1024-thread workgroup, and a second round of workgroups would only add a tail
        const int64_t target_blocks = PISA_DEV_INT("MULTI_BLOCKS", device_cus());
        for (int base = 0; base < n_containers; base += MAX_CONT) {
            const int nc = n_containers - base < MAX_CONT ? n_containers - base : MAX_CONT;
            MultiArgs a;
            memset(&a, 0, sizeof(a));
            a.n_cont = nc;
            a.cont_base = base;
            a.n_bins = (int32_t)n_bins;
            a.k_stride = n_points;
            a.n_nodes = n_nodes;
            a.limb_stride = limb_stride;
            a.pepmu = reinterpret_cast<const double2 *>(d_pepmu_points) + k0;
            int64_t nev[MAX_CONT];
            for (int c = 0; c < nc; c++) {
                const pisa_hip_container &h = h_containers[base + c];
                a.cont[c].n = h.n_events;
                a.cont[c].idx16 = h.d_node_bin16;
                a.cont[c].wflux_q = reinterpret_cast<const double2 *>(h.d_weighted_flux_q);
                a.cont[c].flav = h.flav;
                a.cont[c].side = h.nubar > 0 ? 0 : 1;
                nev[c] = h.n_events;
                for (int k = 0; k < kp; k++)
                    a.scale[k][c] = h_scales ? h_scales[(size_t)(k0 + k) * n_containers + base + c] : h.scale;
            }
            int64_t chunk;
            const int nblocks = plan_blocks_balanced(nev, nc, HIST_THREADS, target_blocks, chunk, a.blk_start);
            if (nblocks <= 0) continue;
            const size_t shmem = (size_t)lds_acc_bytes(n_bins) * kp;
            unsigned long long *out = reinterpret_cast<unsigned long long *>(d_limbs) + (int64_t)k0 * limb_stride;
            if (g_prof_start) PISA_TRY_HIP(hipEventRecord(g_prof_start, s));
            switch (kp) {
            case 1: rc = launch_multi<1>(a, nblocks, shmem, out, d_status, s); break;
            case 2: rc = launch_multi<2>(a, nblocks, shmem, out, d_status, s); break;
            case 3: rc = launch_multi<3>(a, nblocks, shmem, out, d_status, s); break;
            case 4: rc = launch_multi<4>(a, nblocks, shmem, out, d_status, s); break;
            case 5: rc = launch_multi<5>(a, nblocks, shmem, out, d_status, s); break;
            case 6: rc = launch_multi<6>(a, nblocks, shmem, out, d_status, s); break;
            case 7: rc = launch_multi<7>(a, nblocks, shmem, out, d_status, s); break;
            default: rc = launch_multi<8>(a, nblocks, shmem, out, d_status, s); break;
            }
            if (rc) return rc;
            if (g_prof_stop) PISA_TRY_HIP(hipEventRecord(g_prof_stop, s));
        }
        k0 += kp;
    }
    return PISA_HIP_OK;
}

static int finalize_metric_impl(int64_t *d_limbs, int32_t n_points, int32_t n_containers,
                                int64_t n_bins, double *d_hist, double *d_sumw2, int32_t kind,
                                const double *d_actual, const double *d_scale,
                                int64_t scale_point_stride, const double *d_extra, double *total,
                                int32_t *d_status, int32_t *d_metric_status,
                                int32_t clear_limbs, void *stream, int n_parts) {
    if (!d_limbs || !d_hist || !d_sumw2 || !d_actual || !total || n_containers < 1 || n_bins < 1 ||
        n_points < 1 || n_points > PISA_HIP_MAX_POINTS)
        return PISA_HIP_ERR_INVALID;
    if (kind < PISA_HIP_METRIC_LLH || kind > PISA_HIP_METRIC_MOD_CHI2) return PISA_HIP_ERR_INVALID;
    if (n_parts != 1 && n_parts != 4 && n_parts != 16) return PISA_HIP_ERR_INVALID;
    const bool split = n_parts > 1;
    if (split && kind == PISA_HIP_METRIC_CHI2) return PISA_HIP_ERR_INVALID;
    if ((int64_t)n_containers * n_bins > PISA_HIP_FINALIZE_METRIC_MAX) return PISA_HIP_ERR_INVALID;
    const int n_tot = (int)(n_containers * n_bins);
    // a thread per accumulator of the workgroup's share where 1 024 threads allow it
    int threads = (((split ? (2 * n_tot + n_parts - 1) / n_parts : n_tot) + 63) / 64) * 64;
    if (threads < 256) threads = 256;  // the metric reduction tree is 256 wide
    if (threads > 1024) threads = 1024;
    const int64_t limb_stride = (int64_t)n_tot * 2 * NL;
    auto launch = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3((unsigned)(n_points * n_parts)), dim3(threads), (size_t)n_tot * 16,
                           as_stream(stream), (long long *)d_limbs, (int)n_containers, (int)n_bins, d_hist, d_sumw2,
                           d_actual, total, d_status, d_metric_status, (int)clear_limbs, d_scale, d_extra,
                           limb_stride, (int64_t)scale_point_stride);
    };
    if (n_parts == 16) {
        switch (kind) {
        case PISA_HIP_METRIC_LLH: launch(finalize_metric_kernel<PISA_HIP_METRIC_LLH, 16>); break;
        case PISA_HIP_METRIC_POISSON_LLH: launch(finalize_metric_kernel<PISA_HIP_METRIC_POISSON_LLH, 16>); break;
        default: launch(finalize_metric_kernel<PISA_HIP_METRIC_MOD_CHI2, 16>);
        }
    } else if (split) {
        switch (kind) {
        case PISA_HIP_METRIC_LLH: launch(finalize_metric_kernel<PISA_HIP_METRIC_LLH, 4>); break;
        case PISA_HIP_METRIC_POISSON_LLH: launch(finalize_metric_kernel<PISA_HIP_METRIC_POISSON_LLH, 4>); break;
        default: launch(finalize_metric_kernel<PISA_HIP_METRIC_MOD_CHI2, 4>);
        }
    } else {
        switch (kind) {
        case PISA_HIP_METRIC_LLH: launch(finalize_metric_kernel<PISA_HIP_METRIC_LLH>); break;
        case PISA_HIP_METRIC_POISSON_LLH: launch(finalize_metric_kernel<PISA_HIP_METRIC_POISSON_LLH>); break;
        case PISA_HIP_METRIC_CHI2: launch(finalize_metric_kernel<PISA_HIP_METRIC_CHI2>); break;
        default: launch(finalize_metric_kernel<PISA_HIP_METRIC_MOD_CHI2>);
        }
    }
    PISA_CHECK_LAUNCH("finalize_metric_kernel");
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_finalize_metric_multi(int64_t *d_limbs, int32_t n_points, int32_t n_containers,
                                            int64_t n_bins, double *d_hist, double *d_sumw2, int32_t kind,
                                            const double *d_actual, const double *d_scale,
                                            int64_t scale_point_stride, const double *d_extra, double *total,
                                            int32_t *d_status, int32_t *d_metric_status,
                                            int32_t clear_limbs, void *stream) {
    return finalize_metric_impl(d_limbs, n_points, n_containers, n_bins, d_hist, d_sumw2, kind, d_actual, d_scale,
                                scale_point_stride, d_extra, total, d_status, d_metric_status, clear_limbs, stream, 1);
}

PISA_API int pisa_hip_finalize_metric_split(int64_t *d_limbs, int32_t n_points, int32_t n_containers,
                                            int64_t n_bins, double *d_hist, double *d_sumw2, int32_t kind,
                                            const double *d_actual, const double *d_scale,
                                            int64_t scale_point_stride, const double *d_extra, double *partial,
                                            int32_t *d_status, int32_t *d_metric_status,
                                            int32_t clear_limbs, void *stream) {
    return finalize_metric_impl(d_limbs, n_points, n_containers, n_bins, d_hist, d_sumw2, kind, d_actual, d_scale,
                                scale_point_stride, d_extra, partial, d_status, d_metric_status, clear_limbs, stream, 4);
}

PISA_API int pisa_hip_finalize_metric_parts(int64_t *d_limbs, int32_t n_points, int32_t n_containers,
                                            int64_t n_bins, double *d_hist, double *d_sumw2, int32_t kind,
                                            const double *d_actual, const double *d_scale,
                                            int64_t scale_point_stride, const double *d_extra, double *partial,
                                            int32_t n_parts, int32_t *d_status, int32_t *d_metric_status,
                                            int32_t clear_limbs, void *stream) {
    if (n_parts != 4 && n_parts != 16) return PISA_HIP_ERR_INVALID;
    return finalize_metric_impl(d_limbs, n_points, n_containers, n_bins, d_hist, d_sumw2, kind, d_actual, d_scale,
                                scale_point_stride, d_extra, partial, d_status, d_metric_status, clear_limbs, stream,
                                n_parts);
}

PISA_API int pisa_hip_finalize_metric_scaled(int64_t *d_limbs, int32_t n_containers, int64_t n_bins,
                                             double *d_hist, double *d_sumw2, int32_t kind,
                                             const double *d_actual, const double *d_scale,
                                             const double *d_extra, double *total, int32_t *d_status,
                                             int32_t *d_metric_status, int32_t clear_limbs,
                                             void *stream) {
    return pisa_hip_finalize_metric_multi(d_limbs, 1, n_containers, n_bins, d_hist, d_sumw2, kind, d_actual,
                                          d_scale, 0, d_extra, total, d_status, d_metric_status, clear_limbs,
                                          stream);
}

PISA_API int pisa_hip_finalize_metric(int64_t *d_limbs, int32_t n_containers, int64_t n_bins,
                                      double *d_hist, double *d_sumw2, int32_t kind,
                                      const double *d_actual, double *total, int32_t *d_status,
                                      int32_t *d_metric_status, int32_t clear_limbs,
                                      void *stream) {
    return pisa_hip_finalize_metric_scaled(d_limbs, n_containers, n_bins, d_hist, d_sumw2, kind, d_actual,
                                           nullptr, nullptr, total, d_status, d_metric_status, clear_limbs,
                                           stream);
}

PISA_API int pisa_hip_hist_finalize(const int64_t *d_limbs, int32_t n_containers, int64_t n_bins,
                                    double *d_hist, double *d_sumw2, int32_t *d_status,
                                    void *stream) {
    if (!d_limbs || n_containers < 1 || n_bins < 1) return PISA_HIP_ERR_INVALID;
    int64_t total = (int64_t)n_containers * n_bins;
    dim3 block(256), grid((unsigned)((total + 255) / 256));
    hipLaunchKernelGGL(hist_finalize_kernel, grid, block, 0, as_stream(stream),
                       (const long long *)d_limbs, total, d_hist, d_sumw2, d_status);
    PISA_CHECK_LAUNCH("hist_finalize_kernel");
    return PISA_HIP_OK;
}

// Generic histogram.  Its scratch (limbs, counts, status word) lives in a grow-only buffer per host
// thread: three hipMalloc / hipFree pairs per call cost more than the kernels (72 us for a call that
// moves 80 MB).  The call still ends with a stream synchronisation -- it returns the status.
static thread_local char *g_hist_scratch = nullptr;
static thread_local size_t g_hist_scratch_bytes = 0;

PISA_API int pisa_hip_histogram_regular(const pisa_hip_binning *h_binning,
                                        const double *const *h_d_sample, int64_t n,
                                        const double *d_weights, int32_t averaged, double *d_hist,
                                        void *stream) {
    DevBinning outb;
    int64_t n_bins;
    int rc = make_dev_binning(h_binning, outb, n_bins);
    if (rc) return rc;
    if (n < 0 || !d_hist || !h_d_sample) return PISA_HIP_ERR_INVALID;
    for (int k = 0; k < outb.ndim; k++)
        if (n > 0 && !h_d_sample[k]) return PISA_HIP_ERR_INVALID;
    hipStream_t s = as_stream(stream);
    const size_t limb_bytes = (size_t)n_bins * 2 * NL * 8, cnt_bytes = (size_t)n_bins * 8;
    const size_t need = limb_bytes + cnt_bytes + 256;
    if (need > g_hist_scratch_bytes) {
        if (g_hist_scratch) (void)hipFree(g_hist_scratch);
        g_hist_scratch = nullptr;
        g_hist_scratch_bytes = 0;
        rc = check_hip(hipMalloc(&g_hist_scratch, need), "hipMalloc");
        if (rc) return rc;
        g_hist_scratch_bytes = need;
    }
    long long *limbs = reinterpret_cast<long long *>(g_hist_scratch);
    double *cnt = reinterpret_cast<double *>(g_hist_scratch + limb_bytes);
    int32_t *st = reinterpret_cast<int32_t *>(g_hist_scratch + limb_bytes + cnt_bytes);
    rc = check_hip(hipMemsetAsync(st, 0, 4, s), "memset");
    if (!rc) {
        ContDev c;
        c.n = n; c.gx = c.gy = c.flux = c.aeff = nullptr; c.w0 = d_weights;
        for (int k = 0; k < 3; k++) c.s[k] = k < outb.ndim ? h_d_sample[k] : nullptr;
        c.node = c.bin = nullptr;
        c.node_bin = nullptr; c.aeff_w0 = nullptr; c.pepmu_own = nullptr; c.wflux = nullptr;
        c.idx16 = nullptr; c.wflux_q = nullptr;
        c.scale = 1.0; c.flav = 0; c.side = 0;
        // the per-bin counts (second quantity) are needed for the average only
        rc = run_hist(&c, 1, 0, nullptr, 0, nullptr, nullptr, nullptr, outb, n_bins, limbs, st, s, true,
                      averaged != 0);
    }
    if (!rc) rc = pisa_hip_hist_finalize((const int64_t *)limbs, 1, n_bins, d_hist, cnt, st, s);
    if (!rc && averaged) {
        dim3 block(256), grid((unsigned)((n_bins + 255) / 256));
        hipLaunchKernelGGL(hist_average_kernel, grid, block, 0, s, n_bins, d_hist, cnt);
        rc = check_hip(hipGetLastError(), "hist_average_kernel");
    }
    int32_t h_st = 0;
    if (!rc) rc = check_hip(hipMemcpyAsync(&h_st, st, 4, hipMemcpyDeviceToHost, s), "d2h");
    if (!rc) rc = check_hip(hipStreamSynchronize(s), "sync");
    if (!rc && h_st) rc = PISA_HIP_ERR_OVERFLOW;
    return rc;
}

PISA_API int pisa_hip_lookup_regular(const pisa_hip_binning *h_binning,
                                     const double *const *h_d_sample, int64_t n,
                                     const double *d_flat_hist, int32_t width, double *d_out,
                                     void *stream) {
    DevBinning b;
    int64_t n_bins;
    int rc = make_dev_binning(h_binning, b, n_bins);
    if (rc) return rc;
    if (n < 0 || width < 1 || !h_d_sample) return PISA_HIP_ERR_INVALID;
    if (n == 0) return PISA_HIP_OK;
    if (!d_flat_hist || !d_out) return PISA_HIP_ERR_INVALID;
    for (int k = 0; k < b.ndim; k++)
        if (!h_d_sample[k]) return PISA_HIP_ERR_INVALID;
    dim3 block(256), grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(lookup_regular_kernel, grid, block, 0, as_stream(stream), b, h_d_sample[0],
                       b.ndim > 1 ? h_d_sample[1] : nullptr, b.ndim > 2 ? h_d_sample[2] : nullptr,
                       n, d_flat_hist, (int)width, d_out);
    PISA_CHECK_LAUNCH("lookup_regular_kernel");
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_apply_osc_weights(const double *d_nu_flux, const double *d_prob_e,
                                        const double *d_prob_mu, int64_t n, double *d_weights,
                                        void *stream) {
    if (n < 0) return PISA_HIP_ERR_INVALID;
    if (n == 0) return PISA_HIP_OK;
    if (!d_nu_flux || !d_prob_e || !d_prob_mu || !d_weights) return PISA_HIP_ERR_INVALID;
    dim3 block(256), grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(apply_osc_weights_kernel, grid, block, 0, as_stream(stream), d_nu_flux,
                       d_prob_e, d_prob_mu, n, d_weights);
    PISA_CHECK_LAUNCH("apply_osc_weights_kernel");
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_apply_osc_weights_strided(const double *d_nu_flux, const double *d_prob_e,
                                                const double *d_prob_mu, int64_t prob_stride, int64_t n,
                                                double *d_weights, void *stream) {
    if (n < 0 || prob_stride < 1) return PISA_HIP_ERR_INVALID;
    if (n == 0) return PISA_HIP_OK;
    if (!d_nu_flux || !d_prob_e || !d_prob_mu || !d_weights) return PISA_HIP_ERR_INVALID;
    dim3 block(256), grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(apply_osc_weights_strided_kernel, grid, block, 0, as_stream(stream), d_nu_flux,
                       d_prob_e, d_prob_mu, prob_stride, n, d_weights);
    PISA_CHECK_LAUNCH("apply_osc_weights_strided_kernel");
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_weight_chain_multi(const pisa_hip_chain_set *h_sets, int32_t n_sets, void *stream) {
    if (!h_sets || n_sets < 1 || n_sets > 4096) return PISA_HIP_ERR_INVALID;
    for (int k = 0; k < n_sets; k++) {
        const pisa_hip_chain_set &h = h_sets[k];
        if (h.n < 0 || (h.n > 0 && (!h.d_initial_weights || !h.d_weights)) ||
            (h.n > 0 && h.d_nu_flux && (!h.d_prob_e || !h.d_prob_mu || h.prob_stride < 1)))
            return PISA_HIP_ERR_INVALID;
    }
    for (int base = 0; base < n_sets; base += CHAIN_MAX_SETS) {
        const int nc = n_sets - base < CHAIN_MAX_SETS ? n_sets - base : CHAIN_MAX_SETS;
        ChainArgs a;
        int64_t n_max = 0;
        for (int k = 0; k < nc; k++) { a.set[k] = h_sets[base + k]; n_max = std::max(n_max, h_sets[base + k].n); }
        if (n_max == 0) continue;
        dim3 block(256), grid((unsigned)((n_max + 255) / 256), (unsigned)nc);
        hipLaunchKernelGGL(weight_chain_multi_kernel, grid, block, 0, as_stream(stream), a);
        PISA_CHECK_LAUNCH("weight_chain_multi_kernel");
    }
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_apply_aeff(const double *d_weighted_aeff, double scale, int64_t n,
                                 double *d_weights, void *stream) {
    if (n < 0) return PISA_HIP_ERR_INVALID;
    if (n == 0) return PISA_HIP_OK;
    if (!d_weighted_aeff || !d_weights) return PISA_HIP_ERR_INVALID;
    dim3 block(256), grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(apply_aeff_kernel, grid, block, 0, as_stream(stream), d_weighted_aeff,
                       scale, n, d_weights);
    PISA_CHECK_LAUNCH("apply_aeff_kernel");
    return PISA_HIP_OK;
}
