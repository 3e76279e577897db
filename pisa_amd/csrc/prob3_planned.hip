// prob3_planned.hip -- the planned (E x coszen) grid evaluation of prob3: fused terms+amplitude
// kernel, two-sided chain kernel, and the host-side plan (pisa_hip_grid_plan_*,
// pisa_hip_prob3_grid_planned).
//
// This translation unit alone is compiled with -ffp-contract=fast (see the Makefile): the
// planned form is equal to the reference's operation order only to rounding anyway (products
// associated in parts, reciprocal quotients), and contraction removes a quarter of its
// instructions, which on these latency-bound kernels is time.  Everything else in the
// library (array / one-kernel grid / event kernels, fused reweight, metric ...) keeps
// -ffp-contract=off.
#include <stdlib.h>
#include <string.h>

#include "common.hpp"
#include "prob3_device.hpp"

namespace pisa {

// ------------------------------------------------------- planned grid form
// The work of a grid evaluation is tiny (~0.1 GFLOP); what costs is latency: the
// eigenvalue/projector terms of one (E, density) are a ~15 us dependent chain and a
// row's ordered matrix product is up to 24 dependent 3x3 complex products.  The
// plan (host, once per Earth model / coszen grid) therefore
//   * resolves the reference's layer-matrix cache (numba_osc_kernels.py:236-249)
//     and gives mirrored layers of a row ONE matrix ("pair" = (density, length)),
//   * cuts the pairs of each distinct density into work items of a few pairs,
//   * lists every row's chain as pair indices in path order,
// and an evaluation is two launches:
//   stage A   wave = (distinct density, sign, 64 energies): the terms of (E, rho)
//             -> records terms[sign][density][field][n_e]  (1-3 MB, L2 resident)
//   stage C   workgroup = (row, sign, 64 energies) x G waves: forms each layer matrix
//             A = sum_k phase_k Q_k from the record of the layer's density where it multiplies
//             it (a pair belongs to one row: nothing is computed twice) -- the chain is
//             multiplied from its middle outwards on both sides at once (see
//             prob3_chain_kernel), wave 0 joins the waves' partial products (LDS), rotates to
//             the flavour basis and stores P and the gather tables.
// The earlier split is kept as an option (PISA_HIP_PROB3_FUSED_AMP=0 at plan creation):
//   stage AB  wave = (item of a few pairs of one density, sign, 64 energies): terms in registers,
//             then the item's layer matrices -> amp[sign][pair][18][n_e]; stage C reads them.
//             Stage AB is bound by those stores (20-29 MB per evaluation through HBM).
// Without decay the layer matrices are taken in their SU(3) form (see eigen_terms).
// Stage C associates the product differently from the sequential reference and uses
// fused multiply-adds, so its results agree with prob3_grid_kernel to rounding
// (<= 3e-13 absolute on the probabilities), not bit for bit.
constexpr int CHAIN_GROUPS_DEFAULT = 2;

// Where a kernel finds the per-evaluation constants.  One parameter point: by value in the
// kernel-argument segment (2.3 KB, scalar loads).  SEVERAL independent parameter points in one launch
// (pisa_hip_prob3_grid_planned_multi: blockIdx.y = 2 * point + sign): an array in device memory, the
// point's block read through the constant address space -- the same scalar loads, from another
// address -- so both forms run the same instructions on the same numbers.
struct ConstsByValue {
    Prob3Consts c;
    __device__ __forceinline__ const Prob3Consts &at(int) const { return c; }
};
struct ConstsByPointer {
    const Prob3Consts *__restrict__ p;
    __device__ __forceinline__ const Prob3Consts &at(int k) const { return p[k]; }
};

// host-mapped staging block -> device array of the points' constants (one small launch per batch)
__global__ void __launch_bounds__(256)
prob3_stage_consts_kernel(const unsigned long long *__restrict__ src, unsigned long long *__restrict__ dst, int n_words) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

template <bool DECAY>
__global__ void __launch_bounds__(64)
prob3_terms_amp_kernel(const Prob3Consts c, const double *__restrict__ energy, int n_e,
                       const double *__restrict__ rho_unique, const int32_t *__restrict__ item_u,
                       const int32_t *__restrict__ item_p0, const int32_t *__restrict__ item_cnt,
                       const double *__restrict__ pair_dist, int n_pairs,
                       double *__restrict__ amp) {
    const int item = blockIdx.x;
    const int side = blockIdx.y;
    const int ie = blockIdx.z * 64 + threadIdx.x;
    if (ie >= n_e) return;
    const double e = energy[ie];
    double rec[PROB3_NF];
    auto store = [&](int f, double v) { rec[f] = v; };
    eigen_terms<DECAY>(c.side[side], c.dm, c.vac_order, e, rho_unique[item_u[item]], store);
    auto load = [&](int f) { return rec[f]; };
    const int p0 = item_p0[item], cnt = item_cnt[item];
    for (int q = 0; q < cnt; q++) {
        mat3 A;
        amplitude_from_terms<DECAY>(load, pair_dist[p0 + q] / e, A);
        const int64_t ns = (int64_t)gridDim.z * 64;  // energy stride: whole tiles, cache-line aligned
        double *o = amp + ((int64_t)(side * n_pairs + p0 + q) * 18) * ns + ie;
        // no decay: rows 0 and 1 of the SU(3) form only (12 of the 18 slots of a pair); the
        // chain kernel completes the third row.  This kernel is bound by these stores.
#pragma unroll
        for (int i = 0; i < (DECAY ? 3 : 2); i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                o[(int64_t)(6 * i + 2 * j) * ns] = A.m[i][j].re;
                o[(int64_t)(6 * i + 2 * j + 1) * ns] = A.m[i][j].im;
            }
    }
}

// Stage A alone: the terms of every (distinct density, sign, energy), field-major so that the
// lanes of a wave (64 energies) read and write contiguously.  Used with the chain kernel's
// AMP mode, which forms each layer matrix from these records where it multiplies it.
template <bool DECAY, class CS>
__global__ void __launch_bounds__(64)
prob3_terms_kernel(const CS cs, const double *__restrict__ energy, int n_e,
                   const double *__restrict__ rho_unique, int n_unique,
                   double *__restrict__ terms) {
    const int u = blockIdx.x;
    const int side = blockIdx.y & 1;
    const int pt = blockIdx.y >> 1;     // parameter point (0 in the one-point form)
    const Prob3Consts &c = cs.at(pt);
    const int ie = blockIdx.z * 64 + threadIdx.x;
    if (ie >= n_e) return;
    const int64_t ns = (int64_t)gridDim.z * 64;
    // fields in pairs, [field / 2][E][2]: the chain kernel reads a record with 16-byte loads
    double *o = terms + ((int64_t)((pt * 2 + side) * n_unique + u) * PROB3_NF) * ns + 2 * (int64_t)ie;
    auto store = [&](int f, double v) { o[(int64_t)(f >> 1) * (2 * ns) + (f & 1)] = v; };
    eigen_terms<DECAY>(c.side[side], c.dm, c.vac_order, energy[ie], rho_unique[u], store);
}

// C = A.B with fused multiply-adds (4 per complex multiply-accumulate instead of 4 mul + 4
// add).  Stage C is a short dependent sequence of 3x3 complex products per wave: instruction
// count is latency.  Only used where the product is already associated differently from the
// sequential reference order (equal to it to rounding either way).
__device__ __forceinline__ void mat_mul_fma(const mat3 &A, const mat3 &B, mat3 &C) {
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++) {
            double re = A.m[i][0].re * B.m[0][j].re;
            double im = A.m[i][0].re * B.m[0][j].im;
            re = __builtin_fma(-A.m[i][0].im, B.m[0][j].im, re);
            im = __builtin_fma(A.m[i][0].im, B.m[0][j].re, im);
#pragma unroll
            for (int k = 1; k < 3; k++) {
                re = __builtin_fma(A.m[i][k].re, B.m[k][j].re, re);
                im = __builtin_fma(A.m[i][k].re, B.m[k][j].im, im);
                re = __builtin_fma(-A.m[i][k].im, B.m[k][j].im, re);
                im = __builtin_fma(A.m[i][k].im, B.m[k][j].re, im);
            }
            C.m[i][j] = cmake(re, im);
        }
}

// Two-sided form of stage C.  A row's chain, in path order a_0 .. a_{n-1}, is split at its
// middle element m = n/2:   T = (a_{n-1} .. a_{m+1}) . a_m . (a_{m-1} .. a_0).
// Step s pairs the out-going layer a_{m+s} with the in-going layer a_{m-s}; for the mirror
// symmetric paths through the Earth these are the SAME matrix (plan: one pair per distinct
// matrix of a row), so one load feeds two products,  L <- A . L  and  R <- R . A,  which are
// independent of each other (instruction-level parallelism where the one-sided form has a
// single dependent chain).  Wave g of the workgroup takes the g-th part of the steps; wave 0
// joins  L_{G-1} .. L_0 . a_m . R_0 .. R_{G-1}.  Valid for any sequence (non-mirrored steps just
// load two matrices); same matrices as the other forms, associated differently.
// AMP = 0: layer matrices read from stage AB's `amp`.  AMP = 1 (no decay) / 2 (decay): formed here
// from the stage-A records `terms` of the layer's density (a pair belongs to one row, so nothing
// is computed twice, and the records -- 2 x n_unique x 26 x n_e doubles -- stay in the L2 where
// the amplitudes, 12-18 doubles per pair and energy, had to travel through HBM).
// G = 0: PACKED launch.  Every workgroup has four waves and holds rows by their length -- one long
// row split over four waves, two medium rows over two waves each, or four short rows with a
// wave each (`blk`, built with the plan): what bounds this kernel is the dependent sequence of
// layer matrices per wave of the longest rows (13 for a 24-layer row, ~2 us each), and
// four-wave workgroups for EVERY row do not fit the chip at once (250 VGPRs: two workgroups per
// CU).  Packed, the ~360 workgroups of a 200 x 100 grid are all resident and a 24-layer row has
// four matrices per wave instead of seven.
// Development build (make EXTRA=-DPISA_CHAIN_STAMPS, scripts/dev/chain_stamps.py): wall-clock stamps of lane 0 of
// every wavefront of the packed one-point launch at seven points of the kernel.
#ifdef PISA_CHAIN_STAMPS
__device__ unsigned long long g_chain_stamps[8 * 4096];
#define CSTAMP(k) do { if (G == 0 && (threadIdx.x & 63) == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 1024) \
        g_chain_stamps[8 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + (k)] = wall_clock64(); } while (0)
#else
#define CSTAMP(k) do {} while (0)
#endif
template <int G, int AMP>
__device__ __forceinline__ void chain_tail(const Prob3Consts &c, const Prob3Side &S, mat3 &L, mat3 &R, mat3 &T, bool have_l, bool have_r,
                                           const double *s_part, int part_0, int lane, int Gr, int n_steps, int cnt, int mid,
                                           int e_major, int ie, int n_e, int n_cz, int jcz, int side, int n_points, int pt,
                                           double *__restrict__ out, double2 *__restrict__ pepmu);

template <int G, int AMP, class CS = ConstsByValue>
__global__ void __launch_bounds__(G ? 64 * G : 256)
prob3_chain_kernel(const CS cs, int n_e, const int32_t *__restrict__ row_start,
                    const int32_t *__restrict__ row_cnt, const int32_t *__restrict__ row_pairs,
                    int n_cz, int n_pairs, const double *__restrict__ amp, int e_major,
                    double *__restrict__ prob_nu, double *__restrict__ prob_nubar,
                    double2 *__restrict__ pepmu, const double *__restrict__ energy,
                    const int32_t *__restrict__ pair_u, const double *__restrict__ pair_dist,
                    int n_unique, const int32_t *__restrict__ blk, int n_points, int n_tiles,
                    unsigned long long *__restrict__ signal) {
    auto MM = [](const mat3 &A_, const mat3 &B_, mat3 &C_) { mat_mul_fma(A_, B_, C_); };
    CSTAMP(0);
    // Several points, packed launch (n_tiles > 0): a 1-D grid in which the (point, sign, energy tile)
    // index runs FASTEST and the row-block index slowest -- the plan lists the row blocks longest rows
    // first, so the long rows of every point, sign and tile start first and the short ones fill the tail
    // (with several points the workgroups no longer fit the chip at once).  The fast index is rotated by
    // the row block: consecutive workgroups go to consecutive XCDs, and without the rotation an XCD would
    // see the same few (point, sign, tile) records from all of its CUs at the same moment.  One point:
    // the 3-D grid (row block fastest), all workgroups resident at once.
    const bool lin = G == 0 && n_tiles > 0;
    const int n_yz = lin ? 2 * n_points * n_tiles : 1;
    const int bx = lin ? (int)(blockIdx.x / n_yz) : (int)blockIdx.x;
    const int byz = lin ? (int)((blockIdx.x + bx) % n_yz) : 0;
    const int by = lin ? byz % (2 * n_points) : (int)blockIdx.y;
    const int bz = lin ? byz / (2 * n_points) : (int)blockIdx.z;
    if (!lin) n_tiles = gridDim.z;
    const int pt = by >> 1;     // parameter point (0 in the one-point form)
    const Prob3Consts &c = cs.at(pt);
    constexpr bool PACKED = G == 0;
    __shared__ double s_part[(PACKED ? 3 : (G > 1 ? G - 1 : 1)) * 2 * 18 * 64];  // [partial][L|R][18][lane]
    const int lane = threadIdx.x & 63;
    // wave index: uniform within a wave, which the compiler cannot see -- without this the loop
    // bounds derived from it are 'divergent' and every index look-up becomes a vector load
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // row, group of this wave within the row, groups of the row; partial slot of a writer wave
    int jcz = bx, g = wv, Gr = G;
    int k0_blk = 0, cnt_blk = 0;
    if (PACKED) {
        // (row | g << 16 | groups << 24 (-1: idle wave), first position and length of the row's chain, 0): one 16-byte
        // scalar load instead of the code and then row_start / row_cnt behind it
        const int4 rec = reinterpret_cast<const int4 *>(blk)[bx * 4 + wv];
        const int32_t code = __builtin_amdgcn_readfirstlane(rec.x);
        k0_blk = __builtin_amdgcn_readfirstlane(rec.y);
        cnt_blk = __builtin_amdgcn_readfirstlane(rec.z);
        jcz = code < 0 ? -1 : (code & 0xffff);
        g = code < 0 ? 0 : ((code >> 16) & 0xff);
        Gr = code < 0 ? 1 : ((code >> 24) & 0xff);
    }
    const int part_w = PACKED ? wv - 1 : g - 1;   // where a wave with g > 0 leaves its partials
    const int part_0 = PACKED ? wv : 0;           // leader: partner h reads slot part_0 + h - 1
    const int side = by & 1;
    const bool decay = AMP == 0 ? c.decay != 0 : AMP == 2;  // full 3x3 matrices stored / formed
    const int ie = bz * 64 + lane;
    const bool live = ie < n_e && jcz >= 0;
    double *out = side == 0 ? prob_nu : prob_nubar;
    const Prob3Side &S = c.side[side];
    const int k0 = PACKED ? k0_blk : (jcz >= 0 ? row_start[jcz] : 0);
    const int cnt = PACKED ? cnt_blk : (jcz >= 0 ? row_cnt[jcz] : 0);
    const int mid = cnt >> 1;
    const int n_steps = mid;  // steps s = 1..mid (out-going side may be one shorter)
    const int s0 = 1 + (int)(((int64_t)n_steps * g) / Gr);
    const int s1 = 1 + (int)(((int64_t)n_steps * (g + 1)) / Gr);
    const int64_t ns = (int64_t)n_tiles * 64;
    // 1/E once per lane: L/E as a product (one rounding more than the quotient, 1e-16 on a phase)
    const double inv_e = (AMP != 0 && live) ? 1.0 / energy[ie] : 1.0;
    // `pos` = position in the row-pairs list.  In the AMP modes the density index and the length of
    // the layer are read from lists indexed by that position (chain_u / chain_dist) rather than
    // through the pair number: one dependent scalar load less in front of every layer matrix.
    auto load_pair = [&](int pos, mat3 &A) {
        if (AMP != 0) {
            // `amp` holds the stage-A records here
            const double *r = amp + ((int64_t)((pt * 2 + side) * n_unique + pair_u[pos]) * PROB3_NF) * ns + 2 * (int64_t)ie;
            auto load = [&](int f) { return r[(int64_t)(f >> 1) * (2 * ns) + (f & 1)]; };
            amplitude_from_terms<AMP == 2>(load, pair_dist[pos] * inv_e, A);
            if (AMP == 1) su3_complete(A);
            return;
        }
        const int k = row_pairs[pos];
        const double *a = amp + ((int64_t)(side * n_pairs + k) * 18) * ns + ie;
#pragma unroll
        for (int i = 0; i < (decay ? 3 : 2); i++)
#pragma unroll
            for (int j = 0; j < 3; j++)
                A.m[i][j] = cmake(a[(int64_t)(6 * i + 2 * j) * ns], a[(int64_t)(6 * i + 2 * j + 1) * ns]);
        if (!decay) su3_complete(A);
    };
    mat3 L, R;
    bool have_l = false, have_r = false;
    CSTAMP(1);
#ifdef PISA_CHAIN_STAMPS
    if (G == 0 && (threadIdx.x & 63) == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 1024)
        g_chain_stamps[8 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 7] = (unsigned long long)cnt | ((unsigned long long)g << 16) | ((unsigned long long)Gr << 24) | ((unsigned long long)(s1 - s0) << 32) | (jcz >= 0 ? 1ull << 48 : 0ull);
#endif
    if (live && cnt > 0 && s1 > s0) {
        mat3 A, An;
        load_pair(k0 + mid - s0, A);
        for (int s = s0; s < s1; s++) {
            const int k_in = row_pairs[k0 + mid - s];
            const int k_out = mid + s < cnt ? row_pairs[k0 + mid + s] : -1;
            if (s + 1 < s1) load_pair(k0 + mid - (s + 1), An);  // next in flight
            if (have_r) { mat3 t; MM(R, A, t); R = t; } else { R = A; have_r = true; }
            if (k_out >= 0) {
                if (k_out != k_in) load_pair(k0 + mid + s, A);  // not a mirrored pair (workgroup-uniform)
                if (have_l) { mat3 t; MM(A, L, t); L = t; } else { L = A; have_l = true; }
            }
            A = An;
        }
    }
    CSTAMP(2);
    if (g > 0 && live) {
        double *o = s_part + (size_t)part_w * 2 * 18 * 64 + lane;
        if (have_l) {
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    o[(6 * i + 2 * j) * 64] = L.m[i][j].re;
                    o[(6 * i + 2 * j + 1) * 64] = L.m[i][j].im;
                }
        }
        if (have_r) {
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    o[(18 + 6 * i + 2 * j) * 64] = R.m[i][j].re;
                    o[(18 + 6 * i + 2 * j + 1) * 64] = R.m[i][j].im;
                }
        }
    }
    // the leader forms the middle layer's matrix while it waits for its partners
    mat3 T;
    if (g == 0 && live && cnt > 0) load_pair(k0 + mid, T);
    __syncthreads();
    CSTAMP(3);
#ifdef PISA_DEV_PROBES
    if (signal) {
        // hand-over to a consumer kernel that is already resident (common.hpp, HandOver; guide: "valid forms", producer):
        // every wave waits for its own stores, workgroup barrier, ONE lane releases at agent scope and adds to its counter
        const unsigned lin = ((unsigned)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        if (g == 0 && live) chain_tail<G, AMP>(c, S, L, R, T, have_l, have_r, s_part, part_0, lane, Gr, n_steps, cnt, mid, e_major, ie, n_e, n_cz, jcz,
                                               side, n_points, pt, out, pepmu);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(signal + (lin & (HANDOVER_SLOTS - 1)), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
#endif
    if (g != 0 || !live) return;
    chain_tail<G, AMP>(c, S, L, R, T, have_l, have_r, s_part, part_0, lane, Gr, n_steps, cnt, mid, e_major, ie, n_e, n_cz, jcz, side, n_points, pt,
                       out, pepmu);
}

// the leader's part of the chain kernel: joins the partial products of its partner waves, closes the chain with the
// mixing matrices and stores the node's probabilities
template <int G, int AMP>
__device__ __forceinline__ void chain_tail(const Prob3Consts &c, const Prob3Side &S, mat3 &L, mat3 &R, mat3 &T, bool have_l, bool have_r,
                                           const double *s_part, int part_0, int lane, int Gr, int n_steps, int cnt, int mid,
                                           int e_major, int ie, int n_e, int n_cz, int jcz, int side, int n_points, int pt,
                                           double *__restrict__ out, double2 *__restrict__ pepmu) {
    auto MM = [](const mat3 &A_, const mat3 &B_, mat3 &C_) { mat_mul_fma(A_, B_, C_); };
    // wave 0: T_right = R_0 . R_1 .. (later groups further right), T_left = .. L_1 . L_0
    for (int h = 1; h < Gr; h++) {
        const int h0 = 1 + (int)(((int64_t)n_steps * h) / Gr);
        const int h1 = 1 + (int)(((int64_t)n_steps * (h + 1)) / Gr);
        if (h1 <= h0 || cnt == 0) continue;  // wave-uniform: group h had no steps
        const double *o = s_part + (size_t)(part_0 + h - 1) * 2 * 18 * 64 + lane;
        mat3 Ph;
        // right part of group h always exists when it had steps
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++)
                Ph.m[i][j] = cmake(o[(18 + 6 * i + 2 * j) * 64], o[(18 + 6 * i + 2 * j + 1) * 64]);
        if (have_r) { mat3 t; MM(R, Ph, t); R = t; } else { R = Ph; have_r = true; }
        // its left part exists unless its only step was the unpaired last one
        const bool h_has_l = mid + h0 < cnt;
        if (h_has_l) {
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++)
                    Ph.m[i][j] = cmake(o[(6 * i + 2 * j) * 64], o[(6 * i + 2 * j + 1) * 64]);
            if (have_l) { mat3 t; MM(Ph, L, t); L = t; } else { L = Ph; have_l = true; }
        }
    }
    CSTAMP(4);
    if (cnt > 0) {
        if (have_r) { mat3 t; MM(T, R, t); T = t; }
        if (have_l) { mat3 t; MM(L, T, t); T = t; }
    } else {
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) T.m[i][j] = cmake(0.0, 0.0);
    }
    mat3 t2, Tf;
    MM(T, S.Ud, t2);
    MM(S.U, t2, Tf);
    double P[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            P[3 * i + j] = Tf.m[j][i].re * Tf.m[j][i].re + Tf.m[j][i].im * Tf.m[j][i].im;
    CSTAMP(5);
    int64_t node = e_major ? (int64_t)ie * n_cz + jcz : (int64_t)jcz * n_e + ie;
    if (n_points > 1) {
        // several points: the gather tables of the points interleaved, [sign][flavour][node][point], so
        // that the multi-point fused kernel fetches an event's pairs of all points in one contiguous run
#pragma unroll
        for (int f = 0; f < 3; f++)
            pepmu[(((int64_t)side * 3 + f) * ((int64_t)n_e * n_cz) + node) * n_points + pt] = make_double2(P[f], P[3 + f]);
        return;
    }
    store_node(P, node, (int64_t)n_e * n_cz, side, out, pepmu);
    CSTAMP(6);
}

#ifdef PISA_CHAIN_STAMPS
extern "C" __attribute__((visibility("default"))) int pisa_hip_debug_chain_stamps(unsigned long long *h_out) {
    return check_hip(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_chain_stamps), sizeof(g_chain_stamps)), "stamps");
}
#endif

static int make_consts(const pisa_hip_prob3_params *p, Prob3Consts &c) {
    if (!p) return PISA_HIP_ERR_INVALID;
    prob3_make_consts(p->dm, p->mix, p->mat_pot, p->mat_decay, p->lri_pot, p->decay_flag, c);
    return PISA_HIP_OK;
}

}  // namespace pisa

using namespace pisa;

// ------------------------------------------------------------ grid plan (host)
struct pisa_hip_grid_plan {
    int n_cz, n_layers, n_unique, n_pairs, n_items, n_chain;
    int32_t *d_item_u;     // [n_items] distinct-density index of the item
    int32_t *d_item_p0;    // [n_items] first pair of the item
    int32_t *d_item_cnt;   // [n_items] pairs of the item (consecutive, same density)
    double *d_pair_dist;   // [n_pairs] (cache-resolved) layer length of each pair
    int32_t *d_row_start;  // [n_cz] first chain entry of the row
    int32_t *d_row_cnt;    // [n_cz] crossed layers of the row
    int32_t *d_row_pairs;  // [n_chain] pair index per crossed layer, in path order
    double *d_rho;         // [n_unique]
    double *d_amp;         // stage-AB amplitudes [2][n_pairs][18][n_e]
    int n_e_alloc;
    int32_t *d_pair_u;     // [n_pairs] distinct-density index of each pair
    int32_t *d_blk;        // [4 * n_blk][4] packed launch, per wave: row | group << 16 | groups << 24 (-1 idle), chain start, chain length, 0
    int n_blk;
    int chain_packed;      // 1 (default): packed chain launch; PISA_HIP_CHAIN_MODE=split: one row per workgroup
    int32_t *d_chain_u;    // [n_chain] the same per chain entry (position in d_row_pairs)
    double *d_chain_dist;  // [n_chain] layer length per chain entry
    double *d_terms;       // stage-A records [points][2][n_unique][PROB3_NF][n_e] (AMP mode)
    int n_e_terms;
    int n_pt_terms;        // parameter points d_terms has room for
    // several parameter points in one launch (pisa_hip_prob3_grid_planned_multi)
    Prob3Consts *d_consts;   // [PISA_HIP_MAX_POINTS] the points' constants in device memory
    Prob3Consts *h_consts;   // [MULTI_RING][PISA_HIP_MAX_POINTS] pinned, device-mapped staging blocks
    int consts_slot;
    int fused_amp;         // 1 (default): stage A + chain kernel forming the layer matrices itself;
                           // 0 (PISA_HIP_PROB3_FUSED_AMP=0 when the plan is created): stage AB
                           // stores the layer matrices, the chain kernel reads them
    // host copies for re-cutting the items when n_e changes
    int32_t *h_pair_u;
    int items_for_n_e;
};

PISA_API int pisa_hip_grid_plan_destroy(pisa_hip_grid_plan *p) {
    if (!p) return PISA_HIP_OK;
    void *ptrs[] = {p->d_item_u, p->d_item_p0, p->d_item_cnt, p->d_pair_dist, p->d_row_start,
                    p->d_row_cnt, p->d_row_pairs, p->d_rho, p->d_amp, p->d_pair_u, p->d_terms,
                    p->d_chain_u, p->d_chain_dist, p->d_blk};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    if (p->d_consts) (void)hipFree(p->d_consts);
    if (p->h_consts) (void)hipHostFree(p->h_consts);
    delete[] p->h_pair_u;
    delete p;
    return PISA_HIP_OK;
}

// Cut the (density-sorted) pairs into items of <= ch consecutive pairs of one density.
// ch is chosen so that stage AB has about 2000 waves: enough to occupy the chip
// with the ~8 us terms chain running once per wave.
static int cut_items(pisa_hip_grid_plan *p, int n_e) {
    const int tiles = (n_e + 63) / 64;
    int ch = (int)(((int64_t)p->n_pairs * 2 * tiles + 2047) / 2048);
    if (ch < 1) ch = 1;
    if (const int v = PISA_DEV_INT("PROB3_CH", 0); v > 0) ch = v;
    const int np = p->n_pairs;
    int32_t *iu = new int32_t[np + 1], *ip0 = new int32_t[np + 1], *icnt = new int32_t[np + 1];
    int ni = 0;
    for (int k = 0; k < np;) {
        int e = k;
        while (e < np && e - k < ch && p->h_pair_u[e] == p->h_pair_u[k]) e++;
        iu[ni] = p->h_pair_u[k]; ip0[ni] = k; icnt[ni] = e - k;
        ni++;
        k = e;
    }
    int rc = 0;
    for (void *q : {(void *)p->d_item_u, (void *)p->d_item_p0, (void *)p->d_item_cnt})
        if (q) (void)hipFree(q);
    p->d_item_u = p->d_item_p0 = p->d_item_cnt = nullptr;
    size_t bytes = (size_t)(ni > 0 ? ni : 1) * 4;
    rc = check_hip(hipMalloc(&p->d_item_u, bytes), "hipMalloc");
    if (!rc) rc = check_hip(hipMalloc(&p->d_item_p0, bytes), "hipMalloc");
    if (!rc) rc = check_hip(hipMalloc(&p->d_item_cnt, bytes), "hipMalloc");
    if (!rc && ni > 0) rc = check_hip(hipMemcpy(p->d_item_u, iu, (size_t)ni * 4, hipMemcpyHostToDevice), "h2d");
    if (!rc && ni > 0) rc = check_hip(hipMemcpy(p->d_item_p0, ip0, (size_t)ni * 4, hipMemcpyHostToDevice), "h2d");
    if (!rc && ni > 0) rc = check_hip(hipMemcpy(p->d_item_cnt, icnt, (size_t)ni * 4, hipMemcpyHostToDevice), "h2d");
    delete[] iu; delete[] ip0; delete[] icnt;
    p->n_items = ni;
    p->items_for_n_e = n_e;
    return rc;
}

PISA_API int pisa_hip_grid_plan_create(const double *d_densities, const double *d_distances,
                                       int32_t n_cz, int32_t n_layers,
                                       pisa_hip_grid_plan **out) {
    if (!out || n_cz < 1 || n_layers < 1 || !d_densities || !d_distances) return PISA_HIP_ERR_INVALID;
    if (n_layers > PISA_HIP_MAX_LAYERS) return PISA_HIP_ERR_LAYERS;
    size_t n = (size_t)n_cz * n_layers;
    double *rho = new double[n], *dist = new double[n], *pdist = new double[n + 1];
    int32_t *pu = new int32_t[n + 1], *rstart = new int32_t[n_cz], *rcnt = new int32_t[n_cz];
    int32_t *chain = new int32_t[n + 1], *layer_pair = new int32_t[n_layers];
    double *uniq = new double[n + 1];
    int nu = 0, np = 0, nc = 0;
    int rc = check_hip(hipMemcpy(rho, d_densities, n * 8, hipMemcpyDeviceToHost), "d2h");
    if (!rc) rc = check_hip(hipMemcpy(dist, d_distances, n * 8, hipMemcpyDeviceToHost), "d2h");
    if (!rc) {
        for (int r = 0; r < n_cz; r++) {
            const double *rr = rho + (size_t)r * n_layers, *dd = dist + (size_t)r * n_layers;
            rstart[r] = nc;
            for (int i = 0; i < n_layers; i++) {
                layer_pair[i] = -1;
                if (!(dd[i] > 0.0)) continue;
                // follow the reference's cache matches (numba_osc_kernels.py:236-249)
                // to the layer whose matrix is actually computed
                int cur = i;
                double cr = rr[i], cd = dd[i];
                while (true) {
                    int found = -1;
                    for (int j = 0; j < cur; j++)
                        if (dd[j] > 0.0 && fabs(rr[j] - cr) < 1e-5 && fabs(dd[j] - cd) < 1e-5) found = j;
                    if (found < 0) break;
                    cur = found; cr = rr[cur]; cd = dd[cur];
                }
                if (cur != i) {
                    layer_pair[i] = layer_pair[cur];  // the same matrix, stored once
                } else {
                    int u = -1;
                    for (int k = 0; k < nu; k++)
                        if (uniq[k] == cr) { u = k; break; }
                    if (u < 0) { u = nu; uniq[nu++] = cr; }
                    pu[np] = u;
                    pdist[np] = cd;
                    layer_pair[i] = np++;
                }
                chain[nc++] = layer_pair[i];
            }
            rcnt[r] = nc - rstart[r];
        }
    }
    // renumber the pairs sorted by density so that an item is a run of consecutive pairs
    int32_t *order = new int32_t[np + 1], *newid = new int32_t[np + 1];
    double *sdist = new double[np + 1];
    int32_t *su = new int32_t[np + 1];
    {
        int w = 0;
        for (int u = 0; u < nu; u++)
            for (int k = 0; k < np; k++)
                if (pu[k] == u) order[w++] = k;
        for (int k = 0; k < np; k++) { newid[order[k]] = k; sdist[k] = pdist[order[k]]; su[k] = pu[order[k]]; }
        for (int k = 0; k < nc; k++) chain[k] = newid[chain[k]];
    }
    pisa_hip_grid_plan *p = nullptr;
    if (!rc) {
        p = new pisa_hip_grid_plan();
        memset(p, 0, sizeof(*p));
        p->n_cz = n_cz; p->n_layers = n_layers;
        p->n_unique = nu > 0 ? nu : 1;
        p->n_pairs = np;
        p->n_chain = nc;
        p->h_pair_u = su;
        su = nullptr;
        p->fused_amp = PISA_DEV_INT("PROB3_FUSED_AMP", 1) != 0;
        if (nu == 0) uniq[0] = 0.0;
        size_t npa = np > 0 ? np : 1, nca = nc > 0 ? nc : 1;
        rc = check_hip(hipMalloc(&p->d_pair_dist, npa * 8), "hipMalloc");
        if (!rc) rc = check_hip(hipMalloc(&p->d_row_start, (size_t)n_cz * 4), "hipMalloc");
        if (!rc) rc = check_hip(hipMalloc(&p->d_row_cnt, (size_t)n_cz * 4), "hipMalloc");
        if (!rc) rc = check_hip(hipMalloc(&p->d_row_pairs, nca * 4), "hipMalloc");
        if (!rc) rc = check_hip(hipMalloc(&p->d_rho, (size_t)p->n_unique * 8), "hipMalloc");
        {
            // packed chain launch: rows by length, longest first; 4 waves per workgroup =
            // one long row x 4 groups | two medium rows x 2 | four short rows x 1
            const char *m = PISA_DEV_STR("CHAIN_MODE");
            p->chain_packed = (m && strcmp(m, "split") == 0) ? 0 : 1;
            int32_t *code = new int32_t[((size_t)4 * n_cz + 4) * 4];   // per wave: code, chain start, chain length, 0
            int nb = 0;
            int t4 = 14, t2 = 0;  // crossed layers from which a row gets four / two waves (measured: 22.1 us; 14/6: 23.4; one row per workgroup: 24.4)
            t4 = PISA_DEV_INT("PACK_T4", t4);
            t2 = PISA_DEV_INT("PACK_T2", t2);
            auto groups_of = [&](int c_) { return c_ >= t4 ? 4 : (c_ >= t2 ? 2 : 1); };
            for (int want = 4; want >= 1; want >>= 1) {
                int fill = 0;  // wave slots used in the open workgroup
                for (int r = 0; r < n_cz && n_cz <= 0xffff; r++) {
                    if (groups_of(rcnt[r]) != want) continue;
                    for (int gi = 0; gi < want; gi++) {
                        int32_t *w = code + (size_t)(nb * 4 + fill + gi) * 4;
                        w[0] = r | (gi << 16) | (want << 24); w[1] = rstart[r]; w[2] = rcnt[r]; w[3] = 0;
                    }
                    fill += want;
                    if (fill == 4) { nb++; fill = 0; }
                }
                if (fill > 0) {
                    for (int k = fill; k < 4; k++) {
                        int32_t *w = code + (size_t)(nb * 4 + k) * 4;
                        w[0] = -1; w[1] = w[2] = w[3] = 0;
                    }
                    nb++;
                }
            }
            if (n_cz > 0xffff) p->chain_packed = 0;
            p->n_blk = nb;
            if (!rc) rc = check_hip(hipMalloc(&p->d_blk, (size_t)(nb > 0 ? nb : 1) * 64), "hipMalloc");
            if (!rc && nb > 0) rc = check_hip(hipMemcpy(p->d_blk, code, (size_t)nb * 64, hipMemcpyHostToDevice), "h2d");
            delete[] code;
        }
        if (!rc) rc = check_hip(hipMalloc(&p->d_chain_u, nca * 4), "hipMalloc");
        if (!rc) rc = check_hip(hipMalloc(&p->d_chain_dist, nca * 8), "hipMalloc");
        if (!rc && nc > 0) {
            int32_t *cu = new int32_t[nc];
            double *cd = new double[nc];
            for (int k = 0; k < nc; k++) { cu[k] = p->h_pair_u[chain[k]]; cd[k] = sdist[chain[k]]; }
            rc = check_hip(hipMemcpy(p->d_chain_u, cu, (size_t)nc * 4, hipMemcpyHostToDevice), "h2d");
            if (!rc) rc = check_hip(hipMemcpy(p->d_chain_dist, cd, (size_t)nc * 8, hipMemcpyHostToDevice), "h2d");
            delete[] cu; delete[] cd;
        }
        if (!rc) rc = check_hip(hipMalloc(&p->d_pair_u, npa * 4), "hipMalloc");
        if (!rc && np > 0) rc = check_hip(hipMemcpy(p->d_pair_u, p->h_pair_u, (size_t)np * 4, hipMemcpyHostToDevice), "h2d");
        if (!rc && np > 0) rc = check_hip(hipMemcpy(p->d_pair_dist, sdist, (size_t)np * 8, hipMemcpyHostToDevice), "h2d");
        if (!rc) rc = check_hip(hipMemcpy(p->d_row_start, rstart, (size_t)n_cz * 4, hipMemcpyHostToDevice), "h2d");
        if (!rc) rc = check_hip(hipMemcpy(p->d_row_cnt, rcnt, (size_t)n_cz * 4, hipMemcpyHostToDevice), "h2d");
        if (!rc && nc > 0) rc = check_hip(hipMemcpy(p->d_row_pairs, chain, (size_t)nc * 4, hipMemcpyHostToDevice), "h2d");
        if (!rc) rc = check_hip(hipMemcpy(p->d_rho, uniq, (size_t)p->n_unique * 8, hipMemcpyHostToDevice), "h2d");
    }
    delete[] rho; delete[] dist; delete[] pdist; delete[] pu; delete[] rstart; delete[] rcnt; delete[] uniq;
    delete[] chain; delete[] layer_pair; delete[] order; delete[] newid; delete[] sdist; delete[] su;
    if (rc && p) { pisa_hip_grid_plan_destroy(p); p = nullptr; }
    *out = p;
    return rc;
}

// the two launches of the fused-amplitude form for `n_points` parameter points whose constants `cs`
// provides (by value: one point; by pointer: the plan's device array)
template <class CS>
static int launch_planned(const CS &cs, bool decay, int n_points, pisa_hip_grid_plan *plan, const double *d_energy,
                          int32_t n_e, int32_t e_major, double *d_prob_nu, double *d_prob_nubar, double *d_pepmu,
                          hipStream_t s) {
    const unsigned tiles = (unsigned)((n_e + 63) / 64);
    static const int groups = []() {
        const int g = PISA_DEV_INT("CHAIN_GROUPS", CHAIN_GROUPS_DEFAULT);
        return (g == 1 || g == 2 || g == 4) ? g : CHAIN_GROUPS_DEFAULT;
    }();
    if (plan->n_e_terms < n_e || plan->n_pt_terms < n_points) {
        if (plan->d_terms) (void)hipFree(plan->d_terms);
        plan->d_terms = nullptr;
        plan->n_e_terms = plan->n_pt_terms = 0;
        size_t bytes = (size_t)n_points * 2 * plan->n_unique * PROB3_NF * ((size_t)tiles * 64) * sizeof(double);
        PISA_TRY_HIP(hipMalloc(&plan->d_terms, bytes));
        plan->n_e_terms = n_e;
        plan->n_pt_terms = n_points;
    }
    dim3 tblock(64), tgrid((unsigned)plan->n_unique, 2u * n_points, tiles);
    if (decay)
        hipLaunchKernelGGL((prob3_terms_kernel<true, CS>), tgrid, tblock, 0, s, cs, d_energy, (int)n_e,
                           plan->d_rho, plan->n_unique, plan->d_terms);
    else
        hipLaunchKernelGGL((prob3_terms_kernel<false, CS>), tgrid, tblock, 0, s, cs, d_energy, (int)n_e,
                           plan->d_rho, plan->n_unique, plan->d_terms);
    PISA_CHECK_LAUNCH("prob3_terms_kernel");
    dim3 cblock(64 * groups), cgrid((unsigned)plan->n_cz, 2u * n_points, tiles);
#define CHAIN(G, A) hipLaunchKernelGGL((prob3_chain_kernel<G, A, CS>), cgrid, cblock, 0, s, cs, (int)n_e, plan->d_row_start, \
                       plan->d_row_cnt, plan->d_row_pairs, plan->n_cz, plan->n_pairs, plan->d_terms,              \
                       (int)e_major, d_prob_nu, d_prob_nubar, (double2 *)d_pepmu, d_energy, plan->d_chain_u,      \
                       plan->d_chain_dist, plan->n_unique, plan->d_blk, n_points, lin_tiles, sig)
    int lin_tiles = 0;
    unsigned long long *sig = nullptr;
    if (plan->chain_packed && plan->n_blk > 0) {
        cblock = dim3(256);
        cgrid = dim3((unsigned)plan->n_blk, 2u * n_points, tiles);
        if (n_points > 1) {
            lin_tiles = (int)tiles;
            cgrid = dim3((unsigned)plan->n_blk * 2u * n_points * tiles, 1, 1);
        }
#ifdef PISA_DEV_PROBES
        if (n_points == 1 && g_chain_signal.flags) {   // (evaluator.hip: the accumulate kernel of this evaluation polls these counters)
            sig = g_chain_signal.flags;
            g_chain_signal.n_wg = (int)(cgrid.x * cgrid.y * cgrid.z);
            g_chain_signal.epoch++;
        }
#endif
        if (decay) CHAIN(0, 2); else CHAIN(0, 1);
    } else if (decay) { if (groups == 1) CHAIN(1, 2); else if (groups == 4) CHAIN(4, 2); else CHAIN(2, 2); }
    else { if (groups == 1) CHAIN(1, 1); else if (groups == 4) CHAIN(4, 1); else CHAIN(2, 1); }
#undef CHAIN
    PISA_CHECK_LAUNCH("prob3_chain_kernel");
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_prob3_grid_planned(const pisa_hip_prob3_params *h_params,
                                         pisa_hip_grid_plan *plan, const double *d_energy,
                                         int32_t n_e, int32_t e_major, double *d_prob_nu,
                                         double *d_prob_nubar, double *d_pepmu, void *stream) {
    if (!plan || n_e < 1 || !d_energy) return PISA_HIP_ERR_INVALID;
    ConstsByValue cv;
    Prob3Consts &c = cv.c;
    int rc = make_consts(h_params, c);
    if (rc) return rc;
    hipStream_t s = as_stream(stream);
    const unsigned tiles = (unsigned)((n_e + 63) / 64);
    static const int groups = []() {
        const int g = PISA_DEV_INT("CHAIN_GROUPS", CHAIN_GROUPS_DEFAULT);
        return (g == 1 || g == 2 || g == 4) ? g : CHAIN_GROUPS_DEFAULT;
    }();
    const int fused_amp = plan->fused_amp;
    dim3 cblock(64 * groups), cgrid((unsigned)plan->n_cz, 2, tiles);
    if (fused_amp) return launch_planned(cv, c.decay != 0, 1, plan, d_energy, n_e, e_major, d_prob_nu, d_prob_nubar, d_pepmu, s);
    if (plan->n_e_alloc < n_e) {
        if (plan->d_amp) (void)hipFree(plan->d_amp);
        plan->d_amp = nullptr;
        plan->n_e_alloc = 0;
        size_t amp_bytes = (size_t)2 * (plan->n_pairs > 0 ? plan->n_pairs : 1) * 18 * (((size_t)n_e + 63) / 64 * 64) * sizeof(double);
        PISA_TRY_HIP(hipMalloc(&plan->d_amp, amp_bytes));
        plan->n_e_alloc = n_e;
    }
    if (plan->items_for_n_e != n_e && (rc = cut_items(plan, n_e))) return rc;
    dim3 ablock(64), agrid((unsigned)(plan->n_items > 0 ? plan->n_items : 1), 2, tiles);
    if (plan->n_items > 0) {
        if (c.decay)
            hipLaunchKernelGGL(prob3_terms_amp_kernel<true>, agrid, ablock, 0, s, c, d_energy, (int)n_e,
                               plan->d_rho, plan->d_item_u, plan->d_item_p0, plan->d_item_cnt,
                               plan->d_pair_dist, plan->n_pairs, plan->d_amp);
        else
            hipLaunchKernelGGL(prob3_terms_amp_kernel<false>, agrid, ablock, 0, s, c, d_energy, (int)n_e,
                               plan->d_rho, plan->d_item_u, plan->d_item_p0, plan->d_item_cnt,
                               plan->d_pair_dist, plan->n_pairs, plan->d_amp);
    }
#define CHAIN(G) hipLaunchKernelGGL((prob3_chain_kernel<G, 0, ConstsByValue>), cgrid, cblock, 0, s, cv, (int)n_e, plan->d_row_start, \
                       plan->d_row_cnt, plan->d_row_pairs, plan->n_cz, plan->n_pairs, plan->d_amp,          \
                       (int)e_major, d_prob_nu, d_prob_nubar, (double2 *)d_pepmu, d_energy, plan->d_pair_u,  \
                       plan->d_pair_dist, plan->n_unique, plan->d_blk, 1, 0, (unsigned long long *)nullptr)
    if (groups == 1) CHAIN(1); else if (groups == 4) CHAIN(4); else CHAIN(2);
#undef CHAIN
    PISA_CHECK_LAUNCH("prob3_chain_kernel");
    return PISA_HIP_OK;
}

// Staging blocks for the constants of a batch: the host fills one (pinned, device-mapped) and a small
// kernel copies it into the device array the batch kernels read.  A ring, so that a caller that queues
// a second batch before the first has started does not overwrite what the first copy has yet to read.
constexpr int MULTI_RING = 8;

PISA_API int pisa_hip_prob3_grid_planned_multi(const pisa_hip_prob3_params *h_params, int32_t n_points,
                                               pisa_hip_grid_plan *plan, const double *d_energy,
                                               int32_t n_e, int32_t e_major, double *d_pepmu_points,
                                               void *stream) {
    if (!plan || n_e < 1 || !d_energy || !h_params || !d_pepmu_points || n_points < 1 ||
        n_points > PISA_HIP_MAX_POINTS)
        return PISA_HIP_ERR_INVALID;
    if (!plan->fused_amp) return PISA_HIP_ERR_INVALID;   // the stored-amplitude option has no batch form
    hipStream_t s = as_stream(stream);
    if (!plan->d_consts) {
        PISA_TRY_HIP(hipMalloc(&plan->d_consts, sizeof(Prob3Consts) * PISA_HIP_MAX_POINTS));
        PISA_TRY_HIP(hipHostMalloc(&plan->h_consts, sizeof(Prob3Consts) * PISA_HIP_MAX_POINTS * MULTI_RING,
                                   hipHostMallocMapped));
        plan->consts_slot = 0;
    }
    Prob3Consts *hc = plan->h_consts + (size_t)plan->consts_slot * PISA_HIP_MAX_POINTS;
    plan->consts_slot = (plan->consts_slot + 1) % MULTI_RING;
    bool decay = false;
    for (int k = 0; k < n_points; k++) {
        int rc = make_consts(h_params + k, hc[k]);
        if (rc) return rc;
        if (k == 0) decay = hc[k].decay != 0;
        else if ((hc[k].decay != 0) != decay) return PISA_HIP_ERR_INVALID;   // one kernel variant per batch
    }
    static_assert(sizeof(Prob3Consts) % 8 == 0, "copied in 8-byte words");
    const int n_words = (int)(sizeof(Prob3Consts) / 8) * n_points;
    hipLaunchKernelGGL(prob3_stage_consts_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const unsigned long long *>(hc),
                       reinterpret_cast<unsigned long long *>(plan->d_consts), n_words);
    PISA_CHECK_LAUNCH("prob3_stage_consts_kernel");
    ConstsByPointer cp{plan->d_consts};
    if (n_points == 1) {
        // a batch of one writes the ordinary [sign][flavour][node] tables
        return launch_planned(cp, decay, 1, plan, d_energy, n_e, e_major, nullptr, nullptr, d_pepmu_points, s);
    }
    return launch_planned(cp, decay, (int)n_points, plan, d_energy, n_e, e_major, nullptr, nullptr,
                          d_pepmu_points, s);
}
