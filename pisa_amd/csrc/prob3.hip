// prob3.hip -- oscillation-probability kernels for gfx950 + their C ABI.
//
//   prob3_array_kernel   generic `propagate_array` contract: one thread per
//                        element, layers from [N][L] (or one shared row)
//   prob3_grid_kernel    (E x coszen) grid: workgroup = one coszen row of one
//                        nu/nubar sign, lanes run along energy, so the layer
//                        list (length, cache matches) is workgroup-uniform and
//                        comes from scalar loads -- no divergence
//   calc_layers_kernel   extCalcLayers (layers.py:38-169), one thread / coszen
//   prob3_events_kernel  event mode: layers rebuilt per event from coszen with
//                        the PREM shell table in LDS (no [N][L] arrays in HBM)
//
// Roofline: all of these are FP64-VALU / transcendental bound (~16 kflop per
// element against <= 88 B of traffic); there is no dense contraction, so no MFMA.
#include <string.h>

#include "common.hpp"
#include "prob3_device.hpp"

namespace pisa {

// P[init][final] of one node -> full matrix and/or the compact gather tables
// pepmu[side][flav][node] = (P[e->flav], P[mu->flav]) read by the fused kernel
__device__ __forceinline__ void store_node(const double (&P)[9], int64_t node, int64_t n_nodes,
                                           int side, double *__restrict__ out,
                                           double2 *__restrict__ pepmu) {
    if (out) {
#pragma unroll
        for (int k = 0; k < 9; k++) out[9 * node + k] = P[k];
    }
    if (pepmu) {
#pragma unroll
        for (int f = 0; f < 3; f++)
            pepmu[((int64_t)side * 3 + f) * n_nodes + node] = make_double2(P[f], P[3 + f]);
    }
}

// ----------------------------------------------------------------- array form
template <bool DECAY>
__global__ void __launch_bounds__(256)
prob3_array_kernel(const Prob3Consts c, int side, const double *__restrict__ energy,
                   const double *__restrict__ densities, const double *__restrict__ distances,
                   int64_t n, int n_layers, int64_t row_stride, double *__restrict__ prob) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *rho_row = densities + i * row_stride;
    const double *dist_row = distances + i * row_stride;
    auto layer = [&](int l, double &rho, double &dist) {
        rho = rho_row[l];
        dist = dist_row[l];
    };
    double P[9];
    propagate_element<DECAY>(c.side[side], c.dm, energy[i], n_layers, layer, P);
#pragma unroll
    for (int k = 0; k < 9; k++) prob[9 * i + k] = P[k];
}

// ------------------------------------------------------------------ grid form
// blockIdx.x = coszen row, blockIdx.y = 0 (nu) / 1 (nubar), threads over energy.
template <bool DECAY>
__global__ void __launch_bounds__(256)
prob3_grid_kernel(const Prob3Consts c, const double *__restrict__ energy, int n_e,
                  const double *__restrict__ densities, const double *__restrict__ distances,
                  int n_cz, int n_layers, int e_major, double *__restrict__ prob_nu,
                  double *__restrict__ prob_nubar, double2 *__restrict__ pepmu) {
    const int jcz = blockIdx.x;
    const int side = blockIdx.y;
    double *out = side == 0 ? prob_nu : prob_nubar;
    if (out == nullptr && pepmu == nullptr) return;
    // workgroup-uniform row pointers -> scalar loads
    const double *rho_row = densities + (int64_t)jcz * n_layers;
    const double *dist_row = distances + (int64_t)jcz * n_layers;
    auto layer = [&](int l, double &rho, double &dist) {
        rho = rho_row[l];
        dist = dist_row[l];
    };
    for (int ie = threadIdx.x; ie < n_e; ie += blockDim.x) {
        double P[9];
        propagate_element<DECAY>(c.side[side], c.dm, energy[ie], n_layers, layer, P);
        int64_t node = e_major ? (int64_t)ie * n_cz + jcz : (int64_t)jcz * n_e + ie;
        store_node(P, node, (int64_t)n_e * n_cz, side, out, pepmu);
    }
}

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
//   stage AB  wave = (item, sign, 64 energies): terms of (E, rho) in registers, then
//             A = sum_k phase_k Q_k for the item's pairs -> amp[side][pair][18][n_e]
//   stage C   workgroup = (row, sign, 64 energies) x G waves: the chain is multiplied from
//             its middle outwards on both sides at once (see prob3_chain_kernel), wave 0
//             joins the waves' partial products (LDS), rotates to the flavour basis and
//             stores P and the gather tables.
// Stage C associates the product differently from the sequential reference and uses
// fused multiply-adds, so its results agree with prob3_grid_kernel to rounding
// (<= 1e-13 absolute on the probabilities), not bit for bit.
constexpr int CHAIN_GROUPS_DEFAULT = 2;

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
    eigen_terms<DECAY>(c.side[side], c.dm, e, rho_unique[item_u[item]], store);
    auto load = [&](int f) { return rec[f]; };
    const int p0 = item_p0[item], cnt = item_cnt[item];
    for (int q = 0; q < cnt; q++) {
        mat3 A;
        amplitude_from_terms<DECAY>(load, pair_dist[p0 + q] / e, A);
        const int64_t ns = (int64_t)gridDim.z * 64;  // energy stride: whole tiles, cache-line aligned
        double *o = amp + ((int64_t)(side * n_pairs + p0 + q) * 18) * ns + ie;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                o[(int64_t)(6 * i + 2 * j) * ns] = A.m[i][j].re;
                o[(int64_t)(6 * i + 2 * j + 1) * ns] = A.m[i][j].im;
            }
    }
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
template <int G>
__global__ void __launch_bounds__(64 * G)
prob3_chain_kernel(const Prob3Consts c, int n_e, const int32_t *__restrict__ row_start,
                    const int32_t *__restrict__ row_cnt, const int32_t *__restrict__ row_pairs,
                    int n_cz, int n_pairs, const double *__restrict__ amp, int e_major,
                    double *__restrict__ prob_nu, double *__restrict__ prob_nubar,
                    double2 *__restrict__ pepmu) {
    auto MM = [](const mat3 &A_, const mat3 &B_, mat3 &C_) { mat_mul_fma(A_, B_, C_); };
    __shared__ double s_part[(G > 1 ? G - 1 : 1) * 2 * 18 * 64];  // [group-1][L|R][18][lane]
    const int jcz = blockIdx.x;
    const int side = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int g = threadIdx.x >> 6;
    const int ie = blockIdx.z * 64 + lane;
    const bool live = ie < n_e;
    double *out = side == 0 ? prob_nu : prob_nubar;
    const Prob3Side &S = c.side[side];
    const int k0 = row_start[jcz];
    const int cnt = row_cnt[jcz];
    const int mid = cnt >> 1;
    const int n_steps = mid;  // steps s = 1..mid (out-going side may be one shorter)
    const int s0 = 1 + (int)(((int64_t)n_steps * g) / G);
    const int s1 = 1 + (int)(((int64_t)n_steps * (g + 1)) / G);
    const int64_t ns = (int64_t)gridDim.z * 64;
    auto load_pair = [&](int k, mat3 &A) {
        const double *a = amp + ((int64_t)(side * n_pairs + k) * 18) * ns + ie;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++)
                A.m[i][j] = cmake(a[(int64_t)(6 * i + 2 * j) * ns], a[(int64_t)(6 * i + 2 * j + 1) * ns]);
    };
    mat3 L, R;
    bool have_l = false, have_r = false;
    if (live && cnt > 0 && s1 > s0) {
        mat3 A, An;
        load_pair(row_pairs[k0 + mid - s0], A);
        for (int s = s0; s < s1; s++) {
            const int k_in = row_pairs[k0 + mid - s];
            const int k_out = mid + s < cnt ? row_pairs[k0 + mid + s] : -1;
            if (s + 1 < s1) load_pair(row_pairs[k0 + mid - (s + 1)], An);  // next in flight
            if (have_r) { mat3 t; MM(R, A, t); R = t; } else { R = A; have_r = true; }
            if (k_out >= 0) {
                if (k_out != k_in) load_pair(k_out, A);  // not a mirrored pair (workgroup-uniform)
                if (have_l) { mat3 t; MM(A, L, t); L = t; } else { L = A; have_l = true; }
            }
            A = An;
        }
    }
    if (g > 0 && live) {
        double *o = s_part + (size_t)(g - 1) * 2 * 18 * 64 + lane;
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
    __syncthreads();
    if (g != 0 || !live) return;
    // wave 0: T_right = R_0 . R_1 .. (later groups further right), T_left = .. L_1 . L_0
    for (int h = 1; h < G; h++) {
        const int h0 = 1 + (int)(((int64_t)n_steps * h) / G);
        const int h1 = 1 + (int)(((int64_t)n_steps * (h + 1)) / G);
        if (h1 <= h0 || cnt == 0) continue;  // workgroup-uniform: group h had no steps
        const double *o = s_part + (size_t)(h - 1) * 2 * 18 * 64 + lane;
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
    mat3 T;
    if (cnt > 0) {
        load_pair(row_pairs[k0 + mid], T);
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
    int64_t node = e_major ? (int64_t)ie * n_cz + jcz : (int64_t)jcz * n_e + ie;
    store_node(P, node, (int64_t)n_e * n_cz, side, out, pepmu);
}

// ------------------------------------------------------------------- layers
struct EarthDev {
    int32_t n_shell;
    int32_t idx;  // first shell with radius < r_detector (layers.py:90)
    double r_detector;
    double radii[PISA_HIP_MAX_SHELLS];
    double rhos[PISA_HIP_MAX_SHELLS];
    double coszen_limit[PISA_HIP_MAX_SHELLS];
};

// Geometry of one path (layers.py:86-159), evaluated lazily:
// segment(i) returns the i-th (rho, length) in path order, production -> detector.
struct PathGeom {
    double coszen, neg_rd_cz, base;  // base = rd^2 cz^2 - rd^2
    int m;                            // shells crossed (coszen_limit > coszen)
    int nseg;
    bool tangent_free;                // case A of layers.py:94
};

template <class E>
__device__ __forceinline__ double root_term(const E &e, const PathGeom &g, int k) {
    return sqrt(g.base + e.radii[k] * e.radii[k]);
}

template <class E>
__device__ __forceinline__ PathGeom make_path(const E &e, double coszen) {
    PathGeom g;
    g.coszen = coszen;
    double rd = e.r_detector;
    double rd2 = rd * rd;
    g.neg_rd_cz = -rd * coszen;
    g.base = rd2 * (coszen * coszen) - rd2;
    g.tangent_free = coszen >= e.coszen_limit[e.idx];
    int m = 0;
    for (int k = 0; k < e.n_shell; k++) m += (e.coszen_limit[k] > coszen) ? 1 : 0;
    g.m = m;
    g.nseg = g.tangent_free ? e.idx : (2 * m - 2);
    return g;
}

// returns false if the reference's own construction breaks down for this path
template <class E>
__device__ __forceinline__ bool path_valid(const E &e, const PathGeom &g) {
    if (g.tangent_free) return true;
    // densities list has 2m-2 entries, segments 2m-idx (layers.py:148-158)
    return e.idx == 2 && g.m >= 3;
}

template <class E>
__device__ __forceinline__ void path_segment(const E &e, const PathGeom &g, int i, double &rho,
                                             double &len) {
    if (g.tangent_free) {
        // cumulative distance to shell k's outer radius, k < idx (layers.py:95-101)
        double ck = g.neg_rd_cz + root_term(e, g, i);
        double prev = (i == e.idx - 1) ? 0.0 : (g.neg_rd_cz + root_term(e, g, i + 1));
        len = ck - prev;
        rho = e.rhos[i] * (len > 0. ? 1.0 : 0.0);
        return;
    }
    const int m = g.m;
    int shell;
    if (i < m - 1) {  // far side, going in: l_i - l_{i+1}
        len = (g.neg_rd_cz + root_term(e, g, i)) - (g.neg_rd_cz + root_term(e, g, i + 1));
        shell = i;
    } else if (i == m - 1) {  // innermost chord: l_{m-1} - s_{m-1}
        double t = root_term(e, g, m - 1);
        len = (g.neg_rd_cz + t) - (g.neg_rd_cz - t);
        shell = m - 1;
    } else {  // near side, coming out: s_{sh+1} - s_sh  (s_1 := 0 at the detector)
        shell = 2 * m - 2 - i;
        double hi = g.neg_rd_cz - root_term(e, g, shell + 1);
        double lo = (shell >= e.idx) ? (g.neg_rd_cz - root_term(e, g, shell)) : 0.0;
        len = hi - lo;
    }
    rho = e.rhos[shell] * (len > 0. ? 1.0 : 0.0);
}

__global__ void __launch_bounds__(256)
calc_layers_kernel(const EarthDev e, const double *__restrict__ cz, int64_t n, int max_layers,
                   double *__restrict__ n_layers_out, double *__restrict__ densities,
                   double *__restrict__ distances, int32_t *__restrict__ status) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    PathGeom g = make_path(e, cz[i]);
    double *rho_row = densities + i * max_layers;
    double *len_row = distances + i * max_layers;
    bool ok = path_valid(e, g);
    if (!ok && status) atomicOr(status, 1);
    double cnt = 0.0;
    for (int l = 0; l < max_layers; l++) {
        double rho = 0.0, len = 0.0;
        if (ok && l < g.nseg) path_segment(e, g, l, rho, len);
        if (!(len == len)) {  // NaN: sqrt of a negative argument
            if (status) atomicOr(status, 1);
            len = 0.0; rho = 0.0;
        }
        rho_row[l] = rho;
        len_row[l] = len;
        cnt += (len > 0.) ? 1.0 : 0.0;
    }
    if (n_layers_out) n_layers_out[i] = cnt;
}

// ---------------------------------------------------------------- event mode
// Per-thread path staged in LDS as [layer][lane] (conflict free): length (f64)
// and shell index (u8).  Dynamic LDS = blockDim.x * max_seg * 9 bytes + table.
constexpr int EV_MAX_CONT = 16;
struct EvCont {
    int64_t n;
    const double *energy, *coszen;
    double *prob;      // [n][3][3] or NULL
    double2 *pepmu;    // [n] (P[e->flav], P[mu->flav]) or NULL
    int32_t side, flav;
};
struct EvArgs {
    int32_t n_cont;
    int32_t blk_start[EV_MAX_CONT + 1];
    EvCont cont[EV_MAX_CONT];
};

template <bool DECAY>
__global__ void __launch_bounds__(128)
prob3_events_kernel(const Prob3Consts c, const EarthDev earth, const EvArgs ev, int max_seg,
                    int32_t *__restrict__ status) {
    int ci = 0;
    while (ci + 1 < ev.n_cont && (int)blockIdx.x >= ev.blk_start[ci + 1]) ci++;  // workgroup-uniform
    const EvCont &C = ev.cont[ci];
    const int side = C.side;
    const double *__restrict__ energy = C.energy;
    const double *__restrict__ coszen = C.coszen;
    const int64_t n = C.n;
    double *__restrict__ prob = C.prob;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // shell table in LDS (radii, rhos, coszen_limit)
    double *s_radii = reinterpret_cast<double *>(smem);
    double *s_rhos = s_radii + PISA_HIP_MAX_SHELLS;
    double *s_lim = s_rhos + PISA_HIP_MAX_SHELLS;
    double *s_len = s_lim + PISA_HIP_MAX_SHELLS;                       // [max_seg][blockDim]
    unsigned char *s_shell = reinterpret_cast<unsigned char *>(s_len + (size_t)max_seg * blockDim.x);
    for (int k = threadIdx.x; k < earth.n_shell; k += blockDim.x) {
        s_radii[k] = earth.radii[k];
        s_rhos[k] = earth.rhos[k];
        s_lim[k] = earth.coszen_limit[k];
    }
    __syncthreads();
    struct LdsEarth {
        int32_t n_shell, idx;
        double r_detector;
        const double *radii, *rhos, *coszen_limit;
    } e{earth.n_shell, earth.idx, earth.r_detector, s_radii, s_rhos, s_lim};

    int64_t i = (int64_t)(blockIdx.x - ev.blk_start[ci]) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int lane = threadIdx.x;
    const int bd = blockDim.x;
    PathGeom g = make_path(e, coszen[i]);
    bool ok = path_valid(e, g);
    int nseg = ok ? g.nseg : 0;
    if (!ok && status) atomicOr(status, 1);
    for (int l = 0; l < nseg; l++) {
        double rho, len;
        path_segment(e, g, l, rho, len);
        int shell = g.tangent_free ? l : (l < g.m ? l : 2 * g.m - 2 - l);
        if (!(len == len)) { len = 0.0; if (status) atomicOr(status, 1); }
        s_len[(size_t)l * bd + lane] = len;
        s_shell[(size_t)l * bd + lane] = (unsigned char)shell;
    }
    auto layer = [&](int l, double &rho, double &dist) {
        dist = s_len[(size_t)l * bd + lane];
        rho = s_rhos[s_shell[(size_t)l * bd + lane]] * (dist > 0. ? 1.0 : 0.0);
    };
    double P[9];
    // through-going paths: in 0..m-2, innermost m-1, out m..2m-3 (path_segment)
    propagate_path_nested<DECAY>(c.side[side], c.dm, energy[i], nseg,
                                 (ok && !g.tangent_free) ? g.m - 1 : -1, layer, P);
    if (prob) {
#pragma unroll
        for (int k = 0; k < 9; k++) prob[9 * i + k] = P[k];
    }
    if (C.pepmu) C.pepmu[i] = make_double2(P[C.flav], P[3 + C.flav]);
}

__global__ void fill_probs_kernel(const double *__restrict__ prob, int init_flav, int flav,
                                  int64_t n, double *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = prob[9 * i + 3 * init_flav + flav];
}

static int make_earth_dev(const pisa_hip_earth *h, EarthDev &e) {
    if (!h || h->n_shell < 2 || h->n_shell > PISA_HIP_MAX_SHELLS) return PISA_HIP_ERR_INVALID;
    e.n_shell = h->n_shell;
    e.r_detector = h->r_detector;
    e.idx = -1;
    for (int k = 0; k < PISA_HIP_MAX_SHELLS; k++) {
        e.radii[k] = k < h->n_shell ? h->radii[k] : 0.0;
        e.rhos[k] = k < h->n_shell ? h->rhos[k] : 0.0;
        e.coszen_limit[k] = k < h->n_shell ? h->coszen_limit[k] : -2.0;
    }
    for (int k = 0; k < h->n_shell; k++)
        if (h->radii[k] < h->r_detector) { e.idx = k; break; }
    if (e.idx < 1) return PISA_HIP_ERR_GEOMETRY;
    return PISA_HIP_OK;
}

static int make_consts(const pisa_hip_prob3_params *p, Prob3Consts &c) {
    if (!p) return PISA_HIP_ERR_INVALID;
    prob3_make_consts(p->dm, p->mix, p->mat_pot, p->mat_decay, p->lri_pot, p->decay_flag, c);
    return PISA_HIP_OK;
}

}  // namespace pisa

using namespace pisa;

PISA_API int pisa_hip_propagate_array(const pisa_hip_prob3_params *h_params, int64_t nubar,
                                      const double *d_energy, const double *d_densities,
                                      const double *d_distances, int64_t n, int32_t n_layers,
                                      int32_t layers_per_element, double *d_probability,
                                      void *stream) {
    if (n < 0 || n_layers < 0 || (nubar != 1 && nubar != -1)) return PISA_HIP_ERR_INVALID;
    if (n_layers > PISA_HIP_MAX_LAYERS) return PISA_HIP_ERR_LAYERS;
    if (n == 0) return PISA_HIP_OK;
    if (!d_energy || !d_probability || (n_layers > 0 && (!d_densities || !d_distances)))
        return PISA_HIP_ERR_INVALID;
    Prob3Consts c;
    int rc = make_consts(h_params, c);
    if (rc) return rc;
    int side = nubar > 0 ? 0 : 1;
    int64_t stride = layers_per_element ? n_layers : 0;
    dim3 block(256), grid((unsigned)((n + 255) / 256));
    if (c.decay)
        hipLaunchKernelGGL(prob3_array_kernel<true>, grid, block, 0, as_stream(stream), c, side,
                           d_energy, d_densities, d_distances, n, (int)n_layers, stride,
                           d_probability);
    else
        hipLaunchKernelGGL(prob3_array_kernel<false>, grid, block, 0, as_stream(stream), c, side,
                           d_energy, d_densities, d_distances, n, (int)n_layers, stride,
                           d_probability);
    PISA_CHECK_LAUNCH("prob3_array_kernel");
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_propagate_array_host(const pisa_hip_prob3_params *h_params, int64_t nubar,
                                           const double *h_energy, const double *h_densities,
                                           const double *h_distances, int64_t n, int32_t n_layers,
                                           int32_t layers_per_element, double *h_probability) {
    if (n < 0 || n_layers < 0) return PISA_HIP_ERR_INVALID;
    if (n_layers > PISA_HIP_MAX_LAYERS) return PISA_HIP_ERR_LAYERS;
    if (n == 0) return PISA_HIP_OK;
    size_t ne = (size_t)n * sizeof(double);
    size_t nl = (size_t)(layers_per_element ? n : 1) * (size_t)n_layers * sizeof(double);
    double *d_e = nullptr, *d_rho = nullptr, *d_len = nullptr, *d_p = nullptr;
    int rc = PISA_HIP_OK;
    hipStream_t s = nullptr;
    do {
        if ((rc = check_hip(hipMalloc(&d_e, ne), "hipMalloc"))) break;
        if ((rc = check_hip(hipMalloc(&d_rho, nl ? nl : 8), "hipMalloc"))) break;
        if ((rc = check_hip(hipMalloc(&d_len, nl ? nl : 8), "hipMalloc"))) break;
        if ((rc = check_hip(hipMalloc(&d_p, 9 * ne), "hipMalloc"))) break;
        if ((rc = check_hip(hipMemcpyAsync(d_e, h_energy, ne, hipMemcpyHostToDevice, s), "h2d"))) break;
        if (nl) {
            if ((rc = check_hip(hipMemcpyAsync(d_rho, h_densities, nl, hipMemcpyHostToDevice, s), "h2d"))) break;
            if ((rc = check_hip(hipMemcpyAsync(d_len, h_distances, nl, hipMemcpyHostToDevice, s), "h2d"))) break;
        }
        if ((rc = pisa_hip_propagate_array(h_params, nubar, d_e, d_rho, d_len, n, n_layers,
                                           layers_per_element, d_p, s))) break;
        if ((rc = check_hip(hipMemcpyAsync(h_probability, d_p, 9 * ne, hipMemcpyDeviceToHost, s), "d2h"))) break;
        rc = check_hip(hipStreamSynchronize(s), "sync");
    } while (0);
    if (d_e) (void)hipFree(d_e);
    if (d_rho) (void)hipFree(d_rho);
    if (d_len) (void)hipFree(d_len);
    if (d_p) (void)hipFree(d_p);
    return rc;
}

PISA_API int pisa_hip_prob3_grid(const pisa_hip_prob3_params *h_params, const double *d_energy,
                                 int32_t n_e, const double *d_densities,
                                 const double *d_distances, int32_t n_cz, int32_t n_layers,
                                 int32_t e_major, double *d_prob_nu, double *d_prob_nubar,
                                 double *d_pepmu, void *stream) {
    if (n_e < 0 || n_cz < 0 || n_layers < 0) return PISA_HIP_ERR_INVALID;
    if (n_layers > PISA_HIP_MAX_LAYERS) return PISA_HIP_ERR_LAYERS;
    if (n_e == 0 || n_cz == 0) return PISA_HIP_OK;
    if (!d_energy || !d_densities || !d_distances) return PISA_HIP_ERR_INVALID;
    Prob3Consts c;
    int rc = make_consts(h_params, c);
    if (rc) return rc;
    int threads = ((n_e + 63) / 64) * 64;
    if (threads > 256) threads = 256;
    dim3 block(threads), grid((unsigned)n_cz, 2);
    if (c.decay)
        hipLaunchKernelGGL(prob3_grid_kernel<true>, grid, block, 0, as_stream(stream), c, d_energy,
                           (int)n_e, d_densities, d_distances, (int)n_cz, (int)n_layers,
                           (int)e_major, d_prob_nu, d_prob_nubar, (double2 *)d_pepmu);
    else
        hipLaunchKernelGGL(prob3_grid_kernel<false>, grid, block, 0, as_stream(stream), c, d_energy,
                           (int)n_e, d_densities, d_distances, (int)n_cz, (int)n_layers,
                           (int)e_major, d_prob_nu, d_prob_nubar, (double2 *)d_pepmu);
    PISA_CHECK_LAUNCH("prob3_grid_kernel");
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_calc_layers(const pisa_hip_earth *h_earth, const double *d_coszen, int64_t n,
                                  int32_t max_layers, double *d_n_layers, double *d_densities,
                                  double *d_distances, int32_t *d_status, void *stream) {
    EarthDev e;
    int rc = make_earth_dev(h_earth, e);
    if (rc) return rc;
    if (n < 0 || max_layers < 2 * e.n_shell) return PISA_HIP_ERR_INVALID;
    if (n == 0) return PISA_HIP_OK;
    if (!d_coszen || !d_densities || !d_distances) return PISA_HIP_ERR_INVALID;
    dim3 block(256), grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(calc_layers_kernel, grid, block, 0, as_stream(stream), e, d_coszen, n,
                       (int)max_layers, d_n_layers, d_densities, d_distances, d_status);
    PISA_CHECK_LAUNCH("calc_layers_kernel");
    return PISA_HIP_OK;
}

static int launch_events(const Prob3Consts &c, const EarthDev &e, const EvCont *conts, int n_cont,
                         int32_t *d_status, hipStream_t s) {
    int max_seg = 2 * e.n_shell;
    if (max_seg > PISA_HIP_MAX_LAYERS + 8) return PISA_HIP_ERR_LAYERS;
    const int threads = 64;
    size_t lds = 3 * PISA_HIP_MAX_SHELLS * sizeof(double) + (size_t)max_seg * threads * 9 + 16;
    for (int base = 0; base < n_cont; base += EV_MAX_CONT) {
        int nc = n_cont - base < EV_MAX_CONT ? n_cont - base : EV_MAX_CONT;
        EvArgs a;
        a.n_cont = nc;
        a.blk_start[0] = 0;
        for (int k = 0; k < nc; k++) {
            a.cont[k] = conts[base + k];
            a.blk_start[k + 1] = a.blk_start[k] + (int)((conts[base + k].n + threads - 1) / threads);
        }
        if (a.blk_start[nc] == 0) continue;
        dim3 block(threads), grid((unsigned)a.blk_start[nc]);
        if (c.decay)
            hipLaunchKernelGGL(prob3_events_kernel<true>, grid, block, lds, s, c, e, a, max_seg, d_status);
        else
            hipLaunchKernelGGL(prob3_events_kernel<false>, grid, block, lds, s, c, e, a, max_seg, d_status);
        PISA_CHECK_LAUNCH("prob3_events_kernel");
    }
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_prob3_events(const pisa_hip_prob3_params *h_params,
                                   const pisa_hip_earth *h_earth, int64_t nubar,
                                   const double *d_energy, const double *d_coszen, int64_t n,
                                   double *d_probability, int32_t *d_status, void *stream) {
    if (n < 0 || (nubar != 1 && nubar != -1)) return PISA_HIP_ERR_INVALID;
    EarthDev e;
    int rc = make_earth_dev(h_earth, e);
    if (rc) return rc;
    Prob3Consts c;
    if ((rc = make_consts(h_params, c))) return rc;
    if (n == 0) return PISA_HIP_OK;
    if (!d_energy || !d_coszen || !d_probability) return PISA_HIP_ERR_INVALID;
    EvCont ec;
    ec.n = n; ec.energy = d_energy; ec.coszen = d_coszen; ec.prob = d_probability;
    ec.pepmu = nullptr; ec.side = nubar > 0 ? 0 : 1; ec.flav = 0;
    return launch_events(c, e, &ec, 1, d_status, as_stream(stream));
}

PISA_API int pisa_hip_prob3_events_multi(const pisa_hip_prob3_params *h_params,
                                         const pisa_hip_earth *h_earth,
                                         const pisa_hip_event_set *h_sets, int32_t n_sets,
                                         int32_t *d_status, void *stream) {
    if (!h_sets || n_sets < 1 || n_sets > 1024) return PISA_HIP_ERR_INVALID;
    EarthDev e;
    int rc = make_earth_dev(h_earth, e);
    if (rc) return rc;
    Prob3Consts c;
    if ((rc = make_consts(h_params, c))) return rc;
    EvCont *ec = new EvCont[n_sets];
    for (int k = 0; k < n_sets; k++) {
        const pisa_hip_event_set &h = h_sets[k];
        bool bad = h.n_events < 0 || (h.nubar != 1 && h.nubar != -1) || h.flav < 0 || h.flav > 2 ||
                   (h.n_events > 0 && (!h.d_energy || !h.d_coszen || (!h.d_probability && !h.d_pepmu)));
        if (bad) { delete[] ec; return PISA_HIP_ERR_INVALID; }
        ec[k].n = h.n_events; ec[k].energy = h.d_energy; ec[k].coszen = h.d_coszen;
        ec[k].prob = h.d_probability; ec[k].pepmu = reinterpret_cast<double2 *>(h.d_pepmu);
        ec[k].side = h.nubar > 0 ? 0 : 1; ec[k].flav = h.flav;
    }
    rc = launch_events(c, e, ec, n_sets, d_status, as_stream(stream));
    delete[] ec;
    return rc;
}

PISA_API int pisa_hip_fill_probs(const double *d_probability, int64_t init_flav, int64_t flav,
                                 int64_t n, double *d_out, void *stream) {
    if (n < 0 || init_flav < 0 || init_flav > 2 || flav < 0 || flav > 2) return PISA_HIP_ERR_INVALID;
    if (n == 0) return PISA_HIP_OK;
    dim3 block(256), grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(fill_probs_kernel, grid, block, 0, as_stream(stream), d_probability,
                       (int)init_flav, (int)flav, n, d_out);
    PISA_CHECK_LAUNCH("fill_probs_kernel");
    return PISA_HIP_OK;
}

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
    // host copies for re-cutting the items when n_e changes
    int32_t *h_pair_u;
    int items_for_n_e;
};

PISA_API int pisa_hip_grid_plan_destroy(pisa_hip_grid_plan *p) {
    if (!p) return PISA_HIP_OK;
    void *ptrs[] = {p->d_item_u, p->d_item_p0, p->d_item_cnt, p->d_pair_dist, p->d_row_start,
                    p->d_row_cnt, p->d_row_pairs, p->d_rho, p->d_amp};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
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
    if (const char *v = getenv("PISA_HIP_PROB3_CH")) ch = atoi(v) > 0 ? atoi(v) : ch;  // development probe
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
        if (nu == 0) uniq[0] = 0.0;
        size_t npa = np > 0 ? np : 1, nca = nc > 0 ? nc : 1;
        rc = check_hip(hipMalloc(&p->d_pair_dist, npa * 8), "hipMalloc");
        if (!rc) rc = check_hip(hipMalloc(&p->d_row_start, (size_t)n_cz * 4), "hipMalloc");
        if (!rc) rc = check_hip(hipMalloc(&p->d_row_cnt, (size_t)n_cz * 4), "hipMalloc");
        if (!rc) rc = check_hip(hipMalloc(&p->d_row_pairs, nca * 4), "hipMalloc");
        if (!rc) rc = check_hip(hipMalloc(&p->d_rho, (size_t)p->n_unique * 8), "hipMalloc");
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

PISA_API int pisa_hip_prob3_grid_planned(const pisa_hip_prob3_params *h_params,
                                         pisa_hip_grid_plan *plan, const double *d_energy,
                                         int32_t n_e, int32_t e_major, double *d_prob_nu,
                                         double *d_prob_nubar, double *d_pepmu, void *stream) {
    if (!plan || n_e < 1 || !d_energy) return PISA_HIP_ERR_INVALID;
    Prob3Consts c;
    int rc = make_consts(h_params, c);
    if (rc) return rc;
    if (plan->n_e_alloc < n_e) {
        if (plan->d_amp) (void)hipFree(plan->d_amp);
        plan->d_amp = nullptr;
        plan->n_e_alloc = 0;
        size_t amp_bytes = (size_t)2 * (plan->n_pairs > 0 ? plan->n_pairs : 1) * 18 * (((size_t)n_e + 63) / 64 * 64) * sizeof(double);
        PISA_TRY_HIP(hipMalloc(&plan->d_amp, amp_bytes));
        plan->n_e_alloc = n_e;
    }
    if (plan->items_for_n_e != n_e && (rc = cut_items(plan, n_e))) return rc;
    hipStream_t s = as_stream(stream);
    const unsigned tiles = (unsigned)((n_e + 63) / 64);
    dim3 ablock(64), agrid((unsigned)(plan->n_items > 0 ? plan->n_items : 1), 2, tiles);
    static const int groups = []() {
        const char *v = getenv("PISA_HIP_CHAIN_GROUPS");
        int g = v ? atoi(v) : CHAIN_GROUPS_DEFAULT;
        return (g == 1 || g == 2 || g == 4) ? g : CHAIN_GROUPS_DEFAULT;
    }();
    dim3 cblock(64 * groups), cgrid((unsigned)plan->n_cz, 2, tiles);
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
#define CHAIN(G) hipLaunchKernelGGL(prob3_chain_kernel<G>, cgrid, cblock, 0, s, c, (int)n_e, plan->d_row_start, \
                       plan->d_row_cnt, plan->d_row_pairs, plan->n_cz, plan->n_pairs, plan->d_amp,          \
                       (int)e_major, d_prob_nu, d_prob_nubar, (double2 *)d_pepmu)
    if (groups == 1) CHAIN(1); else if (groups == 4) CHAIN(4); else CHAIN(2);
#undef CHAIN
    PISA_CHECK_LAUNCH("prob3_chain_kernel");
    return PISA_HIP_OK;
}
