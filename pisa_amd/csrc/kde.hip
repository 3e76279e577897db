// kde.hip -- Gaussian kernel density estimator of the KDE smoothing stage
// (pisa/stages/utils/kde.py -> pisa/utils/kde_hist.py:35-217 -> external
// `kde.gaussian_kde`, whose source is NOT in the reference tree: parity of the
// KDE core is UNPINNED).
//
// The estimator (call contract of kde_hist.py:110-120):
//   f(q)  = sum_i coef_i * exp(-0.5 * s2_i * (q - x_i)^T inv_cov (q - x_i))
//   pilot:  coef_i = w_i / norm, s2_i = 1            (fixed bandwidth, evaluated at the x_i)
//   final:  lam_i = (pilot_i / g)^alpha, g = geometric mean of the pilot densities,
//           coef_i = w_i lam_i^d / norm, s2_i = lam_i^2
//
// Two families of entry points:
//
// * `pisa_hip_kde_eval`        all pairs, no cut-off: O(N*M).  The plain double loop, kept as the
//                              exact form (small problems, and the GPU-side cross-check of the
//                              cut-off form).
// * `pisa_hip_kde_create/evaluate/destroy`   the estimator object.  Everything is done in
//   WHITENED coordinates y = U (x - mean), U^T U = inv_cov, where the quadratic form is the
//   plain Euclidean |y_q - y_i|^2, and with a CELL LIST: the sources are radix-sorted
//   (hipCUB) into a uniform grid of cells of side r_cut / 8, the query points into tiles of
//   cells; a workgroup owns up to 512 queries of one tile and streams through LDS only the
//   cells that can hold a pair with kernel value above the cut-off:
//       pair dropped  =>  exp(-0.5 s2_i r^2) < tol     (r_cut^2 = 2 ln(1/tol))
//   so the truncation error of every density value is below tol * sum_i coef_i.  The cell
//   test uses the smallest s2 (widest kernel) of the cell, so variable bandwidths keep the
//   guarantee.  Cost O(N * k), k = sources within the cut-off (7-20 % of N for the
//   Silverman bandwidth at N = 10^5..10^7 in 2-D).  Fixed summation order (sorted cells,
//   sorted sources, split partial sums added in split order) => bit-reproducible.
//   `exp` is an own routine for non-positive arguments (one rint, two-term Cody-Waite
//   reduction, degree-13 polynomial, ldexp: 19 instructions instead of the library's ~35).
// Roofline: FP64 VALU, 23 instructions per pair in 2-D (compute bound; 32 B of LDS per pair
// and pair of queries).
//
// On top of the cell list, in 2-D with a cut-off (the KDE stage's case):
// * the PILOT is a fast Gauss transform: Hermite series of every non-empty cell (`kde_hermite_coef_kernel`),
//   translated into one local expansion per target cell in two separable passes over the cell grid
//   (`kde_h2l4_kernel<P, 0 / 1>`), evaluated with P^2 multiply-adds per target (`kde_local_pilot_kernel`);
// * a map is evaluated on the LATTICE of its points (`pisa_hip_kde_evaluate_lattice`, `kde_lattice_kernel`): along a
//   lattice line the kernel values of a source follow g_{c+-k} = g_c r^k Q_k, one multiply and one multiply-add each;
// * the estimators of one evaluation of the stage run as one job list on the library's own threads and streams
//   (`kde_batch.hip`: `pisa_hip_kde_lattice_submit / _wait`).
// History and the measurements behind every choice: EXPERIMENTS.md (section 4.4, R3-8).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include <hipcub/hipcub.hpp>
#include <rocprim/rocprim.hpp>

#include "common.hpp"

namespace pisa {

// ------------------------------------------------------------------ all pairs (exact) form
constexpr int KDE_TILE = 1024;
constexpr int KDE_THREADS = 256;

template <int D>
__global__ void __launch_bounds__(KDE_THREADS)
kde_eval_kernel(const double *__restrict__ src, const double *__restrict__ coef,
                const double *__restrict__ s2, int64_t n_src, const double *__restrict__ qry,
                int64_t n_qry, double ic00, double ic01, double ic02, double ic11, double ic12,
                double ic22, int64_t src_chunk, double *__restrict__ out) {
    __shared__ double t_x[D][KDE_TILE];
    __shared__ double t_c[KDE_TILE];
    __shared__ double t_s[KDE_TILE];
    const int64_t j = (int64_t)blockIdx.x * KDE_THREADS + threadIdx.x;
    double q[D];
#pragma unroll
    for (int d = 0; d < D; d++) q[d] = j < n_qry ? qry[(int64_t)d * n_qry + j] : 0.0;
    double acc = 0.0;
    // this workgroup's share of the sources; its sums go to row blockIdx.y of `out`
    const int64_t src_lo = (int64_t)blockIdx.y * src_chunk;
    const int64_t src_hi = src_lo + src_chunk < n_src ? src_lo + src_chunk : n_src;
    out += (int64_t)blockIdx.y * n_qry;
    for (int64_t base = src_lo; base < src_hi; base += KDE_TILE) {
        const int cnt = (int)((src_hi - base) < KDE_TILE ? (src_hi - base) : KDE_TILE);
        __syncthreads();
        for (int k = threadIdx.x; k < cnt; k += KDE_THREADS) {
#pragma unroll
            for (int d = 0; d < D; d++) t_x[d][k] = src[(int64_t)d * n_src + base + k];
            t_c[k] = coef[base + k];
            t_s[k] = s2[base + k];
        }
        __syncthreads();
        for (int k = 0; k < cnt; k++) {
            double d0 = q[0] - t_x[0][k];
            double r2 = ic00 * d0 * d0;
            if (D > 1) {
                double d1 = q[1] - t_x[D > 1 ? 1 : 0][k];
                r2 += 2.0 * ic01 * d0 * d1 + ic11 * d1 * d1;
                if (D > 2) {
                    double d2 = q[2] - t_x[D > 2 ? 2 : 0][k];
                    r2 += 2.0 * ic02 * d0 * d2 + 2.0 * ic12 * d1 * d2 + ic22 * d2 * d2;
                }
            }
            acc += t_c[k] * exp(-0.5 * t_s[k] * r2);
        }
    }
    if (j < n_qry) out[j] = acc;
}

// out[j] = partial[0][j] + partial[1][j] + ... in split order
__global__ void __launch_bounds__(256)
kde_reduce_kernel(const double *__restrict__ partial, int n_split, int64_t n_qry,
                  double *__restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_qry) return;
    double acc = partial[j];
    for (int s = 1; s < n_split; s++) acc += partial[(int64_t)s * n_qry + j];
    out[j] = acc;
}

// scratch for the per-split partial sums and the expansion coefficients (grown on demand).  One
// per host thread: estimators may be built and evaluated from several host threads at once, each
// on its own stream (the utils.kde stage does), and a thread's calls are ordered on its stream.
static thread_local double *g_kde_scratch = nullptr;
static thread_local size_t g_kde_scratch_bytes = 0;
static thread_local hipStream_t g_kde_scratch_stream = nullptr;
static thread_local bool g_kde_scratch_used = false;

// The scratch is keyed by host thread, not by stream: a thread that moves to another stream waits for
// the kernels of the previous one, which may still be reading it.  Returns the buffer (>= `bytes`).
static int kde_scratch(size_t bytes, hipStream_t s, double **out) {
    if (g_kde_scratch_used && g_kde_scratch_stream != s) {
        int rc = check_hip(hipStreamSynchronize(g_kde_scratch_stream), "hipStreamSynchronize");
        if (rc) return rc;
    }
    if (bytes > g_kde_scratch_bytes) {
        if (g_kde_scratch) {
            // kernels of this thread's stream may still use the old buffer
            int rc = check_hip(hipStreamSynchronize(s), "hipStreamSynchronize");
            if (rc) return rc;
            (void)hipFree(g_kde_scratch);
        }
        g_kde_scratch = nullptr;
        g_kde_scratch_bytes = 0;
        int rc = check_hip(hipMalloc(&g_kde_scratch, bytes), "hipMalloc");
        if (rc) return rc;
        g_kde_scratch_bytes = bytes;
    }
    g_kde_scratch_stream = s;
    g_kde_scratch_used = true;
    *out = g_kde_scratch;
    return PISA_HIP_OK;
}

// =================================================================== estimator object
constexpr int RED_BLOCKS = 256;   // fixed reduction geometry => fixed summation order
constexpr int RED_THREADS = 256;
constexpr int Q_PER_THREAD = 2;
constexpr int Q_CHUNK = KDE_THREADS * Q_PER_THREAD;   // queries per workgroup
constexpr int SRC_TILE = 512;                          // sources per LDS tile
constexpr int64_t KEY_OFF = 1 << 20;                   // tile coordinates are stored + 2^20
constexpr int MAX_CELLS = 1 << 22;
// the cell grid never has more than max(4096, 4 n) cells (larger cells beyond that: less pruning,
// same results), so that the workspace scales with the number of sources
static inline int64_t cells_cap(int64_t n) { return std::min<int64_t>(MAX_CELLS, std::max<int64_t>(4096, 4 * n)); }

struct KdeGeom {
    int32_t dim;
    int32_t nc[3];       // cells per dimension (1 for unused dimensions)
    double ylo[3];       // lower corner of the cell grid in whitened coordinates
    double cell, inv_cell;
    double rcut2;        // 2 ln(1/tol); <= 0: no cut-off
    double U[9];         // whitening: y = U (x - mean), upper triangular, row-major 3x3
    double mean[3];
};

struct KdeBlock {        // one workgroup of the pair kernel
    int32_t q_begin, q_count;
    int32_t c0[3], c1[3];   // cell bounds of the queries' tile (inclusive; may lie outside the grid)
    int32_t head;           // index of the tile among the non-empty tiles (sorted order)
};

__device__ inline double block_sum(double v, double *lds) {
    const int t = threadIdx.x;
    __syncthreads();
    lds[t] = v;
    __syncthreads();
    for (int s = RED_THREADS / 2; s > 0; s >>= 1) {
        if (t < s) lds[t] += lds[t + s];
        __syncthreads();
    }
    return lds[0];
}
__device__ inline double block_min(double v, double *lds) {
    const int t = threadIdx.x;
    __syncthreads();
    lds[t] = v;
    __syncthreads();
    for (int s = RED_THREADS / 2; s > 0; s >>= 1) {
        if (t < s) lds[t] = fmin(lds[t], lds[t + s]);
        __syncthreads();
    }
    return lds[0];
}

// pass 1: sum w, sum w x_d, min x_d, max x_d   -> partial[block][1 + 3 D]
template <int D>
__global__ void __launch_bounds__(RED_THREADS)
kde_moments1_kernel(const double *__restrict__ x, const double *__restrict__ w, int64_t n,
                    double *__restrict__ partial) {
    __shared__ double lds[RED_THREADS];
    double sw = 0.0, sx[D], mn[D], mx[D];
#pragma unroll
    for (int d = 0; d < D; d++) { sx[d] = 0.0; mn[d] = INFINITY; mx[d] = -INFINITY; }
    for (int64_t i = (int64_t)blockIdx.x * RED_THREADS + threadIdx.x; i < n;
         i += (int64_t)RED_BLOCKS * RED_THREADS) {
        const double wi = w ? w[i] : 1.0;
        sw += wi;
#pragma unroll
        for (int d = 0; d < D; d++) {
            const double v = x[(int64_t)d * n + i];
            sx[d] += wi * v;
            mn[d] = fmin(mn[d], v);
            mx[d] = fmax(mx[d], v);
        }
    }
    double *out = partial + (int64_t)blockIdx.x * (1 + 3 * D);
    double r = block_sum(sw, lds);
    if (threadIdx.x == 0) out[0] = r;
#pragma unroll
    for (int d = 0; d < D; d++) {
        r = block_sum(sx[d], lds);
        if (threadIdx.x == 0) out[1 + d] = r;
        r = block_min(mn[d], lds);
        if (threadIdx.x == 0) out[1 + D + d] = r;
        r = -block_min(-mx[d], lds);
        if (threadIdx.x == 0) out[1 + 2 * D + d] = r;
    }
}

// pass 2 (about the weighted mean): sum w^2, sum w xc_d xc_e (d <= e) -> partial[block][1 + 6]
template <int D>
__global__ void __launch_bounds__(RED_THREADS)
kde_moments2_kernel(const double *__restrict__ x, const double *__restrict__ w, int64_t n,
                    double m0, double m1, double m2, double *__restrict__ partial) {
    __shared__ double lds[RED_THREADS];
    const double mean[3] = {m0, m1, m2};
    double sww = 0.0, c[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * RED_THREADS + threadIdx.x; i < n;
         i += (int64_t)RED_BLOCKS * RED_THREADS) {
        const double wi = w ? w[i] : 1.0;
        sww += wi * wi;
        double xc[3] = {0, 0, 0};
#pragma unroll
        for (int d = 0; d < D; d++) xc[d] = x[(int64_t)d * n + i] - mean[d];
        int k = 0;
#pragma unroll
        for (int d = 0; d < D; d++)
#pragma unroll
            for (int e = d; e < D; e++) c[k++] += wi * xc[d] * xc[e];
    }
    double *out = partial + (int64_t)blockIdx.x * 7;
    double r = block_sum(sww, lds);
    if (threadIdx.x == 0) out[0] = r;
    for (int k = 0; k < 6; k++) {
        r = block_sum(c[k], lds);
        if (threadIdx.x == 0) out[1 + k] = r;
    }
}

// columns [0, n_sum) of partial[RED_BLOCKS][width] are summed, [n_sum, n_sum+n_min) minimised,
// the rest maximised, in block order
__global__ void __launch_bounds__(RED_THREADS)
kde_final_reduce_kernel(const double *__restrict__ partial, int width, int n_sum, int n_min,
                        double *__restrict__ out) {
    __shared__ double lds[RED_THREADS];
    for (int k = 0; k < width; k++) {
        const double v = partial[(int64_t)threadIdx.x * width + k];
        double r;
        if (k < n_sum) r = block_sum(v, lds);
        else if (k < n_sum + n_min) r = block_min(v, lds);
        else r = -block_min(-v, lds);
        if (threadIdx.x == 0) out[k] = r;
    }
}

// kde_final_reduce_kernel on the host: the same pairing (element t joined with element t + s, s = 128 .. 1)
static void final_reduce_host(const double *partial, int width, int n_sum, int n_min, double *out) {
    static_assert(RED_BLOCKS == RED_THREADS, "one partial per thread of the device form");
    double v[RED_THREADS];
    for (int k = 0; k < width; k++) {
        for (int t = 0; t < RED_THREADS; t++) v[t] = partial[(size_t)t * width + k];
        const bool is_max = k >= n_sum + n_min;
        if (is_max) for (int t = 0; t < RED_THREADS; t++) v[t] = -v[t];
        for (int st = RED_THREADS / 2; st > 0; st >>= 1)
            for (int t = 0; t < st; t++) v[t] = k < n_sum ? v[t] + v[t + st] : fmin(v[t], v[t + st]);
        out[k] = is_max ? -v[0] : v[0];
    }
}

__device__ inline uint64_t tile_key(const double *y, const KdeGeom &g, int tile) {
    uint64_t key = 0;
#pragma unroll
    for (int d = 2; d >= 0; d--) {
        int64_t c = 0;
        if (d < g.dim) {
            double f = floor((y[d] - g.ylo[d]) * g.inv_cell);
            // NaN or far outside: parked in the outermost tile (contributes / receives nothing)
            if (!(f > -(double)(KEY_OFF - 2) * tile)) f = -(double)(KEY_OFF - 2) * tile;
            if (f > (double)(KEY_OFF - 2) * tile) f = (double)(KEY_OFF - 2) * tile;
            c = (int64_t)f;
            c = (c >= 0 ? c / tile : -((-c + tile - 1) / tile));   // floor division
        }
        key = (key << 21) | (uint64_t)(c + KEY_OFF);
    }
    return key;
}

// whitened coordinates + tile key of every point; `clamp`: sources are put inside the cell grid
template <int D>
__global__ void __launch_bounds__(256)
kde_whiten_key_kernel(const double *__restrict__ x, int64_t n, KdeGeom g, int tile, int clamp,
                      double *__restrict__ y, uint64_t *__restrict__ keys, uint32_t *__restrict__ idx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double xc[3] = {0, 0, 0}, yy[3] = {0, 0, 0};
#pragma unroll
    for (int d = 0; d < D; d++) xc[d] = x[(int64_t)d * n + i] - g.mean[d];
#pragma unroll
    for (int d = 0; d < D; d++) {
        double a = 0.0;
#pragma unroll
        for (int e = d; e < D; e++) a += g.U[d * 3 + e] * xc[e];
        yy[d] = a;
        if (clamp) {   // rounding may put a point a hair outside the transformed bounding box
            const double lo = g.ylo[d], hi = g.ylo[d] + (g.nc[d] - 0.5) * g.cell;
            const double v = fmin(fmax(a, lo), hi);
            yy[d] = (a == a) ? v : a;
        }
        y[(int64_t)d * n + i] = a;
    }
    keys[i] = tile_key(yy, g, tile);
    idx[i] = (uint32_t)i;
}

#ifdef PISA_DEV_PROBES
__global__ void kde_noop_kernel(const double *p) { (void)p; }
#endif

// sorted copies: ys[d][k] = y[d][perm[k]], ws[k] = w[perm[k]] * scale
template <int D>
__global__ void __launch_bounds__(256)
kde_gather_kernel(const double *__restrict__ y, const double *__restrict__ w, double scale,
                  const uint32_t *__restrict__ perm, int64_t n, double *__restrict__ ys,
                  double *__restrict__ ws) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const uint32_t i = perm[k];
#pragma unroll
    for (int d = 0; d < D; d++) ys[(int64_t)d * n + k] = y[(int64_t)d * n + i];
    if (ws) ws[k] = (w ? w[i] : 1.0) * scale;
}

__device__ inline int64_t flat_cell(uint64_t key, const KdeGeom &g) {
    const int64_t cx = (int64_t)(key & 0x1FFFFF) - KEY_OFF;
    const int64_t cy = (int64_t)((key >> 21) & 0x1FFFFF) - KEY_OFF;
    const int64_t cz = (int64_t)((key >> 42) & 0x1FFFFF) - KEY_OFF;
    return (cz * g.nc[1] + cy) * g.nc[0] + cx;
}

// The sources are sorted by their FLAT cell index (a 32-bit key of ceil(log2 n_cells) significant bits: two or three
// 8-bit radix passes at C3 sizes) instead of the 63-bit tile key (a 64-bit merge sort below 2^20 items): both
// orders are the same (cz, cy, cx lexicographic, stable), so the sorted order is what it was.
// whitened coordinates + flat cell index of every source (put inside the cell grid like kde_whiten_key_kernel's clamp)
template <int D>
__global__ void __launch_bounds__(256)
kde_whiten_flat_kernel(const double *__restrict__ x, const double *__restrict__ w, int64_t n, KdeGeom g,
                       double *__restrict__ rec, uint32_t *__restrict__ flat, uint32_t *__restrict__ idx, int pack) {
    // rec[i] = (y_0, y_1, y_2, weight): ONE 32-byte record per source, so that the gather into cell order touches one
    // sector per source instead of one per array (three random 8-byte reads each pulled their own: 27 us at 5.8e5 sources)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double xc[3] = {0, 0, 0}, yy[3] = {0, 0, 0}, out[4] = {0, 0, 0, w ? w[i] : 1.0};
#pragma unroll
    for (int d = 0; d < D; d++) xc[d] = x[(int64_t)d * n + i] - g.mean[d];
#pragma unroll
    for (int d = 0; d < D; d++) {
        double a = 0.0;
#pragma unroll
        for (int e = d; e < D; e++) a += g.U[d * 3 + e] * xc[e];
        const double lo = g.ylo[d], hi = g.ylo[d] + (g.nc[d] - 0.5) * g.cell;
        const double v = fmin(fmax(a, lo), hi);   // rounding may put a point a hair outside the transformed bounding box
        yy[d] = (a == a) ? v : a;
        out[d] = a;
    }
    typedef double __attribute__((ext_vector_type(2))) d2;
    d2 *dst = reinterpret_cast<d2 *>(rec + 4 * i);
    dst[0] = (d2){out[0], out[1]};
    dst[1] = (d2){out[2], out[3]};
    // pack > 0: (cell << pack) | index in ONE 32-bit word (the index fits below the cell's bits): the sort then moves keys only
    const uint32_t cell_i = (uint32_t)flat_cell(tile_key(yy, g, 1), g);
    flat[i] = pack ? (cell_i << pack) | (uint32_t)i : cell_i;
    if (!pack) idx[i] = (uint32_t)i;
}

// cell_start[c] = first sorted source of cell c, from the sorted flat cell indices
__global__ void __launch_bounds__(256)
kde_cell_start_flat_kernel(const uint32_t *__restrict__ flat, int64_t n, int64_t n_cells,
                           int32_t *__restrict__ cell_start, int pack) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k > n) return;
    const int64_t prev = k == 0 ? -1 : (int64_t)(flat[k - 1] >> pack);
    const int64_t cur = k == n ? n_cells : (int64_t)(flat[k] >> pack);
    for (int64_t c = prev + 1; c <= cur; c++) cell_start[c] = (int32_t)k;
}

// sorted copies of the sources (from the records of kde_whiten_flat_kernel): ys[d][k] = y[d][perm[k]], wn[k] = w[perm[k]] / sum w, and the fixed-bandwidth
// coefficients coef[k] = wn[k] / norm (s2[k] = 1 if given)
template <int D>
__global__ void __launch_bounds__(256)
kde_gather_sources_kernel(const double *__restrict__ rec, double scale, double inv_norm,
                          const uint32_t *__restrict__ perm, uint32_t perm_mask, int64_t n, double *__restrict__ ys,
                          double *__restrict__ wn, double *__restrict__ coef, double *__restrict__ s2) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const uint32_t i = perm[k] & perm_mask;            // (packed keys: the index is the low part of the sorted word)
    typedef double __attribute__((ext_vector_type(2))) d2;
    const d2 *src = reinterpret_cast<const d2 *>(rec + 4 * (int64_t)i);
    const d2 r01 = src[0], r23 = src[1];
    const double r[4] = {r01.x, r01.y, r23.x, r23.y};
#pragma unroll
    for (int d = 0; d < D; d++) ys[(int64_t)d * n + k] = r[d];
    const double v = r[3] * scale;
    wn[k] = v;
    coef[k] = v * inv_norm;
    if (s2) s2[k] = 1.0;
}

// heads of runs of equal key
__global__ void __launch_bounds__(256)
kde_heads_kernel(const uint64_t *__restrict__ keys, int64_t n, uint8_t *__restrict__ flags) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    flags[k] = (k == 0 || keys[k] != keys[k - 1]) ? 1 : 0;
}
__global__ void __launch_bounds__(256)
kde_head_keys_kernel(const uint64_t *__restrict__ keys, const int32_t *__restrict__ starts,
                     const int32_t *__restrict__ n_heads, uint64_t *__restrict__ out) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < *n_heads) out[k] = keys[starts[k]];
}

// exp(t) for t <= 0: |rel. error| < 2 ulp down to the subnormal range (where ldexp rounds)
__device__ inline double exp_nonpos(double t) {
    t = fmax(t, -800.0);
    const double k = __builtin_rint(t * 1.4426950408889634074);
    double r = __builtin_fma(k, -6.93147180369123816490e-01, t);
    r = __builtin_fma(k, -1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;                       // 1/13!
    p = __builtin_fma(p, r, 2.08767569878680989792e-09);     // 1/12!
    p = __builtin_fma(p, r, 2.50521083854417187751e-08);     // 1/11!
    p = __builtin_fma(p, r, 2.75573192239858906526e-07);     // 1/10!
    p = __builtin_fma(p, r, 2.75573192239858906526e-06);     // 1/9!
    p = __builtin_fma(p, r, 2.48015873015873015873e-05);     // 1/8!
    p = __builtin_fma(p, r, 1.98412698412698412698e-04);     // 1/7!
    p = __builtin_fma(p, r, 1.38888888888888888889e-03);     // 1/6!
    p = __builtin_fma(p, r, 8.33333333333333333333e-03);     // 1/5!
    p = __builtin_fma(p, r, 4.16666666666666666667e-02);     // 1/4!
    p = __builtin_fma(p, r, 1.66666666666666666667e-01);     // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)k);
}

// The pair kernel.  Workgroup = one KdeBlock (<= 256 QPT queries of one tile, QPT per thread) x
// one share `blockIdx.y` of the visited cells.  Sources, coef and s2h = -0.5 s2 come sorted by
// cell; VAR_BW = false: s2 = 1 for every source (pilot estimate).
template <int D, bool VAR_BW, int QPT>
__global__ void __launch_bounds__(KDE_THREADS)
kde_pairs_kernel(KdeGeom g, const KdeBlock *__restrict__ blocks, const double *__restrict__ qy,
                 int64_t n_qry, const double *__restrict__ sy, int64_t n_src,
                 const double *__restrict__ coef, const double *__restrict__ s2,
                 const int32_t *__restrict__ cell_start, const double *__restrict__ cell_s2min,
                 const double *__restrict__ s2min_glob, int n_split, double *__restrict__ out,
                 unsigned long long *__restrict__ pair_count) {
    constexpr int W = (D == 3) ? 6 : 4;   // doubles per staged source: y[D], coef, s2h (+ pad)
    __shared__ double t_src[SRC_TILE * W];
    const KdeBlock b = blocks[blockIdx.x];
    const int split = blockIdx.y;
    double q[QPT][D], acc[QPT];
#pragma unroll
    for (int u = 0; u < QPT; u++) {
        const int jq = threadIdx.x + u * KDE_THREADS;
        const int64_t j = b.q_begin + (jq < b.q_count ? jq : 0);
#pragma unroll
        for (int d = 0; d < D; d++) q[u][d] = qy[(int64_t)d * n_qry + j];
        acc[u] = 0.0;
    }
    // candidate cells: everything within the widest kernel's reach of the tile
    const double s2min = VAR_BW ? *s2min_glob : 1.0;
    const bool cut = g.rcut2 > 0.0;
    int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
#pragma unroll
    for (int d = 0; d < D; d++) {
        lo[d] = 0;
        hi[d] = g.nc[d] - 1;
        if (cut) {
            const double reach = ceil(sqrt(g.rcut2 / s2min) * g.inv_cell);
            const double l = (double)b.c0[d] - reach, h = (double)b.c1[d] + reach;
            lo[d] = l > 0.0 ? (int)l : 0;
            hi[d] = h < (double)(g.nc[d] - 1) ? (int)h : g.nc[d] - 1;
        }
    }
    unsigned long long pairs = 0;
    int visited = 0;
    for (int cz = lo[2]; cz <= hi[2]; cz++) {
        double gz = 0.0;
        if (D > 2) {
            const int gap = cz < b.c0[2] ? b.c0[2] - cz - 1 : (cz > b.c1[2] ? cz - b.c1[2] - 1 : 0);
            gz = gap * g.cell;
        }
        for (int cy = lo[1]; cy <= hi[1]; cy++) {
            double gy = 0.0;
            if (D > 1) {
                const int gap = cy < b.c0[1] ? b.c0[1] - cy - 1 : (cy > b.c1[1] ? cy - b.c1[1] - 1 : 0);
                gy = gap * g.cell;
            }
            const int64_t row = ((int64_t)cz * g.nc[1] + cy) * g.nc[0];
            for (int cx = lo[0]; cx <= hi[0]; cx++) {
                const int64_t c = row + cx;
                const int begin = cell_start[c], end = cell_start[c + 1];
                if (end == begin) continue;
                if (cut) {
                    const int gap = cx < b.c0[0] ? b.c0[0] - cx - 1 : (cx > b.c1[0] ? cx - b.c1[0] - 1 : 0);
                    const double gx = gap * g.cell;
                    const double d2 = gx * gx + gy * gy + gz * gz;
                    if (d2 * (VAR_BW ? cell_s2min[c] : 1.0) > g.rcut2) continue;
                }
                if ((visited++) % n_split != split) continue;
                pairs += (unsigned long long)(end - begin) * b.q_count;
                for (int base = begin; base < end; base += SRC_TILE) {
                    const int cnt = end - base < SRC_TILE ? end - base : SRC_TILE;
                    __syncthreads();
                    for (int k = threadIdx.x; k < cnt; k += KDE_THREADS) {
#pragma unroll
                        for (int d = 0; d < D; d++) t_src[k * W + d] = sy[(int64_t)d * n_src + base + k];
                        t_src[k * W + D] = coef[base + k];
                        t_src[k * W + D + 1] = VAR_BW ? -0.5 * s2[base + k] : -0.5;
                    }
                    __syncthreads();
#pragma unroll 2
                    for (int k = 0; k < cnt; k++) {
                        const double *s = t_src + k * W;
                        const double cf = s[D], sh = s[D + 1];
#pragma unroll
                        for (int u = 0; u < QPT; u++) {
                            double d0 = q[u][0] - s[0];
                            double r2 = d0 * d0;
                            if (D > 1) { const double d1 = q[u][D > 1 ? 1 : 0] - s[D > 1 ? 1 : 0]; r2 = __builtin_fma(d1, d1, r2); }
                            if (D > 2) { const double d2 = q[u][D > 2 ? 2 : 0] - s[D > 2 ? 2 : 0]; r2 = __builtin_fma(d2, d2, r2); }
                            acc[u] = __builtin_fma(cf, exp_nonpos(r2 * sh), acc[u]);
                        }
                    }
                }
            }
        }
    }
    out += (int64_t)split * n_qry;
#pragma unroll
    for (int u = 0; u < QPT; u++) {
        const int jq = threadIdx.x + u * KDE_THREADS;
        if (jq < b.q_count) out[b.q_begin + jq] = acc[u];
    }
    if (pair_count && threadIdx.x == 0 && pairs) atomicAdd(pair_count, pairs);
}

// out[dst] = sum over splits (split order); dst = perm[k] (scatter back to the caller's order) or k
__global__ void __launch_bounds__(256)
kde_combine_kernel(const double *__restrict__ partial, int n_split, int64_t n,
                   const uint32_t *__restrict__ perm, double *__restrict__ out) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    double acc = partial[k];
    for (int s = 1; s < n_split; s++) acc += partial[(int64_t)s * n + k];
    out[perm ? perm[k] : k] = acc;
}

// ------------------------------------------------------------------ evaluation on a lattice
// A map's evaluation points are a tensor grid with uniform steps (oversampled bin centres,
// kde_hist.py:122-190).  In whitened coordinates (U upper triangular) lattice index 0 moves y_a
// only, by da = U00 step0, and index 1 moves (y_a, y_b) by (sa, db) = (U01, U11) step1: along a
// line of constant index 1 the kernel values of one source are a Gaussian sampled at equal steps,
//   g_{k+1} = g_k r_k,  r_{k+1} = r_k q,  q = exp(-s2 da^2),
// two multiplications per point instead of an exponential.  A thread owns a strip of R
// consecutive points of one line and starts at the strip's middle point (three exponentials per
// source and strip: the middle value and the first ratio in either direction).  The host admits
// R only if R da sqrt(max s2) <= 50: the middle point of a strip within reach of a source is then
// at most (sqrt(rcut2) + 25) kernel widths away, its value >= exp(-600), and no ratio leaves the
// double range.  Rounding: g_k carries the error of k^2 / 2 multiplications and of exponents up
// to a few hundred, < 3e-13 relative for R = 32.  A strip is skipped for a source only if every
// one of its points is beyond the cut-off, so the stated tolerance holds.  Fixed order (sources
// in sorted order inside a share, shares added in order): bit-reproducible.
struct KdeLattice {
    double ya0, yb0;   // whitened coordinates of lattice point (0, 0)
    double da;         // y_a step of index 0 (> 0)
    double sa, db;     // (y_a, y_b) step of index 1
    int32_t n0, n1, strips_a;   // strips_a = ceil(n0 / R)
    int32_t sw, lpw, n_colblk;  // a wavefront's sub-patch: sw strips of lpw consecutive lines (sw lpw = LG lanes, 64 / LG lane groups); column blocks per line
};

// up = h exp(t), dn = h exp(-t), |t| <= 700: one range reduction and the even / odd halves of the same
// degree-13 polynomial serve both (exp(+-r) = C(r^2) +- r S(r^2))
__device__ inline void exp_pair(double t, double h, double &up, double &dn) {
    t = fmin(fmax(t, -700.0), 700.0);
    const double k = __builtin_rint(t * 1.4426950408889634074);
    double r = __builtin_fma(k, -6.93147180369123816490e-01, t);
    r = __builtin_fma(k, -1.90821492927058770002e-10, r);
    const double r2 = r * r;
    double ce = 2.08767569878680989792e-09;                     // 1/12!
    double co = 1.6059043836821613e-10;                         // 1/13!
    ce = __builtin_fma(ce, r2, 2.75573192239858906526e-07);     // 1/10!
    co = __builtin_fma(co, r2, 2.50521083854417187751e-08);     // 1/11!
    ce = __builtin_fma(ce, r2, 2.48015873015873015873e-05);     // 1/8!
    co = __builtin_fma(co, r2, 2.75573192239858906526e-06);     // 1/9!
    ce = __builtin_fma(ce, r2, 1.38888888888888888889e-03);     // 1/6!
    co = __builtin_fma(co, r2, 1.98412698412698412698e-04);     // 1/7!
    ce = __builtin_fma(ce, r2, 4.16666666666666666667e-02);     // 1/4!
    co = __builtin_fma(co, r2, 8.33333333333333333333e-03);     // 1/5!
    ce = __builtin_fma(ce, r2, 0.5);
    co = __builtin_fma(co, r2, 1.66666666666666666667e-01);     // 1/3!
    ce = __builtin_fma(ce, r2, 1.0);
    co = __builtin_fma(co, r2, 1.0);
    const double so = r * co;
    const int ki = (int)k;
    up = __builtin_ldexp(ce + so, ki) * h;
    dn = __builtin_ldexp(ce - so, -ki) * h;
}

// Source record as the lattice kernel uses it in LDS (192 B): [0] y_a  [1] y_b  [2] coef  [3] -s2 / 2  [4] -s2 da
// [5] h = exp(-s2 da^2 / 2)  [8 + k - 2] Q_k = exp(-s2 da^2 k (k - 1) / 2) = h^(k (k - 1)), k = 2 .. 16.
// With the strip's middle value g_c and the ratios r_up = g_{c+1} / g_c, r_dn = g_{c-1} / g_c the
// values along the line are  g_{c +- k} = g_c r^k Q_k : one multiplication (p <- p r) and one
// multiply-add (acc += p Q_k, Q_k a scalar operand) per point.
// In GLOBAL memory a record is its first eight numbers only (LAT_GREC, 64 B; round 6): the Q table is rebuilt in LDS when a
// piece is staged, Q_k = Q_{k-1} h^(2 (k - 1)) -- two multiplications per entry by one lane per record.  A record is fetched
// ~11 times per launch (once per sub-patch and lane group within its reach): 0.9 GB per estimator at 192 B, a quarter of it now.
constexpr int LAT_GREC = 8;
constexpr int LAT_REC = 24;
constexpr int LAT_Q0 = 8;
constexpr int LAT_QMAX = 16;
constexpr int LAT_SHARE = 64;   // sources per share (= the workgroup of kde_lattice_prep_kernel, which writes the share's box)
__global__ void __launch_bounds__(64)
kde_lattice_prep_kernel(const double *__restrict__ ys, const double *__restrict__ coef,
                        const double *__restrict__ s2, int64_t n, double da, double rcut2, double *__restrict__ rec,
                        double *__restrict__ box) {
    // one source per thread; the 64 records of a wavefront go through LDS so that they leave as 16-byte
    // stores of consecutive lanes (4 KB contiguous per wavefront)
    typedef double __attribute__((ext_vector_type(2))) d2;
    __shared__ __attribute__((aligned(16))) double stage[64 * LAT_GREC];
    const int lane = threadIdx.x;
    const int64_t k0 = (int64_t)blockIdx.x * 64, k = k0 + lane;
    if (k < n) {
        const double v = s2[k];
        double *r = stage + lane * LAT_GREC;
        r[0] = ys[k];
        r[1] = ys[n + k];
        r[2] = coef[k];
        r[3] = -0.5 * v;
        r[4] = -v * da;
        r[5] = exp_nonpos(-0.5 * v * da * da);
        r[6] = 0.0;
        r[7] = 0.0;
    }
    // the share's (y_a, y_b) box: outside it no lattice point is within the cut-off of any of the 64 sources
    // (sources are sorted by cell row, then cell column: a share is compact)
    {
        double alo = INFINITY, ahi = -INFINITY, blo = INFINITY, bhi = -INFINITY;
        if (k < n) {
            const double ya = ys[k], yb = ys[n + k];
            const double reach = sqrt(rcut2 / s2[k]);
            alo = ya - reach; ahi = ya + reach; blo = yb - reach; bhi = yb + reach;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            alo = fmin(alo, __shfl_down(alo, off));
            ahi = fmax(ahi, __shfl_down(ahi, off));
            blo = fmin(blo, __shfl_down(blo, off));
            bhi = fmax(bhi, __shfl_down(bhi, off));
        }
        if (lane == 0) {
            box[4 * blockIdx.x] = alo;
            box[4 * blockIdx.x + 1] = ahi;
            box[4 * blockIdx.x + 2] = blo;
            box[4 * blockIdx.x + 3] = bhi;
        }
    }
    __syncthreads();
    const int64_t n_here = n - k0 < 64 ? n - k0 : 64;
    const int n_d2 = (int)n_here * (LAT_GREC / 2);
    const d2 *src = reinterpret_cast<const d2 *>(stage);
    d2 *dst = reinterpret_cast<d2 *>(rec + k0 * LAT_GREC);
    for (int i = lane; i < n_d2; i += 64) dst[i] = src[i];
}

// bounding box of patch p in (y_a, y_b)
__device__ inline void lattice_patch_box(const KdeLattice &L, int R, int p, double &pa_lo, double &pa_hi,
                                         double &pb_lo, double &pb_hi) {
    const int cb = p % L.n_colblk, rb = p / L.n_colblk;
    const int t_f = cb * L.sw, j_f = rb * L.lpw;
    const int t_l = (t_f + L.sw < L.strips_a ? t_f + L.sw : L.strips_a) - 1;
    const int j_l = (j_f + L.lpw < L.n1 ? j_f + L.lpw : L.n1) - 1;
    const double pb0 = L.yb0 + j_f * L.db, pb1 = L.yb0 + j_l * L.db;
    pb_lo = fmin(pb0, pb1);
    pb_hi = fmax(pb0, pb1);
    pa_lo = L.ya0 + fmin(j_f * L.sa, j_l * L.sa) + (double)(t_f * R) * L.da;
    pa_hi = L.ya0 + fmax(j_f * L.sa, j_l * L.sa) + (double)((t_l + 1) * R - 1) * L.da;
}

// The patches see very different numbers of sources (margins of the lattice against its centre), and a
// launch lasts as long as its slowest wavefront: the wavefronts are dealt to the patches in proportion to
// the shares within reach.  load[p] = number of shares whose box meets patch p (one workgroup per patch).
// wstart[p] .. wstart[p + 1] = the wavefronts of patch p: every patch gets one, the remaining n_waves - n_patches go
// by load (integer arithmetic: the same plan for the same data).  The plan is made by the workgroup that finishes
// last (`done`: a counter the estimator keeps at zero between launches).
// lists[p][0 .. load[p]) = the shares within reach of patch p, in share order (ordered compaction: ballot + prefix per 256
// candidates): the wavefronts of a patch take CONTIGUOUS, equal parts of this list (+- one share), where a strided walk
// over all shares with a box test per candidate gave a wavefront 13 +- 4 shares (round 5: the slowest wavefront of a launch
// had 1.66 x the mean work).
constexpr int LOAD_THREADS = 1024;   // (256: 36 dependent rounds of 32-byte loads per sub-patch at C3's size, 21 us; 1 024: 9 rounds)
__global__ void __launch_bounds__(LOAD_THREADS)
kde_lattice_load_kernel(const KdeLattice L, int R, const double *__restrict__ box, int64_t n_shares,
                        unsigned int *__restrict__ load, int32_t *__restrict__ lists, int n_patches, int n_waves,
                        int min_shares, int32_t *__restrict__ wstart, unsigned long long *__restrict__ done) {
    constexpr int NW = LOAD_THREADS / 64;
    constexpr int MAXR = 32;                       // rounds of LOAD_THREADS candidates per pass through the two phases below
    __shared__ unsigned short wcnt[MAXR][NW];
    __shared__ int last;
    double pa_lo, pa_hi, pb_lo, pb_hi;
    lattice_patch_box(L, R, (int)blockIdx.x, pa_lo, pa_hi, pb_lo, pb_hi);
    int32_t *__restrict__ mine_list = lists + (int64_t)blockIdx.x * n_shares;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned int base = 0;
    // Two phases per pass of MAXR rounds: (1) every round's box tests, all loads independent of each other, the wavefronts'
    // counts of every round to LDS; ONE barrier; (2) per round, from the counts, where a wavefront's entries go.  (A barrier
    // and a dependent 32-byte load per round made the 9 rounds of a C3 estimator 9 memory latencies: 18 us per launch.)
    for (int64_t pass0 = 0; pass0 < n_shares; pass0 += (int64_t)MAXR * LOAD_THREADS) {
        const int n_rounds = (int)(((n_shares - pass0) + LOAD_THREADS - 1) / LOAD_THREADS < MAXR
                                       ? ((n_shares - pass0) + LOAD_THREADS - 1) / LOAD_THREADS : MAXR);
        unsigned int okmask = 0;
        if (pass0) __syncthreads();                // (the counts of the previous pass have been read)
        for (int r0 = 0; r0 < n_rounds; r0 += 4) {     // four rounds' boxes requested together (unconditional loads), then tested
            double b[4][4];
            bool in[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int64_t sub = pass0 + (int64_t)(r0 + u) * LOAD_THREADS + threadIdx.x;
                in[u] = r0 + u < n_rounds && sub < n_shares;
                const double *__restrict__ bx = box + 4 * (in[u] ? sub : 0);
#pragma unroll
                for (int e = 0; e < 4; e++) b[u][e] = bx[e];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const bool ok = in[u] && !(b[u][0] > pa_hi || b[u][1] < pa_lo || b[u][2] > pb_hi || b[u][3] < pb_lo);
                okmask |= ok ? (1u << (r0 + u)) : 0u;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
                if (lane == 0 && r0 + u < n_rounds) wcnt[r0 + u][wave] = (unsigned short)__builtin_popcountll(m);
            }
        }
        __syncthreads();
        for (int r = 0; r < n_rounds; r++) {
            const bool ok = (okmask >> r) & 1u;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
            unsigned int before = 0, total = 0;
#pragma unroll
            for (int q = 0; q < NW; q++) {
                const unsigned int c = wcnt[r][q];
                before += q < wave ? c : 0u;
                total += c;
            }
            if (ok)
                mine_list[base + before + (unsigned int)__builtin_popcountll(m & ((1ull << lane) - 1ull))] =
                    (int32_t)(pass0 + (int64_t)r * LOAD_THREADS + threadIdx.x);
            base += total;
        }
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(&load[blockIdx.x], base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        last = atomicAdd(done, 1ull) == (unsigned long long)(n_patches - 1);
    }
    __syncthreads();
    if (!last) return;
    // the plan: the loads fetched by the whole workgroup (one thread walking n_patches dependent agent-scope loads
    // twice cost 45 us at 65 sub-patches), the walk itself on LDS / registers
    __shared__ unsigned long long tot_lds;
    __threadfence();
    if (threadIdx.x == 0) { *done = 0ull; tot_lds = 0ull; }   // back to zero for the next launch
    __syncthreads();
    unsigned long long mine = 0;
    for (int p = threadIdx.x; p < n_patches; p += LOAD_THREADS) mine += __hip_atomic_load(&load[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    atomicAdd(&tot_lds, mine);   // (integer sum: order-independent)
    __syncthreads();
    const unsigned long long total = tot_lds;
    const unsigned long long spare = (unsigned long long)(n_waves - n_patches);
    const unsigned long long per = (unsigned long long)min_shares;   // shares per wavefront at least
    // waves of sub-patch p, then an exclusive prefix sum over the sub-patches in chunks of 256
    __shared__ int32_t scan[256];
    __shared__ int32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const bool act = threadIdx.x < 256;     // (the prefix runs on the first 256 threads; the others only meet the barriers)
    for (int base = 0; base < n_patches; base += 256) {
        const int p = base + (int)threadIdx.x;
        int32_t nw = 0;
        if (act && p < n_patches) {
            const unsigned long long lp = __hip_atomic_load(&load[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long extra = total ? spare * lp / total : 0;
            if (per * (extra + 1) > lp) extra = lp >= per ? lp / per - 1 : 0;
            nw = 1 + (int32_t)extra;
        }
        if (act) scan[threadIdx.x] = nw;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int32_t v = (act && (int)threadIdx.x >= off) ? scan[threadIdx.x - off] : 0;
            __syncthreads();
            if (act) scan[threadIdx.x] += v;
            __syncthreads();
        }
        if (act && p < n_patches) wstart[p] = carry + scan[threadIdx.x] - nw;
        __syncthreads();
        if (threadIdx.x == 255) carry += scan[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) wstart[n_patches] = carry;
}

// Workgroup = one wavefront = one SUB-PATCH of the lattice (sw strips x lpw lines = LG lanes) x every n-th share of the
// sources: the load kernel lists the shares within reach of every sub-patch, the sub-patch's wavefronts (wstart) take
// contiguous, equal parts of its list.  The wavefront's G = 64 / LG lane GROUPS all own the same
// sub-patch, each with accumulators of its own, and work through the wavefront's part of the list side by side (its records
// as ONE stream cut evenly among the groups); the groups' sums are added in the wavefront, partial[wavefront][m][sub-lane];
// the combine kernel adds the wavefronts of a sub-patch in fixed order.
//
// Why groups (round 5; scripts/dev/kde_pass_model.py counts what a shape executes on real estimators): a source reaches a
// disc of the lattice (17 ... 76 lines at the stage's settings); with one 64-line patch per wavefront a pass of the
// recurrence had 24 of 64 lanes within reach (32 with two shares paired lane by lane, the round-3 form), and every record
// was tested by all 64 lanes.  With 8-line sub-patches the share boxes prune per sub-patch, a record is tested by 8 lanes,
// and 46 of 64 lanes are busy per executed pass: 1.45 instead of 1.79 passes and 1.44 instead of 2 x 1.45 record tests
// per source.  The order in which a lane receives its contributions is fixed by the lists: bit-reproducible.
//
// The records reach the wavefront through its own LDS ring: LAT_PIECE records (6 KB) per piece -- LAT_PIECE / G
// consecutive records of every group's share --, every lane fetching 96 B of the next piece (six 16-B loads, contiguous
// runs per group) while the current one is worked through, then storing them to LDS, from where a group reads its
// record back as a broadcast (the groups' parts of a slot are 16 B apart modulo the bank width: no conflicts).  Scalar
// loads, one record ahead, left ~900 cycles of memory latency per source exposed (round 3).
constexpr int LAT_PIECE = 32;   // records per piece, all groups together
template <int R, int LG>
__global__ void __launch_bounds__(64, 3)   // <= 168 VGPRs: at four wavefronts per SIMD (128) the record loop spills
kde_lattice_kernel(const KdeLattice L, double rcut2, const double *__restrict__ rec, int64_t n_src,
                   const int32_t *__restrict__ lists, const unsigned int *__restrict__ load,
                   const int32_t *__restrict__ wstart,
                   int n_patches, double *__restrict__ partial,
                   unsigned long long *__restrict__ pair_count, unsigned long long *__restrict__ stamps) {
    constexpr int C = R / 2;   // the strip's middle point
    constexpr int G = 64 / LG;                  // lane groups = shares side by side
    constexpr int HP = LAT_PIECE / G;           // records of one group per piece
    constexpr int GS = HP * LAT_REC + 2;        // doubles of one group in a ring slot (+ 2: the groups on different banks)
    constexpr int GD2 = HP * LAT_GREC / 2;      // 16-byte units of one group per piece in global memory (eight numbers per record)
    constexpr int UNITS = GD2 / LG;             // ... per lane (two, whatever LG)
    constexpr int N_PIECE = LAT_SHARE / HP;     // pieces per share
    static_assert(C <= LAT_QMAX, "Q table");
    static_assert(G * LG == 64 && HP * G == LAT_PIECE && UNITS * LG == GD2 && N_PIECE * HP == LAT_SHARE && HP <= LG, "piece layout");
    __shared__ __attribute__((aligned(16))) double ring[2][G * GS];
    __shared__ int32_t lst[64];
    const int w = (int)blockIdx.x;
    const int lane = (int)threadIdx.x;
    if (w >= wstart[n_patches]) return;
    const unsigned long long t_start = stamps ? wall_clock64() : 0ull;
    unsigned long long n_steps = 0, n_pass = 0;
    int p = 0;
    {
        int hi = n_patches;
        while (p + 1 < hi) {
            const int mid = (p + hi) >> 1;
            if (wstart[mid] <= w) p = mid; else hi = mid;
        }
    }
    const int split = w - wstart[p], n_split = wstart[p + 1] - wstart[p];
    const int cb = p % L.n_colblk, rb = p / L.n_colblk;
    const int grp = lane / LG, sl = lane % LG;
    const int ls = sl % L.sw, ll = sl / L.sw;
    const int t_f = cb * L.sw, j_f = rb * L.lpw;
    const bool live = ll < L.lpw && t_f + ls < L.strips_a && j_f + ll < L.n1;
    const int t = live ? t_f + ls : t_f, j = live ? j_f + ll : j_f;
    const double yb = L.yb0 + j * L.db;
    const double ya_c = L.ya0 + j * L.sa + (double)(t * R + C) * L.da;
    const double ext_lo = C * L.da, ext_hi = (R - 1 - C) * L.da;
    double acc[R];
#pragma unroll
    for (int k = 0; k < R; k++) acc[k] = 0.0;
    unsigned long long strips = 0;
    typedef double __attribute__((ext_vector_type(2))) d2;
    const int64_t n_shares = (n_src + LAT_SHARE - 1) / LAT_SHARE;

    // one pass of the recurrence for the lane's record rp (LDS), at distance (xc, dbb) from the strip's middle
    auto deposit = [&](const double *rp, double xc, double dbb) {
        double Q[C + 1];
        const double cf = rp[2], sh = rp[3], shd = rp[4], h = rp[5];
#pragma unroll
        for (int kk = 2; kk <= C / 2; kk++) Q[kk] = rp[LAT_Q0 + kk - 2];
        __builtin_amdgcn_sched_barrier(0);   // (the scheduler would sink the reads to their first use)
        const double gc = cf * exp_nonpos(sh * __builtin_fma(xc, xc, dbb * dbb));
        double r_up, r_dn;
        exp_pair(shd * xc, h, r_up, r_dn);   // g(c+1) / g(c), g(c-1) / g(c)
#pragma unroll
        for (int kk = C / 2 + 1; kk <= C; kk++) Q[kk] = rp[LAT_Q0 + kk - 2];
        __builtin_amdgcn_sched_barrier(0);
        double pu = gc, pd = gc * r_dn;
#pragma unroll
        for (int kk = 0; kk < C; kk++) {   // both directions interleaved: two independent chains
            // up: point C + kk (k = kk); down: point C - 1 - kk (k = kk + 1)
            acc[C + kk] = kk < 2 ? acc[C + kk] + pu : __builtin_fma(pu, Q[kk], acc[C + kk]);
            if (kk + 1 < R - C) pu *= r_up;
            acc[C - 1 - kk] = kk + 1 < 2 ? acc[C - 1 - kk] + pd : __builtin_fma(pd, Q[kk + 1], acc[C - 1 - kk]);
            if (kk + 1 < C) pd *= r_dn;
        }
    };

    // The sorted sources are cut into shares of LAT_SHARE sources (compact in y_a and y_b); the load kernel has listed the
    // shares within reach of this sub-patch, in share order; the sub-patch's wavefronts take contiguous, equal parts of
    // that list (+- one share), 64 entries at a time.
    const int32_t *__restrict__ my_list = lists + (int64_t)p * n_shares;
    const int64_t n_in = (int64_t)load[p];
    const int64_t e_end = ((int64_t)(split + 1) * n_in) / n_split;
    for (int64_t e0 = ((int64_t)split * n_in) / n_split; e0 < e_end; e0 += 64) {
        const int nl = (int)(e_end - e0 < 64 ? e_end - e0 : 64);
        __syncthreads();
        if (lane < nl) lst[lane] = my_list[e0 + lane];
        __syncthreads();
        // The list is ONE stream of nl x 64 records, cut evenly among the groups: group g works records
        // [g nl LG, (g + 1) nl LG) of it, piece by piece (a piece never crosses a share: HP divides 64 and nl LG).
        // (Whole shares per group left the groups of a last, partial round idle: 10 % of the steps at G = 8.)
        {
            const int n_pc = nl * (LG / HP);   // pieces per group
            const int pos0 = grp * nl * LG;
            auto piece_src = [&](int c) -> const d2 * {
                const int pos = pos0 + c * HP;
                return reinterpret_cast<const d2 *>(rec + ((int64_t)lst[pos >> 6] * LAT_SHARE + (pos & 63)) * LAT_GREC) + sl;
            };
            // (the record array is padded by a whole share: the loads of a last, partial share stay inside it)
            d2 a[UNITS], b[UNITS];
            const d2 *__restrict__ src = piece_src(0);
#pragma unroll
            for (int u = 0; u < UNITS; u++) a[u] = src[u * LG];
            for (int c = 0; c < n_pc; c++) {
                const int pos = pos0 + c * HP;
                const int64_t k0 = (int64_t)lst[pos >> 6] * LAT_SHARE + (pos & 63);   // first record of this piece
                if (c + 1 < n_pc) {
                    const d2 *__restrict__ nx = piece_src(c + 1);
#pragma unroll
                    for (int u = 0; u < UNITS; u++) b[u] = nx[u * LG];
                }
                double *slot = ring[c & 1] + grp * GS;   // this group's part of the slot
                {
                    // unit uid of the piece = numbers 2 (uid % 4), 2 (uid % 4) + 1 of its record uid / 4, at the record's place in LDS
                    d2 *dst = reinterpret_cast<d2 *>(slot);
#pragma unroll
                    for (int u = 0; u < UNITS; u++) {
                        const int uid = sl + u * LG;
                        dst[(uid >> 2) * (LAT_REC / 2) + (uid & 3)] = a[u];
                    }
                }
                __syncthreads();   // one wavefront: orders the stores above against the reads below
                if (sl < HP) {
                    // the record's Q table, rebuilt here from h: Q_2 = h^2, Q_k = Q_{k-1} h^(2 (k - 1)) (lane sl of the group: record sl)
                    double *q = slot + sl * LAT_REC;
                    const double hh = q[5] * q[5];
                    double t = hh, qk = hh;
                    q[LAT_Q0] = qk;
#pragma unroll
                    for (int kk = 3; kk <= C; kk++) {
                        t *= hh;
                        qk *= t;
                        q[LAT_Q0 + kk - 2] = qk;
                    }
                }
                __syncthreads();
                const int nr = (int)(n_src - k0 < HP ? n_src - k0 : HP);   // records of this piece that exist
                // (y_a, y_b, -s2 / 2) of the record, read one iteration ahead
                double ay = slot[0], by = slot[1], shh = slot[3];
                for (int r = 0; r < HP; r++) {
                    const double *rp = slot + r * LAT_REC;
                    const double *np_ = slot + (r + 1 < HP ? r + 1 : r) * LAT_REC;
                    const double n0 = np_[0], n1 = np_[1], n3 = np_[3];
                    const double xc = ya_c - ay, dbb = yb - by;
                    const double dn = fmax(fmax(xc - ext_lo, -(xc + ext_hi)), 0.0);   // nearest point of the strip
                    const bool in = live && r < nr && (dbb * dbb + dn * dn) * shh * -2.0 <= rcut2;
                    n_steps++;
                    if (__builtin_amdgcn_ballot_w64(in)) {
                        n_pass++;
                        if (in) {
                            strips++;
                            deposit(rp, xc, dbb);
                        }
                    }
                    ay = n0; by = n1; shh = n3;
                }
                __syncthreads();   // the slot is overwritten two pieces later: its reads are done
#pragma unroll
                for (int u = 0; u < UNITS; u++) a[u] = b[u];
            }
        }
    }
    if (stamps && lane == 0) {   // development: when this wavefront ran and what it did
        stamps[4 * w] = t_start;
        stamps[4 * w + 1] = wall_clock64();
        stamps[4 * w + 2] = n_steps;
        stamps[4 * w + 3] = (n_pass << 16) | (unsigned long long)p;
    }
    // The G groups' accumulators of a point are added here, in group order, through the (now idle) ring: the wavefront
    // leaves R x LG partial sums instead of R x 64 (2 KB instead of 16 KB at LG = 8), so that a launch can be cut into
    // twice the resident wavefronts -- which fills the tail of the launch, see the host side -- without 100 MB of partials.
    {
        double *stage = &ring[0][0];                     // [m of this half][64 lanes], 8 KB per half
        constexpr int HALF = R / 2 > 16 ? 16 : R / 2;    // m's per pass through the ring
        static_assert(HALF * 64 <= 2 * G * GS && R % HALF == 0, "staging fits the ring");
        constexpr int PER_LANE = (HALF * LG + 63) / 64;  // (m, sub-lane) sums per lane and pass
        double *out = partial + (int64_t)w * (R * LG);
#pragma unroll
        for (int h = 0; h < R / HALF; h++) {
            __syncthreads();
#pragma unroll
            for (int m = 0; m < HALF; m++) stage[m * 64 + lane] = acc[h * HALF + m];
            __syncthreads();
#pragma unroll
            for (int q = 0; q < PER_LANE; q++) {
                const int e = q * 64 + lane;             // entry (m, sub-lane) of this half
                if (e < HALF * LG) {
                    const int m = e / LG, s_ = e % LG;
                    double v = stage[m * 64 + s_];
#pragma unroll
                    for (int g = 1; g < G; g++) v += stage[m * 64 + g * LG + s_];
                    out[(h * HALF + m) * LG + s_] = v;
                }
            }
        }
    }
    if (pair_count) {
        strips *= R;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) strips += __shfl_down(strips, off);
        if (lane == 0 && strips) atomicAdd(pair_count, strips);   // (one per wavefront, spread over the launch: 2 % of it, measured without)
    }
}

// out[i0 n1 + i1] = sum over the wavefronts of the point's sub-patch of partial[wavefront][m][sub-lane].  One workgroup
// per (sub-patch, m): 256 threads = LG sub-lanes x 256 / LG adders; adder a takes the wavefronts a, a + A, ... in order,
// the A sums are added in order: fixed association, bit-reproducible.
__global__ void __launch_bounds__(256)
kde_lattice_combine_kernel(const double *__restrict__ partial, const KdeLattice L, int R,
                           const int32_t *__restrict__ wstart, double *__restrict__ out) {
    __shared__ double lds[256];
    const int LG = L.sw * L.lpw, A = 256 / LG;
    const int sl = (int)threadIdx.x % LG, a = (int)threadIdx.x / LG;
    const int m = (int)blockIdx.x % R, p = (int)blockIdx.x / R;
    const int ws = wstart[p], we = wstart[p + 1];
    double acc = 0.0;
    for (int w = ws + a; w < we; w += A) acc += partial[(int64_t)w * (R * LG) + m * LG + sl];
    lds[threadIdx.x] = acc;
    __syncthreads();
    if (a == 0) {
        double v = lds[sl];
        for (int g = 1; g < A; g++) v += lds[g * LG + sl];
        const int cb = p % L.n_colblk, rb = p / L.n_colblk;
        const int ls = sl % L.sw, ll = sl / L.sw;
        const int t = cb * L.sw + ls, j = rb * L.lpw + ll;
        const int i0 = t * R + m;
        if (t < L.strips_a && j < L.n1 && i0 < L.n0) out[(int64_t)i0 * L.n1 + j] = v;
    }
}

// lattice -> explicit points x[d][i0 * n1 * n2 + i1 * n2 + i2] (for the general evaluation)
__global__ void __launch_bounds__(256)
kde_lattice_points_kernel(int dim, double o0, double o1, double o2, double s0, double s1, double s2_,
                          int n1, int n2, int64_t m, double *__restrict__ x) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const int64_t i2 = i % n2, i1 = (i / n2) % n1, i0 = i / ((int64_t)n1 * n2);
    x[i] = o0 + (double)i0 * s0;
    if (dim > 1) x[m + i] = o1 + (double)i1 * s1;
    if (dim > 2) x[2 * m + i] = o2 + (double)i2 * s2_;
}

// sum of log(pilot) and the number of sources it runs over -> partial[block][2].  A source of weight ZERO contributes
// nothing to any density; it is left out of the geometric mean and keeps the global bandwidth.  (Its pilot density
// is whatever its weighted neighbours leave there: exactly 0 beyond the cut-off -- log 0 would turn every local
// bandwidth into NaN --, tiny but positive just inside it: counting such sources made the estimate depend on the
// cut-off, found by scripts/dev/fuzz_kde.py in round 4.  A weighted source always has pilot > 0, its own term.)
__global__ void __launch_bounds__(RED_THREADS)
kde_logsum_kernel(const double *__restrict__ pilot, const double *__restrict__ wn, int64_t n,
                  double *__restrict__ partial) {
    __shared__ double lds[RED_THREADS];
    double s = 0.0, c = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * RED_THREADS + threadIdx.x; i < n;
         i += (int64_t)RED_BLOCKS * RED_THREADS) {
        const double v = pilot[i];
        if (wn[i] > 0.0 && v > 0.0) {
            s += log(v);
            c += 1.0;
        }
    }
    const double r = block_sum(s, lds);
    const double rc = block_sum(c, lds);
    if (threadIdx.x == 0) {
        partial[blockIdx.x * 2] = r;
        partial[blockIdx.x * 2 + 1] = rc;
    }
}

// local bandwidths: lam = (pilot / g)^alpha, s2 = lam^2, coef = wn lam^d / norm; per-block min s2
// (logsum[block][2] = kde_logsum_kernel's partial results: sum of log pilot, number of sources of positive pilot density)
__global__ void __launch_bounds__(RED_THREADS)
kde_bandwidth_kernel(const double *__restrict__ pilot, const double *__restrict__ wn, int64_t n,
                     const double *__restrict__ logsum, double alpha, int dim, double inv_norm,
                     double *__restrict__ coef, double *__restrict__ s2, double *__restrict__ partial_min) {
    __shared__ double lds[RED_THREADS];
    // every workgroup joins kde_logsum_kernel's partial results itself (kde_final_reduce_kernel's pairing: one launch less)
    const double lsum = block_sum(logsum[threadIdx.x * 2], lds);
    const double lcnt = block_sum(logsum[threadIdx.x * 2 + 1], lds);
    const double glob = exp(lsum / lcnt);
    double mn = INFINITY, mx = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * RED_THREADS + threadIdx.x; i < n;
         i += (int64_t)RED_BLOCKS * RED_THREADS) {
        const double pv = pilot[i];
        const double lam = (wn[i] > 0.0 && pv > 0.0) ? pow(pv / glob, alpha) : 1.0;
        const double l2 = lam * lam;
        s2[i] = l2;
        coef[i] = wn[i] * (dim == 1 ? lam : (dim == 2 ? l2 : l2 * lam)) * inv_norm;
        mn = fmin(mn, l2);
        mx = fmax(mx, l2);
    }
    const double r = block_min(mn, lds);
    const double r2 = -block_min(-mx, lds);
    if (threadIdx.x == 0) {   // [block][2]: smallest and largest s2 (widest and narrowest kernel)
        partial_min[blockIdx.x * 2] = r;
        partial_min[blockIdx.x * 2 + 1] = r2;
    }
}

__global__ void __launch_bounds__(256)
kde_cell_s2min_kernel(const double *__restrict__ s2, const int32_t *__restrict__ cell_start,
                      int64_t n_cells, double *__restrict__ cell_s2min) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n_cells) return;
    double mn = INFINITY;
    for (int k = cell_start[c]; k < cell_start[c + 1]; k++) mn = fmin(mn, s2[k]);
    cell_s2min[c] = mn;
}


// ------------------------------------------------------------------ Hermite expansion of dense cells
// Fixed-bandwidth sums (the pilot estimate) in 2-D: in u = y / sqrt(2) the kernel is exp(-|dt|^2)
// and the sources of one cell (centre c, half side rho = cell / (2 sqrt 2) = 0.355) act on a
// target t through the Hermite series  (Greengard & Strain 1991)
//     sum_j q_j exp(-|t - s_j|^2) = sum_{n,m} A_nm h_n(t1 - c1) h_m(t2 - c2),
//     A_nm = sum_j q_j (s_j1 - c1)^n (s_j2 - c2)^m / (n! m!),   h_n(x) = (-1)^n d^n/dx^n exp(-x^2).
// Truncated at n, m < P the error per cell is below  2.3 K^2 (rho sqrt2)^P / sqrt(P!) * sum_j q_j
// (K = 1.09): 1e-13 for P = 18, 1.5e-15 for P = 20 -- relative to the cell's total weight, i.e.
// at or below the cut-off tolerance.  A target then costs P^2 + 4 P + 40 multiply-adds per dense
// cell within the cut-off instead of 23 per SOURCE: N * 250 cells * 500 instead of N * 0.2 N * 23
// (x 90 at N = 4e5).  Without local expansions, cells with fewer than HERMITE_MIN_SERIES sources are summed directly.
constexpr int HERMITE_MIN_DEFAULT = 1;   // with local expansions (see pisa_hip_kde_create)
constexpr int HERMITE_MIN_SERIES = 24;   // series evaluated per target
static int hermite_min() {
    static const int v = [] { const int x = PISA_DEV_INT("KDE_HERMITE_MIN", 0); return x > 0 ? x : HERMITE_MIN_DEFAULT; }();
    return v;
}
constexpr double RSQRT2 = 0.70710678118654752440;

__global__ void __launch_bounds__(256)
kde_slot_scatter_kernel(const int32_t *__restrict__ dense_cells, int n_dense, int32_t *__restrict__ slot) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_dense) slot[dense_cells[i]] = i;
}

// A[slot][n][m] of one dense cell per workgroup; fixed order => reproducible.  The P x P coefficients are cut
// into 4 x 4 register tiles; the sources of a staged tile of 64 are dealt to four thread groups (j = s, s + 4, ...),
// each group holding a partial sum of every tile, added in group order at the end: 4 LDS reads (16 bytes each) per
// 16 multiply-adds.
constexpr int HC_THREADS = 128;
template <int P>
__global__ void __launch_bounds__(HC_THREADS)
kde_hermite_coef_kernel(KdeGeom g, const int32_t *__restrict__ dense_cells,
                        const int32_t *__restrict__ cell_start, const double *__restrict__ sy,
                        int64_t n_src, const double *__restrict__ coef, double *__restrict__ herm) {
    constexpr int TJ = 64, PT = (P + 3) / 4, PR = PT * 4, NG = 4;
    static_assert(PT * PT * NG <= HC_THREADS, "tiles x groups must fit the workgroup");
    typedef double __attribute__((ext_vector_type(2))) d2;
    __shared__ __attribute__((aligned(16))) double stage[2][TJ][PR];   // powers of both coordinates, zero beyond P
    static_assert(2 * TJ * PR >= NG * PR * PR, "the partial sums reuse the staging area");
    double (*pa)[PR] = stage[0], (*pb)[PR] = stage[1];
    const int t = threadIdx.x;
    const int c = dense_cells[blockIdx.x];
    const int cx = c % g.nc[0], cy = c / g.nc[0];
    const double c1 = g.ylo[0] + (cx + 0.5) * g.cell, c2 = g.ylo[1] + (cy + 0.5) * g.cell;
    const int begin = cell_start[c], end = cell_start[c + 1];
    const int grp = t / (PT * PT), tile = t % (PT * PT);
    const bool act = grp < NG;
    const int ti = tile / PT, tj = tile % PT;
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = 0.0;
    for (int base = begin; base < end; base += TJ) {
        const int cnt = end - base < TJ ? end - base : TJ;
        __syncthreads();
        {
            const int j = t % TJ, which = t / TJ;   // 128 threads = 64 sources x 2 coordinates
            double *dst = which ? pb[j] : pa[j];
            if (j < cnt) {
                const double d = (sy[(int64_t)which * n_src + base + j] - (which ? c2 : c1)) * RSQRT2;
                double v = which ? coef[base + j] : 1.0;   // the weight rides on the second factor
#pragma unroll
                for (int n = 0; n < P; n++) {
                    dst[n] = v;
                    v = v * d * (1.0 / (double)(n + 1));   // (the reciprocal is a compile-time constant)
                }
#pragma unroll
                for (int n = P; n < PR; n++) dst[n] = 0.0;
            }
        }
        __syncthreads();
        if (act) {
            for (int j = grp; j < cnt; j += NG) {
                const d2 a01 = *reinterpret_cast<const d2 *>(&pa[j][4 * ti]), a23 = *reinterpret_cast<const d2 *>(&pa[j][4 * ti + 2]);
                const d2 b01 = *reinterpret_cast<const d2 *>(&pb[j][4 * tj]), b23 = *reinterpret_cast<const d2 *>(&pb[j][4 * tj + 2]);
                const double a[4] = {a01.x, a01.y, a23.x, a23.y}, bb[4] = {b01.x, b01.y, b23.x, b23.y};
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int k = 0; k < 4; k++) acc[i][k] = __builtin_fma(a[i], bb[k], acc[i][k]);
            }
        }
    }
    __syncthreads();
    double *red = &stage[0][0][0];   // [group][PR * PR]
    if (act) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int k = 0; k < 4; k++) red[grp * (PR * PR) + (4 * ti + i) * PR + 4 * tj + k] = acc[i][k];
    }
    __syncthreads();
    for (int e = t; e < P * P; e += HC_THREADS) {
        const int n = e / P, m = e % P;
        double v = red[n * PR + m];
#pragma unroll
        for (int s = 1; s < NG; s++) v += red[s * (PR * PR) + n * PR + m];
        herm[(int64_t)blockIdx.x * (P * P) + e] = v;
    }
}

// 1 / n!, n < 24 (the factorial scaling of the local expansion: two multiplications instead of two loops and a division per element)
__device__ constexpr double INV_FACT[24] = {1.0, 1.0, 0.5, 0.16666666666666666, 0.041666666666666664, 0.008333333333333333, 0.001388888888888889, 0.0001984126984126984, 2.48015873015873e-05, 2.7557319223985893e-06, 2.755731922398589e-07, 2.505210838544172e-08, 2.08767569878681e-09, 1.6059043836821613e-10, 1.1470745597729725e-11, 7.647163731819816e-13, 4.779477332387385e-14, 2.8114572543455206e-15, 1.5619206968586225e-16, 8.22063524662433e-18, 4.110317623312165e-19, 1.9572941063391263e-20, 8.896791392450574e-22, 3.8681701706306835e-23};
typedef double __attribute__((ext_vector_type(4))) mfma_d4;
// The same coefficients on the matrix cores (round 5; series orders up to 16): A = X^T Y with X[j][n] = d1_j^n / n!,
// Y[j][m] = q_j d2_j^m / m! over the sources j of the cell -- a 16 x J by J x 16 product, four sources per
// `v_mfma_f64_16x16x4_f64`: lane (i = lane & 15, k = lane >> 4) supplies X[4 s + k][i] as the A and Y[4 s + k][i] as the B
// operand.  The vector form above stages the powers of 64 sources in LDS between two workgroup barriers, multiplies 4 x 4
// register tiles and adds four partial sums through LDS: 37 us per estimator for 0.2 GFLOP.  The products are summed in
// the matrix cores' order: the coefficients differ from the vector form's by rounding (1e-16 relative), run to run the same.
constexpr int FGT_WAVES = 4;   // wavefronts per workgroup of the matrix-core kernels: they share the sources of ONE cell / the source
                               // positions of ONE target pair and add their sums in wavefront order through LDS
__global__ void __launch_bounds__(64 * FGT_WAVES)
kde_hermite_coef_mfma_kernel(KdeGeom g, const int32_t *__restrict__ dense_cells, int n_dense,
                             const int32_t *__restrict__ cell_start,
                             const double *__restrict__ sy, int64_t n_src, const double *__restrict__ coef, int P,
                             double *__restrict__ herm) {
    // A workgroup = one cell, its FGT_WAVES wavefronts take every FGT_WAVES-th batch of 32 sources and add their sums in
    // wavefront order through LDS.  A batch: lanes 0-31 form the 16 scaled powers of the first coordinate of "their" source
    // by the running product v <- v d / (n + 1) (the vector kernel's values), lanes 32-63 those of the second coordinate
    // times the weight; the powers go through the wavefront's own LDS tile [power][source] (row stride 36 doubles: the
    // operand reads below fall on all banks) and come back in the operand layout -- lane (i, k) reads power i of source
    // 4 s + k.  (Each lane forming ITS power of ITS source by square-and-multiply, no LDS, cost 8 vector instructions per
    // source against 2 here: 28 us per estimator, 58 us beside the lattice kernels of the other streams.)
    constexpr int BATCH = 32, ROW = 36;
    __shared__ double tile[FGT_WAVES][2][16 * ROW];
    __shared__ double red[FGT_WAVES][256];
    const int lane = (int)threadIdx.x & 63, li = lane & 15, lk = lane >> 4, wave = (int)threadIdx.x >> 6;
    const int slot_i = (int)blockIdx.x;
    const int c = dense_cells[slot_i];
    const int cx = c % g.nc[0], cy = c / g.nc[0];
    const int half = lane >> 5, js = lane & 31;                  // which coordinate, which source of the batch
    const double centre = half ? g.ylo[1] + (cy + 0.5) * g.cell : g.ylo[0] + (cx + 0.5) * g.cell;
    const int begin = cell_start[c], end = cell_start[c + 1];
    const double *__restrict__ coord = sy + (half ? n_src : 0);
    double *mine = &tile[wave][half][0];
    const double *tx = &tile[wave][0][0], *ty = &tile[wave][1][0];
    mfma_d4 acc = (mfma_d4){0.0, 0.0, 0.0, 0.0};
    constexpr int STEP = BATCH * FGT_WAVES;
    // (the next batch's sources are requested before this batch's powers are formed)
    auto fetch = [&](int base, double &d, double &q) {
        const int j = base + js;
        const int64_t jj = j < end ? j : begin;                  // (unconditional loads; a source beyond the cell gets weight zero)
        d = coord[jj];
        q = coef[jj];
    };
    double d_cur, q_cur, d_nxt, q_nxt;
    fetch(begin + BATCH * wave, d_cur, q_cur);
    for (int base = begin + BATCH * wave; base < end; base += STEP) {
        fetch(base + STEP < end ? base + STEP : begin, d_nxt, q_nxt);
        const double d = (d_cur - centre) * RSQRT2;
        double v = half ? (base + js < end ? q_cur : 0.0) : 1.0;   // the weight rides on the second factor
#pragma unroll
        for (int n = 0; n < 16; n++) {
            mine[n * ROW + js] = v;
            v = v * d * (1.0 / (double)(n + 1));                 // (the reciprocal is a compile-time constant)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int sx = 0; sx < BATCH / 4; sx++) {
            const double a = tx[li * ROW + 4 * sx + lk], b = ty[li * ROW + 4 * sx + lk];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc, 0, 0, 0);   // Y^T X = A^T: stored TRANSPOSED, see below
        }
        __builtin_amdgcn_wave_barrier();                         // (the tile is rewritten by the next batch)
        d_cur = d_nxt;
        q_cur = q_nxt;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) red[wave][r * 64 + lane] = acc[r];
    __syncthreads();
    static_assert(FGT_WAVES == 4, "one accumulator register per wavefront in the final sum");
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < FGT_WAVES; q++) v += red[q][wave * 64 + lane];
    // herm[slot][m][n] = A[n][m]: the first translation pass (kde_h2l_mfma_kernel<0>, the only reader of this form) takes
    // A[alpha][beta] as its A operand, lane (alpha, k) <- element [alpha][4 s + k]; from the transposed array that is a
    // row of 16 consecutive doubles per k instead of 16 rows of 32 bytes (the loads of that pass: 30 -> 21 us)
    const int row = lk + 4 * wave;
    if (row < P && li < P) herm[(int64_t)slot_i * (P * P) + row * P + li] = v;
}

// pilot densities at the (cell-sorted) sources: dense cells through their Hermite series, sparse
// ones directly.  Workgroup = one KdeBlock (<= 512 targets of ONE cell) x one share of the cells.
template <int P>
__global__ void __launch_bounds__(KDE_THREADS)
kde_hermite_pilot_kernel(KdeGeom g, const KdeBlock *__restrict__ blocks, const double *__restrict__ sy,
                         int64_t n_src, const double *__restrict__ coef,
                         const int32_t *__restrict__ cell_start, const int32_t *__restrict__ slot,
                         const double *__restrict__ herm, int n_split, double *__restrict__ out,
                         unsigned long long *__restrict__ pair_count) {
    constexpr int W = 4;
    __shared__ double t_src[SRC_TILE * W];
    const KdeBlock b = blocks[blockIdx.x];
    const int split = blockIdx.y;
    double q[Q_PER_THREAD][2], acc[Q_PER_THREAD];
#pragma unroll
    for (int u = 0; u < Q_PER_THREAD; u++) {
        const int jq = threadIdx.x + u * KDE_THREADS;
        const int64_t j = b.q_begin + (jq < b.q_count ? jq : 0);
        q[u][0] = sy[j];
        q[u][1] = sy[n_src + j];
        acc[u] = 0.0;
    }
    int lo[2], hi[2];
    const double reach = ceil(sqrt(g.rcut2) * g.inv_cell);
#pragma unroll
    for (int d = 0; d < 2; d++) {
        const double l = (double)b.c0[d] - reach, h = (double)b.c1[d] + reach;
        lo[d] = l > 0.0 ? (int)l : 0;
        hi[d] = h < (double)(g.nc[d] - 1) ? (int)h : g.nc[d] - 1;
    }
    unsigned long long work = 0;
    int visited = 0;
    for (int cy = lo[1]; cy <= hi[1]; cy++) {
        const int gapy = cy < b.c0[1] ? b.c0[1] - cy - 1 : (cy > b.c1[1] ? cy - b.c1[1] - 1 : 0);
        const double gy = gapy * g.cell;
        const int64_t row = (int64_t)cy * g.nc[0];
        for (int cx = lo[0]; cx <= hi[0]; cx++) {
            const int64_t c = row + cx;
            const int begin = cell_start[c], end = cell_start[c + 1];
            if (end == begin) continue;
            const int gapx = cx < b.c0[0] ? b.c0[0] - cx - 1 : (cx > b.c1[0] ? cx - b.c1[0] - 1 : 0);
            const double gx = gapx * g.cell;
            if (gx * gx + gy * gy > g.rcut2) continue;
            if ((visited++) % n_split != split) continue;
            const int sl = slot[c];
            if (sl >= 0) {
                // Hermite series of the cell; coefficient addresses are wave-uniform (scalar loads)
                const double *__restrict__ A = herm + (int64_t)sl * (P * P);
                const double c1 = g.ylo[0] + (cx + 0.5) * g.cell, c2 = g.ylo[1] + (cy + 0.5) * g.cell;
                work += (unsigned long long)(P * P / 23 + 1) * b.q_count;
                // both targets of the thread go through the series together: every coefficient is
                // loaded once (a row of P at a time lives in scalar registers) and used twice
                double hm[Q_PER_THREAD][P], t1x2[Q_PER_THREAD], hp[Q_PER_THREAD], hn[Q_PER_THREAD],
                    sum[Q_PER_THREAD];
#pragma unroll
                for (int u = 0; u < Q_PER_THREAD; u++) {
                    const double t1 = (q[u][0] - c1) * RSQRT2, t2 = (q[u][1] - c2) * RSQRT2;
                    hm[u][0] = exp_nonpos(-t2 * t2);
                    hm[u][1] = 2.0 * t2 * hm[u][0];
#pragma unroll
                    for (int m = 1; m + 1 < P; m++)
                        hm[u][m + 1] = __builtin_fma(2.0 * t2, hm[u][m], -2.0 * m * hm[u][m - 1]);
                    t1x2[u] = 2.0 * t1;
                    hp[u] = 0.0;
                    hn[u] = exp_nonpos(-t1 * t1);
                    sum[u] = 0.0;
                }
#pragma unroll 1
                for (int n = 0; n < P; n++) {
                    const double *__restrict__ row = A + n * P;
                    double inner[Q_PER_THREAD];
#pragma unroll
                    for (int u = 0; u < Q_PER_THREAD; u++) inner[u] = 0.0;
#pragma unroll
                    for (int m = 0; m < P; m++) {
                        const double a = row[m];
#pragma unroll
                        for (int u = 0; u < Q_PER_THREAD; u++) inner[u] = __builtin_fma(a, hm[u][m], inner[u]);
                    }
#pragma unroll
                    for (int u = 0; u < Q_PER_THREAD; u++) {
                        sum[u] = __builtin_fma(hn[u], inner[u], sum[u]);
                        const double nx = __builtin_fma(t1x2[u], hn[u], -2.0 * n * hp[u]);
                        hp[u] = hn[u];
                        hn[u] = nx;
                    }
                }
#pragma unroll
                for (int u = 0; u < Q_PER_THREAD; u++) acc[u] += sum[u];
                continue;
            }
            work += (unsigned long long)(end - begin) * b.q_count;
            for (int base = begin; base < end; base += SRC_TILE) {
                const int cnt = end - base < SRC_TILE ? end - base : SRC_TILE;
                __syncthreads();
                for (int k = threadIdx.x; k < cnt; k += KDE_THREADS) {
                    t_src[k * W + 0] = sy[base + k];
                    t_src[k * W + 1] = sy[n_src + base + k];
                    t_src[k * W + 2] = coef[base + k];
                }
                __syncthreads();
                for (int k = 0; k < cnt; k++) {
                    const double *s = t_src + k * W;
#pragma unroll
                    for (int u = 0; u < Q_PER_THREAD; u++) {
                        const double d0 = q[u][0] - s[0], d1 = q[u][1] - s[1];
                        const double r2 = __builtin_fma(d1, d1, d0 * d0);
                        acc[u] = __builtin_fma(s[2], exp_nonpos(-0.5 * r2), acc[u]);
                    }
                }
            }
        }
    }
    out += (int64_t)split * n_src;
#pragma unroll
    for (int u = 0; u < Q_PER_THREAD; u++) {
        const int jq = threadIdx.x + u * KDE_THREADS;
        if (jq < b.q_count) out[b.q_begin + jq] = acc[u];
    }
    if (pair_count && threadIdx.x == 0 && work) atomicAdd(pair_count, work);
}


// ------------------------------------------------------------------ Hermite -> local (Taylor) translation
// With many targets per cell the Hermite series of the ~250 dense cells in range need not be
// evaluated target by target: they are first translated into ONE local expansion about the
// target cell's centre c_C (x = t - c_C, |x| <= rho),
//     sum_B sum_{a,b} A^B_ab h_a(t1 - c_B1) h_b(t2 - c_B2)  =  sum_{k,l} L_kl x1^k x2^l,
//     L = D (sum_B H(d1) A^B H(d2)^T) D,   H(d)[k][a] = h_{a+k}(d),  d = c_C - c_B,  D = diag((-1)^k / k!)
// (h_a(d + x) = sum_k (-1)^k h_{a+k}(d) x^k / k!), and a target then costs P^2 multiply-adds in
// all.  The sum over the source cells of one column (same d1) is formed first, W = sum_B A^B H(d2)^T,
// and multiplied by H(d1) once per column: (cells + columns) P^3 multiply-adds per target cell.
// Truncation at k, l < P costs the same (rho sqrt2)^P / sqrt(P!) as the Hermite series.  The Hankel
// entries h_n(j * cell / sqrt2), n < 2P - 1, |j| <= reach, come from the host (long double recurrence).
constexpr int H2L_MAX_REACH = 12;

// The translation is done in two separable passes over the cell grid (like a separable
// convolution), with the SQUARE |d1|, |d2| <= reach of source cells instead of the disc -- the cells
// of the square outside the cut-off disc only add their (correct, below-tolerance) contributions:
//   pass 0   V[cy_C][cx_B] = sum_{cy_B} A^(cx_B, cy_B) H(d2)^T          one workgroup per (cx_B, cy_C)
//   pass 1   L[C]          = D (sum_{cx_B} H(d1) V[cy_C][cx_B]) D       one workgroup per target cell
// 2 (2 reach + 1) P^3 multiply-adds per cell instead of (cells in the disc + columns) P^3.
// Each thread owns a 2 x 2 tile of the P x P result ((P/2)^2 threads of a 128-thread workgroup): a
// multiply-add then costs 0.75 LDS reads instead of 2 (one element per thread: the kernel waited for
// the LDS pipe, 3 TFLOP/s).  Every element is still summed in the same order (offsets ascending, inner
// index ascending): same bits.
constexpr int H2L_THREADS = 128;
template <int P, int PASS>
__global__ void __launch_bounds__(H2L_THREADS)
kde_h2l_kernel(KdeGeom g, const int32_t *__restrict__ tcells, const int32_t *__restrict__ slot,
               const double *__restrict__ herm, const double *__restrict__ hankel, int reach,
               double *__restrict__ V, uint8_t *__restrict__ vflag, double *__restrict__ local) {
    static_assert(P % 2 == 0, "2 x 2 tiles");
    constexpr int NH = 2 * P - 1, H = P / 2, PP = P * P;
    constexpr int PER = (PP + H2L_THREADS - 1) / H2L_THREADS;   // matrix elements a thread stages
    __shared__ double sA[PP];
    __shared__ double sH[(2 * H2L_MAX_REACH + 1) * NH];
    const int t = threadIdx.x;
    const bool act = t < H * H;
    const int ti = act ? t / H : 0, tj = act ? t % H : 0;
    for (int i = t; i < (2 * reach + 1) * NH; i += H2L_THREADS) sH[i] = hankel[i];
    const int nx = g.nc[0], ny = g.nc[1];
    const int c = PASS == 0 ? (int)blockIdx.x : tcells[blockIdx.x];
    const int cx = c % nx, cy = c / nx;
    double a00 = 0.0, a01 = 0.0, a10 = 0.0, a11 = 0.0;
    bool any = false;
    // source matrix of offset o, or nullptr (workgroup-uniform)
    auto source = [&](int o) -> const double * {
        if (o > reach) return nullptr;
        if (PASS == 0) {                 // source cell (cx, cy - o): d2 = o * cell_u
            const int cyB = cy - o;
            if (cyB < 0 || cyB >= ny) return nullptr;
            const int sl = slot[(int64_t)cyB * nx + cx];
            return sl < 0 ? nullptr : herm + (int64_t)sl * PP;
        }
        const int cxB = cx - o;          // column cx - o of row cy: d1 = o * cell_u
        if (cxB < 0 || cxB >= nx) return nullptr;
        const int64_t cb = (int64_t)cy * nx + cxB;
        return vflag[cb] ? V + cb * PP : nullptr;
    };
    auto fetch = [&](const double *src, double (&buf)[PER]) {
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int e = t + u * H2L_THREADS;
            buf[u] = (src && e < PP) ? src[e] : 0.0;
        }
    };
    // The matrices are 3.2 KB each and come from the L2 / HBM: the next one is requested before the
    // current one is multiplied.
    int o = -reach;
    const double *src = source(o);
    while (!src && o <= reach) src = source(++o);
    double mine[PER], ahead[PER];
    fetch(src, mine);
    while (src) {
        int on = o + 1;
        const double *nxt = source(on);
        while (!nxt && on <= reach) nxt = source(++on);
        fetch(nxt, ahead);
        any = true;                      // workgroup-uniform
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int e = t + u * H2L_THREADS;
            if (e < PP) sA[e] = mine[u];
        }
        __syncthreads();
        if (act) {
            const double *hk = sH + (o + reach) * NH;
            if (PASS == 0) {             // W[alpha][l] += sum_beta A[alpha][beta] h_{beta + l}(d2)
                const double *r0 = sA + (2 * ti) * P, *r1 = r0 + P, *h = hk + 2 * tj;
#pragma unroll
                for (int b = 0; b < P; b++) {
                    const double A0 = r0[b], A1 = r1[b], h0 = h[b], h1 = h[b + 1];
                    a00 = __builtin_fma(A0, h0, a00);
                    a01 = __builtin_fma(A0, h1, a01);
                    a10 = __builtin_fma(A1, h0, a10);
                    a11 = __builtin_fma(A1, h1, a11);
                }
            } else {                     // L[k][l] += sum_alpha h_{alpha + k}(d1) V[alpha][l]
                const double *h = hk + 2 * ti, *v = sA + 2 * tj;
#pragma unroll
                for (int a = 0; a < P; a++) {
                    const double h0 = h[a], h1 = h[a + 1], v0 = v[a * P], v1 = v[a * P + 1];
                    a00 = __builtin_fma(h0, v0, a00);
                    a01 = __builtin_fma(h0, v1, a01);
                    a10 = __builtin_fma(h1, v0, a10);
                    a11 = __builtin_fma(h1, v1, a11);
                }
            }
        }
        src = nxt;
        o = on;
#pragma unroll
        for (int u = 0; u < PER; u++) mine[u] = ahead[u];
    }
    const double acc[2][2] = {{a00, a01}, {a10, a11}};
    if (PASS == 0) {
        if (t == 0) vflag[c] = any ? 1 : 0;
        if (act && any) {
#pragma unroll
            for (int di = 0; di < 2; di++)
#pragma unroll
                for (int dj = 0; dj < 2; dj++)
                    V[(int64_t)c * PP + (2 * ti + di) * P + 2 * tj + dj] = acc[di][dj];
        }
    } else if (act) {
#pragma unroll
        for (int di = 0; di < 2; di++)
#pragma unroll
            for (int dj = 0; dj < 2; dj++) {
                const int r0 = 2 * ti + di, r1 = 2 * tj + dj;
                // D_k D_l = (-1)^(k+l) / (k! l!)
                double fk = 1.0, fl = 1.0;
                for (int i = 2; i <= r0; i++) fk *= (double)i;
                for (int i = 2; i <= r1; i++) fl *= (double)i;
                const double sgn = ((r0 + r1) & 1) ? -1.0 : 1.0;
                local[(int64_t)blockIdx.x * PP + r0 * P + r1] = sgn * acc[di][dj] / (fk * fl);
            }
    }
}

// The same two passes with FOUR consecutive targets of the convolution direction per workgroup and a 4 x 4
// register tile per thread (25 tiles x 4 targets = 100 of 128 threads): a source matrix is staged to LDS once
// for the up to four targets that use it (with four different Hankel rows), and a thread reads 5 LDS values per
// 16 multiply-adds (the Hankel window slides in registers) instead of 3 per 4.  Every element is summed in the
// order of the kernel above (offsets ascending = sources descending, inner index ascending): same bits.
//   pass 0   grid (nx, ceil(ny / 4)):  V[cy_C][cx] for cy_C = 4 q .. 4 q + 3
//   pass 1   grid (ceil(nx / 4), ny):  L[head of (cx_C, cy)] for cx_C = 4 q .. 4 q + 3 (cells with sources only)
constexpr int H2L4_T = 4;
template <int P, int PASS>
__global__ void __launch_bounds__(H2L_THREADS)
kde_h2l4_kernel(KdeGeom g, const int32_t *__restrict__ hslot, const int32_t *__restrict__ slot,
                const double *__restrict__ herm, const double *__restrict__ hankel, int reach,
                double *__restrict__ V, uint8_t *__restrict__ vflag, double *__restrict__ local, int n_heads) {
    constexpr int NH = 2 * P - 1, PT = (P + 3) / 4, PR = PT * 4, PP = P * P;
    constexpr int SR = PR + 1;                                  // row stride of the staged matrix: with 20 the four-row tiles of pass 0 fall on one bank
    constexpr int NHP = NH + 8;                                 // padded Hankel row (tiles beyond P read past NH)
    constexpr int PER = (PP + H2L_THREADS - 1) / H2L_THREADS;   // matrix elements a thread stages
    static_assert(PT * PT * H2L4_T <= H2L_THREADS, "tiles x targets must fit the workgroup");
    static_assert(P <= 24, "INV_FACT");
    __shared__ double sA[PR * SR];                              // [row][col], row stride SR, zero beyond P
    __shared__ double sH[(2 * H2L_MAX_REACH + 1) * NHP];
    const int t = threadIdx.x;
    const int tg = t / (PT * PT), tile = t % (PT * PT);
    const bool act = tg < H2L4_T;
    const int ti = tile / PT, tj = tile % PT;
    for (int i = t; i < (2 * reach + 1) * NHP; i += H2L_THREADS) {
        const int row = i / NHP, col = i % NHP;
        sH[i] = col < NH ? hankel[row * NH + col] : 0.0;
    }
    for (int i = t; i < PR * SR; i += H2L_THREADS) sA[i] = 0.0;
    const int nx = g.nc[0], ny = g.nc[1];
    // targets of this workgroup along the convolution direction: u0 .. u0 + 3; the fixed coordinate: w
    const int u0 = (PASS == 0 ? (int)blockIdx.y : (int)blockIdx.x) * H2L4_T;
    const int w = PASS == 0 ? (int)blockIdx.x : (int)blockIdx.y;
    const int nu = PASS == 0 ? ny : nx;
    const int my_u = u0 + tg;                                   // this thread's target
    bool my_target = act && my_u < nu;
    int my_head = -1;
    if (PASS == 1 && my_target) {
        my_head = hslot[(int64_t)w * nx + my_u];
        my_target = my_head >= 0;
    }
    if (PASS == 1 && !__syncthreads_or(my_target)) return;     // no cell with sources among the four
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = 0.0;
    bool any = false;
    // The source positions of a workgroup's targets are a dependent sequence (fetch -> barrier -> stage -> barrier -> P
    // rounds of multiply-adds, ~2.5 us each, 20 of them), and that sequence, not the arithmetic, is what a pass takes (VALU busy
    // 0.35 with every workgroup resident; one target per workgroup, four times the workgroups: the same time).  Round 5: the
    // positions are cut into gridDim.z contiguous parts, part z writes its own V / local (buffers [part][cell]); the
    // consumer adds the parts in part order as it fetches (pass 1: here; the local expansions: kde_local_pilot_kernel).
    const int n_part = (int)gridDim.z, part = (int)blockIdx.z;
    const int64_t n_cells_all = (int64_t)nx * ny;
    // source matrix at position v of the convolution direction, or nullptr (workgroup-uniform); pass 1: the parts of V of
    // that cell that hold anything (bit z of `have`)
    unsigned int have = 0;
    auto source = [&](int v) -> const double * {
        if (v < 0 || v >= nu) return nullptr;
        if (PASS == 0) {
            const int sl = slot[(int64_t)v * nx + w];           // source cell (cx = w, cy_B = v)
            return sl < 0 ? nullptr : herm + (int64_t)sl * PP;
        }
        const int64_t cb = (int64_t)w * nx + v;                 // V of (cx_B = v, cy = w)
        have = 0;
        for (int z = 0; z < n_part; z++) have |= vflag[z * n_cells_all + cb] ? (1u << z) : 0u;
        return have ? V + cb * PP : nullptr;
    };
    auto fetch = [&](const double *src, unsigned int parts, double (&buf)[PER]) {
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int e = t + u * H2L_THREADS;
            double val = 0.0;
            if (src && e < PP) {
                if (PASS == 0) val = src[e];
                else
                    for (int z = 0; z < n_part; z++)              // the parts of pass 0, added in part order
                        if (parts & (1u << z)) val += src[z * n_cells_all * PP + e];
            }
            buf[u] = val;
        }
    };
    // sources from the highest position down: for every target the offsets o = target - source ascend
    const int v_hi_all = u0 + H2L4_T - 1 + reach < nu - 1 ? u0 + H2L4_T - 1 + reach : nu - 1;
    const int v_lo_all = u0 - reach > 0 ? u0 - reach : 0;
    const int n_pos = v_hi_all - v_lo_all + 1;
    const int v_hi = v_hi_all - (part * n_pos) / n_part;
    const int v_lo = v_hi_all - ((part + 1) * n_pos) / n_part + 1;
    int v = v_hi;
    const double *src = v >= v_lo ? source(v) : nullptr;
    while (!src && v > v_lo) src = source(--v);
    double mine[PER], ahead[PER];
    fetch(src, have, mine);
    while (src) {
        int vn = v - 1;
        const double *nxt = vn >= v_lo ? source(vn) : nullptr;
        while (!nxt && vn > v_lo) nxt = source(--vn);
        fetch(nxt, have, ahead);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int e = t + u * H2L_THREADS;
            if (e < PP) sA[(e / P) * SR + e % P] = mine[u];
        }
        __syncthreads();
        const int o = my_u - v;
        if (my_target && o >= -reach && o <= reach) {
            any = true;
            const double *hk = sH + (o + reach) * NHP;
            if (PASS == 0) {             // W[alpha][l] += sum_beta A[alpha][beta] h_{beta + l}(d2)
                const double *r = sA + (4 * ti) * SR, *h = hk + 4 * tj;
                double h0 = h[0], h1 = h[1], h2 = h[2];
#pragma unroll
                for (int b = 0; b < P; b++) {
                    const double h3 = h[b + 3];
                    const double a0 = r[b], a1 = r[SR + b], a2 = r[2 * SR + b], a3 = r[3 * SR + b];
                    acc[0][0] = __builtin_fma(a0, h0, acc[0][0]); acc[0][1] = __builtin_fma(a0, h1, acc[0][1]);
                    acc[0][2] = __builtin_fma(a0, h2, acc[0][2]); acc[0][3] = __builtin_fma(a0, h3, acc[0][3]);
                    acc[1][0] = __builtin_fma(a1, h0, acc[1][0]); acc[1][1] = __builtin_fma(a1, h1, acc[1][1]);
                    acc[1][2] = __builtin_fma(a1, h2, acc[1][2]); acc[1][3] = __builtin_fma(a1, h3, acc[1][3]);
                    acc[2][0] = __builtin_fma(a2, h0, acc[2][0]); acc[2][1] = __builtin_fma(a2, h1, acc[2][1]);
                    acc[2][2] = __builtin_fma(a2, h2, acc[2][2]); acc[2][3] = __builtin_fma(a2, h3, acc[2][3]);
                    acc[3][0] = __builtin_fma(a3, h0, acc[3][0]); acc[3][1] = __builtin_fma(a3, h1, acc[3][1]);
                    acc[3][2] = __builtin_fma(a3, h2, acc[3][2]); acc[3][3] = __builtin_fma(a3, h3, acc[3][3]);
                    h0 = h1; h1 = h2; h2 = h3;
                }
            } else {                     // L[k][l] += sum_alpha h_{alpha + k}(d1) V[alpha][l]
                const double *h = hk + 4 * ti, *c = sA + 4 * tj;
                double h0 = h[0], h1 = h[1], h2 = h[2];
#pragma unroll
                for (int a = 0; a < P; a++) {
                    const double h3 = h[a + 3];
                    const double v0 = c[a * SR], v1 = c[a * SR + 1], v2 = c[a * SR + 2], v3 = c[a * SR + 3];
                    acc[0][0] = __builtin_fma(h0, v0, acc[0][0]); acc[0][1] = __builtin_fma(h0, v1, acc[0][1]);
                    acc[0][2] = __builtin_fma(h0, v2, acc[0][2]); acc[0][3] = __builtin_fma(h0, v3, acc[0][3]);
                    acc[1][0] = __builtin_fma(h1, v0, acc[1][0]); acc[1][1] = __builtin_fma(h1, v1, acc[1][1]);
                    acc[1][2] = __builtin_fma(h1, v2, acc[1][2]); acc[1][3] = __builtin_fma(h1, v3, acc[1][3]);
                    acc[2][0] = __builtin_fma(h2, v0, acc[2][0]); acc[2][1] = __builtin_fma(h2, v1, acc[2][1]);
                    acc[2][2] = __builtin_fma(h2, v2, acc[2][2]); acc[2][3] = __builtin_fma(h2, v3, acc[2][3]);
                    acc[3][0] = __builtin_fma(h3, v0, acc[3][0]); acc[3][1] = __builtin_fma(h3, v1, acc[3][1]);
                    acc[3][2] = __builtin_fma(h3, v2, acc[3][2]); acc[3][3] = __builtin_fma(h3, v3, acc[3][3]);
                    h0 = h1; h1 = h2; h2 = h3;
                }
            }
        }
        src = nxt;
        v = vn;
#pragma unroll
        for (int u = 0; u < PER; u++) mine[u] = ahead[u];
    }
    if (PASS == 0) {
        if (act && my_u < nu) {
            const int64_t c = part * n_cells_all + (int64_t)my_u * nx + w;
            if (tile == 0) vflag[c] = any ? 1 : 0;
            if (any) {
#pragma unroll
                for (int di = 0; di < 4; di++)
#pragma unroll
                    for (int dj = 0; dj < 4; dj++) {
                        const int r0 = 4 * ti + di, r1 = 4 * tj + dj;
                        if (r0 < P && r1 < P) V[c * PP + r0 * P + r1] = acc[di][dj];
                    }
            }
        }
    } else if (my_target) {
#pragma unroll
        for (int di = 0; di < 4; di++)
#pragma unroll
            for (int dj = 0; dj < 4; dj++) {
                const int r0 = 4 * ti + di, r1 = 4 * tj + dj;
                if (r0 < P && r1 < P) {
                    // D_k D_l = (-1)^(k+l) / (k! l!)
                    const double sgn = ((r0 + r1) & 1) ? -1.0 : 1.0;
                    local[((int64_t)part * n_heads + my_head) * PP + r0 * P + r1] = sgn * acc[di][dj] * INV_FACT[r0] * INV_FACT[r1];
                }
            }
    }
}

// The two passes on the matrix cores (round 5; series orders up to 16, i.e. the stage's cut-off).  A pass is a sum of
// products of 16 x 16 matrices -- W = sum_v A_v H(o_v) with the symmetric Hankel matrix H(o)[b][l] = h_{b+l}(o) --, which is
// what `v_mfma_f64_16x16x4_f64` computes: 1 024 multiply-adds per instruction from ONE double per lane and operand,
//     A operand  lane -> [row = lane & 15][k = lane >> 4],   B operand  lane -> [k = lane >> 4][col = lane & 15],
//     C / D      lane, register r -> [row = (lane >> 4) + 4 r][col = lane & 15]
// (four instructions per matrix product: k = 4 s + (lane >> 4)).  The vector form above spends five LDS reads per sixteen
// multiply-adds and two barriers per source position: VALU busy 0.35-0.4, ~50 us per pass and estimator.  Here a workgroup
// has H2LM_T consecutive targets of the convolution direction, its wavefronts every FGT_WAVES-th source position: a source
// matrix comes straight from global memory into the operand registers (pass 0: A_v as the A operand -- read from the
// TRANSPOSED array kde_hermite_coef_mfma_kernel writes --, pass 1: V_v as the B operand; zero beyond a series order of 14), the Hankel entry of a lane -- h_{(lane & 15) + (lane >> 4) + 4 s}(o), the same for both passes --
// from a table in LDS, no barrier in the loop, absent sources and out-of-reach offsets skipped wavefront-uniformly.
// Sources descend as above (offsets ascend per target); inside a product the matrix cores' own order: the results differ
// from the vector form's by rounding (1e-16 relative), and are the same from run to run.
constexpr int H2LM_T = 2;
template <int PASS>
__global__ void __launch_bounds__(64 * FGT_WAVES)
kde_h2l_mfma_kernel(KdeGeom g, const int32_t *__restrict__ hslot, const int32_t *__restrict__ slot,
                    const double *__restrict__ herm, const double *__restrict__ hankel, int reach, int P,
                    double *__restrict__ V, uint8_t *__restrict__ vflag, double *__restrict__ local) {
    constexpr int NHP = 32;                                      // padded Hankel row: indices up to 15 + 15
    __shared__ double sH[(2 * H2L_MAX_REACH + 1) * NHP];
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const int NH = 2 * P - 1, PP = P * P;
    for (int i = (int)threadIdx.x; i < (2 * reach + 1) * NHP; i += 64 * FGT_WAVES) {
        const int row = i / NHP, col = i % NHP;
        sH[i] = col < NH ? hankel[row * NH + col] : 0.0;
    }
    __syncthreads();
    const int nx = g.nc[0], ny = g.nc[1];
    // the FGT_WAVES wavefronts of a workgroup share the source positions of ONE group of targets (every FGT_WAVES-th
    // position each, their sums added in wavefront order through LDS at the end): a wavefront's time is a chain of memory
    // latencies, one per position it multiplies, and that chain is what the launch takes
    __shared__ double red[FGT_WAVES][H2LM_T][256];
    __shared__ int any_w[FGT_WAVES][H2LM_T];
    const int u0 = (PASS == 0 ? (int)blockIdx.y : (int)blockIdx.x) * H2LM_T;
    const int w = PASS == 0 ? (int)blockIdx.x : (int)blockIdx.y;
    const int nu = PASS == 0 ? ny : nx;
    bool tgt[H2LM_T], any[H2LM_T];
    int head[H2LM_T];
    bool some = false;
#pragma unroll
    for (int t = 0; t < H2LM_T; t++) {
        tgt[t] = u0 + t < nu;
        head[t] = -1;
        any[t] = false;
        if (PASS == 1 && tgt[t]) {
            head[t] = hslot[(int64_t)w * nx + u0 + t];
            tgt[t] = head[t] >= 0;
        }
        some = some || tgt[t];
    }
    if (!some) return;                                           // (pass 1: no cell with sources among the targets)
    mfma_d4 acc[H2LM_T];
#pragma unroll
    for (int t = 0; t < H2LM_T; t++) acc[t] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
    const int li = lane & 15, lk = lane >> 4;
    const int v_hi = u0 + H2LM_T - 1 + reach < nu - 1 ? u0 + H2LM_T - 1 + reach : nu - 1;
    const int v_lo = u0 - reach > 0 ? u0 - reach : 0;
    // this lane's element of k-step s of a source matrix: pass 0 A[li][4 s + lk] (A operand; the array holds A transposed),
    // pass 1 V[4 s + lk][li] (B operand): the same offset
    int off[4];
    bool inside[4];
#pragma unroll
    for (int sx = 0; sx < 4; sx++) {
        const int kk = 4 * sx + lk;
        inside[sx] = li < P && kk < P;
        off[sx] = inside[sx] ? kk * P + li : 0;                  // (pass 0 reads the coefficients TRANSPOSED: kde_hermite_coef_mfma_kernel)
    }
    // Which of the <= 2 reach + H2LM_T source positions hold a matrix: looked up by the lanes at once (lane j: position
    // v_lo + j) -- one position per loop iteration, each a dependent global load in front of the matrix's own loads, was
    // what a wavefront's time consisted of (two memory latencies per position against 0.2 us of matrix instructions).
    const int n_pos = v_hi - v_lo + 1;                           // <= 2 * H2L_MAX_REACH + H2LM_T < 64
    int my_slot = -1;
    if (lane < n_pos) {
        if (PASS == 0) my_slot = slot[(int64_t)(v_lo + lane) * nx + w];                  // source cell (cx = w, cy_B = v)
        else my_slot = vflag[(int64_t)w * nx + v_lo + lane] ? 0 : -1;                    // V of (cx_B = v, cy = w)
    }
    unsigned long long present = __builtin_amdgcn_ballot_w64(my_slot >= 0);
    {   // this wavefront's share: ranks wave, wave + FGT_WAVES, ... of the positions that hold a matrix, from the highest down
        unsigned long long all = present, keep = 0;
        for (int rank = 0; all; rank++) {
            const int j = 63 - __builtin_clzll(all);
            all &= ~(1ull << j);
            if (rank % FGT_WAVES == wave) keep |= 1ull << j;
        }
        present = keep;
    }
    auto next_pos = [&]() -> int {                               // highest position first: offsets ascend per target
        if (!present) return -1;
        const int j = 63 - __builtin_clzll(present);
        present &= ~(1ull << j);
        return j;
    };
    // (unconditional loads -- element 0 for a lane outside the matrix, the first matrix of the array behind the last
    //  source --, so that the compiler can count them and wait for the current matrix only, not for those requested ahead)
    auto fetch = [&](int j, double (&m)[4]) {
        const double *from = PASS == 0 ? herm : V;
        if (j >= 0)
            from += PASS == 0 ? (int64_t)__builtin_amdgcn_readlane(my_slot, j) * PP : ((int64_t)w * nx + v_lo + j) * PP;
#pragma unroll
        for (int sx = 0; sx < 4; sx++) m[sx] = from[off[sx]];
    };
    auto multiply = [&](int j, const double (&m)[4]) {
        const int v = v_lo + j;
#pragma unroll
        for (int t = 0; t < H2LM_T; t++) {
            const int o = u0 + t - v;
            if (!tgt[t] || o < -reach || o > reach) continue;    // wavefront-uniform
            any[t] = true;
            const double *hk = sH + (o + reach) * NHP + li + lk;
#pragma unroll
            for (int sx = 0; sx < 4; sx++) {
                const double h = hk[4 * sx], x = inside[sx] ? m[sx] : 0.0;
                acc[t] = PASS == 0 ? __builtin_amdgcn_mfma_f64_16x16x4f64(x, h, acc[t], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f64_16x16x4f64(h, x, acc[t], 0, 0, 0);
            }
        }
    };
    // two matrices requested ahead of the one being multiplied, in three register sets taken in turn (a copy from set to
    // set would wait for the matrix just requested)
    int ja = next_pos(), jb = next_pos(), jc = -1;
    double ma[4], mb[4], mc[4];
    fetch(ja, ma);
    fetch(jb, mb);
    for (;;) {
        if (ja < 0) break;
        jc = next_pos(); fetch(jc, mc); multiply(ja, ma);
        if (jb < 0) break;
        ja = next_pos(); fetch(ja, ma); multiply(jb, mb);
        if (jc < 0) break;
        jb = next_pos(); fetch(jb, mb); multiply(jc, mc);
    }
    // the wavefronts' sums, added in wavefront order; wavefront r of the workgroup finishes register r (rows lk + 4 r)
#pragma unroll
    for (int t = 0; t < H2LM_T; t++) {
#pragma unroll
        for (int r = 0; r < 4; r++) red[wave][t][r * 64 + lane] = acc[t][r];
        if (lane == 0) any_w[wave][t] = any[t] ? 1 : 0;
    }
    __syncthreads();
    static_assert(FGT_WAVES == 4, "one accumulator register per wavefront in the final sum");
#pragma unroll
    for (int t = 0; t < H2LM_T; t++) {
        bool got = false;
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < FGT_WAVES; q++) {
            got = got || any_w[q][t];
            v += red[q][t][wave * 64 + lane];
        }
        const int row = lk + 4 * wave;
        if (PASS == 0) {
            if (u0 + t >= nu) continue;
            const int64_t c = (int64_t)(u0 + t) * nx + w;
            if (threadIdx.x == 0) vflag[c] = got ? 1 : 0;
            if (got && row < P && li < P) V[c * PP + row * P + li] = v;
        } else {
            if (!tgt[t]) continue;
            if (row < P && li < P) {
                // D_k D_l = (-1)^(k+l) / (k! l!)
                const double sgn = ((row + li) & 1) ? -1.0 : 1.0;
                local[(int64_t)head[t] * PP + row * P + li] = sgn * v * INV_FACT[row] * INV_FACT[li];
            }
        }
    }
}

// pilot densities at the (cell-sorted) sources: the local expansion of the target's cell for all
// dense cells in range + direct sums over the sparse ones.
template <int P, bool SPARSE>
__global__ void __launch_bounds__(KDE_THREADS)
kde_local_pilot_kernel(KdeGeom g, const KdeBlock *__restrict__ blocks, const double *__restrict__ sy,
                       int64_t n_src, const double *__restrict__ coef,
                       const int32_t *__restrict__ cell_start, const int32_t *__restrict__ slot,
                       const double *__restrict__ local, int n_heads, int n_part, double *__restrict__ out,
                       unsigned long long *__restrict__ pair_count) {
    constexpr int W = 4;
    __shared__ double t_src[SRC_TILE * W];
    const KdeBlock b = blocks[blockIdx.x];
    double q[Q_PER_THREAD][2], acc[Q_PER_THREAD];
#pragma unroll
    for (int u = 0; u < Q_PER_THREAD; u++) {
        const int jq = threadIdx.x + u * KDE_THREADS;
        const int64_t j = b.q_begin + (jq < b.q_count ? jq : 0);
        q[u][0] = sy[j];
        q[u][1] = sy[n_src + j];
        acc[u] = 0.0;
    }
    unsigned long long work = (unsigned long long)(P * P / 23 + 1) * b.q_count;
    __shared__ double sL[P * P];   // the cell's local expansion: fetched once by the workgroup, read back as broadcasts
    for (int i = threadIdx.x; i < P * P; i += KDE_THREADS) {
        double v = local[(int64_t)b.head * (P * P) + i];
        for (int z = 1; z < n_part; z++) v += local[((int64_t)z * n_heads + b.head) * (P * P) + i];   // the parts of the translation passes, in part order
        sL[i] = v;
    }
    __syncthreads();
    {
        const double *L = sL;
        const double c1 = g.ylo[0] + (b.c0[0] + 0.5) * g.cell, c2 = g.ylo[1] + (b.c0[1] + 0.5) * g.cell;
        double x[Q_PER_THREAD], y[Q_PER_THREAD], sum[Q_PER_THREAD];
#pragma unroll
        for (int u = 0; u < Q_PER_THREAD; u++) {
            x[u] = (q[u][0] - c1) * RSQRT2;
            y[u] = (q[u][1] - c2) * RSQRT2;
            sum[u] = 0.0;
        }
#pragma unroll 1
        for (int k = P - 1; k >= 0; k--) {
            const double *row = L + k * P;
            double inner[Q_PER_THREAD];
#pragma unroll
            for (int u = 0; u < Q_PER_THREAD; u++) inner[u] = 0.0;
#pragma unroll
            for (int l = P - 1; l >= 0; l--) {
                const double a = row[l];
#pragma unroll
                for (int u = 0; u < Q_PER_THREAD; u++) inner[u] = __builtin_fma(inner[u], y[u], a);
            }
#pragma unroll
            for (int u = 0; u < Q_PER_THREAD; u++) sum[u] = __builtin_fma(sum[u], x[u], inner[u]);
        }
#pragma unroll
        for (int u = 0; u < Q_PER_THREAD; u++) acc[u] = sum[u];
    }
    // sparse cells in range: directly (SPARSE = false: every non-empty cell has a series, nothing to look for)
    int lo[2], hi[2];
    const double reach = ceil(sqrt(g.rcut2) * g.inv_cell);
#pragma unroll
    for (int d = 0; d < 2; d++) {
        const double l = (double)b.c0[d] - reach, h = (double)b.c1[d] + reach;
        lo[d] = l > 0.0 ? (int)l : 0;
        hi[d] = h < (double)(g.nc[d] - 1) ? (int)h : g.nc[d] - 1;
    }
    if (!SPARSE) hi[1] = lo[1] - 1;
    for (int cy = lo[1]; cy <= hi[1]; cy++) {
        const int gapy = cy < b.c0[1] ? b.c0[1] - cy - 1 : (cy > b.c1[1] ? cy - b.c1[1] - 1 : 0);
        const double gy = gapy * g.cell;
        const int64_t row = (int64_t)cy * g.nc[0];
        for (int cx = lo[0]; cx <= hi[0]; cx++) {
            const int64_t c = row + cx;
            const int begin = cell_start[c], end = cell_start[c + 1];
            if (end == begin || slot[c] >= 0) continue;
            const int gapx = cx < b.c0[0] ? b.c0[0] - cx - 1 : (cx > b.c1[0] ? cx - b.c1[0] - 1 : 0);
            const double gx = gapx * g.cell;
            if (gx * gx + gy * gy > g.rcut2) continue;
            work += (unsigned long long)(end - begin) * b.q_count;
            for (int base = begin; base < end; base += SRC_TILE) {
                const int cnt = end - base < SRC_TILE ? end - base : SRC_TILE;
                __syncthreads();
                for (int k = threadIdx.x; k < cnt; k += KDE_THREADS) {
                    t_src[k * W + 0] = sy[base + k];
                    t_src[k * W + 1] = sy[n_src + base + k];
                    t_src[k * W + 2] = coef[base + k];
                }
                __syncthreads();
                for (int k = 0; k < cnt; k++) {
                    const double *s = t_src + k * W;
#pragma unroll
                    for (int u = 0; u < Q_PER_THREAD; u++) {
                        const double d0 = q[u][0] - s[0], d1 = q[u][1] - s[1];
                        const double r2 = __builtin_fma(d1, d1, d0 * d0);
                        acc[u] = __builtin_fma(s[2], exp_nonpos(-0.5 * r2), acc[u]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < Q_PER_THREAD; u++) {
        const int jq = threadIdx.x + u * KDE_THREADS;
        if (jq < b.q_count) out[b.q_begin + jq] = acc[u];
    }
    if (pair_count && threadIdx.x == 0 && work) atomicAdd(pair_count, work);
}

// The local expansions at the sources when EVERY non-empty cell has a series (the default: nothing is summed directly):
// one wavefront per 64 targets of a block (<= 512 sources of one cell), the cell's P x P coefficients as
// LDS broadcasts.  The kernel above gives a block 256 threads with two targets each: a cell of the C3 estimators holds
// ~200 sources, so most lanes had one target or none and still ran both Horner schemes (42 us per estimator for 0.2 GFLOP).
// Same nested Horner scheme per target, same bits.
template <int P>
__global__ void __launch_bounds__(64 * FGT_WAVES)
kde_local_pilot_wave_kernel(KdeGeom g, const KdeBlock *__restrict__ blocks, const double *__restrict__ sy, int64_t n_src,
                            const double *__restrict__ local, double *__restrict__ out,
                            unsigned long long *__restrict__ stamps) {
    const unsigned long long t_start = stamps ? wall_clock64() : 0ull;
    // grid (blocks, Q_CHUNK / (64 FGT_WAVES)): a wavefront takes ONE round of 64 targets of its block (a block of 512
    // targets worked through by one wavefront was the launch's critical path), the cell's coefficients fetched once by the
    // workgroup (one element per thread: ONE memory round trip) and read back from LDS as broadcasts
    __shared__ double sL[P * P];
    const int bi = (int)blockIdx.x;
    const int round = (int)blockIdx.y * FGT_WAVES + ((int)threadIdx.x >> 6);
    const KdeBlock b = blocks[bi];
    const int q_count = b.q_count, q_begin = b.q_begin;
    if ((int)blockIdx.y * FGT_WAVES * 64 >= q_count) return;     // (workgroup-uniform)
    for (int i = (int)threadIdx.x; i < P * P; i += 64 * FGT_WAVES) sL[i] = local[(int64_t)b.head * (P * P) + i];
    const double c1 = g.ylo[0] + (b.c0[0] + 0.5) * g.cell, c2 = g.ylo[1] + (b.c0[1] + 0.5) * g.cell;
    const int jq = round * 64 + ((int)threadIdx.x & 63);
    const int64_t j = q_begin + (jq < q_count ? jq : 0);
    const double x = (sy[j] - c1) * RSQRT2, y = (sy[n_src + j] - c2) * RSQRT2;
    __syncthreads();
    if (round * 64 >= q_count) return;
    const unsigned long long t_mid = stamps ? wall_clock64() : 0ull;
    const double *L = sL;
    double sum = 0.0;
#pragma unroll 4
    for (int k = P - 1; k >= 0; k--) {
        const double *row = L + k * P;
        double inner = 0.0;
#pragma unroll
        for (int l = P - 1; l >= 0; l--) inner = __builtin_fma(inner, y, row[l]);
        sum = __builtin_fma(sum, x, inner);
    }
    if (jq < q_count) out[q_begin + jq] = sum;
    // (the work count of this form -- (P^2 / 23 + 1) per target -- is added on the host: one atomic per block on ONE
    //  address took 17 ns each, one after the other across the eight dies: 22 of the launch's 27 us, stamps of round 5)
    if (stamps && (threadIdx.x & 63) == 0) {   // development: when this wavefront started, had its inputs, ended
        unsigned long long *st = stamps + 4 * (((int64_t)bi * gridDim.y + blockIdx.y) * FGT_WAVES + (threadIdx.x >> 6));
        st[0] = t_start; st[1] = t_mid; st[2] = wall_clock64();
        st[3] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned long long)__builtin_amdgcn_s_getreg(63492);   // XCC_ID, HW_ID
    }
}

// ------------------------------------------------------------------ host side
struct Arena {   // carves the caller's workspace
    char *base;
    size_t size, used;
    bool ok;
    Arena(void *p, size_t n) : base((char *)p), size(n), used(0), ok(true) {}
    template <class T> T *take(size_t count) {
        size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
        if (used + bytes > size) { ok = false; return nullptr; }
        T *r = (T *)(base + used);
        used += bytes;
        return r;
    }
};

// radix passes always (rocprim's default switches to a merge sort below 2^20 items)
using FlatSortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;

static size_t sort_temp_bytes(int64_t n) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (uint64_t *)nullptr, (uint64_t *)nullptr,
                                       (uint32_t *)nullptr, (uint32_t *)nullptr, (int)n, 0, 63, 0);
    size_t b3 = 0;
    (void)rocprim::radix_sort_pairs<FlatSortConfig>(nullptr, b3, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                                    (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)n, 0u, 32u);
    bytes = std::max(bytes, b3);
    size_t b4 = 0;
    (void)rocprim::radix_sort_keys<FlatSortConfig>(nullptr, b4, (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)n, 0u, 32u);
    bytes = std::max(bytes, b4);
    size_t b2 = 0;
    (void)hipcub::DeviceSelect::Flagged(nullptr, b2, hipcub::CountingInputIterator<int32_t>(0),
                                  (uint8_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, (int)n, 0);
    return std::max(bytes, b2) + 256;
}

}  // namespace pisa

using namespace pisa;

static int g_kde_expansion = 2;   // 2-D pilot: 0 direct sums, 1 Hermite series per target, 2 + local expansions

struct pisa_hip_kde {
    int dim;
    int64_t n;
    int adaptive;
    double alpha, tol, factor, norm, sum_w, r_cut;
    double mean[3], cov[9], inv_cov[9];
    KdeGeom g;
    int64_t n_cells;
    // device (inside the caller's workspace)
    double *ys, *wn, *coef, *s2, *cell_s2min, *scalars;   // scalars: [0] log-sum, [1] min s2, [2] max s2
    double s2_range[2];   // host copy of scalars[1..2]
    int cell_s2min_valid;
    int32_t *cell_start;
    unsigned long long *pair_count;
    unsigned long long pairs_pilot, pairs_eval;
    int n_dense;   // cells summed through their Hermite series in the pilot estimate
};

namespace pisa {

static int query_blocks(const uint64_t *d_keys, int64_t m, int tile, int chunk, char *d_temp, size_t temp_bytes,
                        uint8_t *d_flags, int32_t *d_starts, int32_t *d_count, uint64_t *d_head_keys,
                        std::vector<KdeBlock> &blocks, hipStream_t s,
                        std::vector<int32_t> *starts_out = nullptr, std::vector<uint64_t> *keys_out = nullptr,
                        int64_t max_heads = 0) {
    hipLaunchKernelGGL(kde_heads_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, d_keys, m, d_flags);
    PISA_CHECK_LAUNCH("kde_heads_kernel");
    PISA_TRY_HIP(hipcub::DeviceSelect::Flagged(d_temp, temp_bytes, hipcub::CountingInputIterator<int32_t>(0),
                                               d_flags, d_starts, d_count, (int)m, s));
    int32_t n_heads = 0;
    std::vector<int32_t> starts;
    std::vector<uint64_t> hk;
    PISA_TRY_HIP(hipMemcpyAsync(&n_heads, d_count, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (max_heads > 0 && max_heads <= 65536) {
        // the caller bounds the number of tiles (cells of the grid): the kernel takes the count from the
        // device and everything comes back behind ONE synchronisation
        const int64_t cap = std::min<int64_t>(max_heads, m);
        hipLaunchKernelGGL(kde_head_keys_kernel, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, s,
                           d_keys, d_starts, d_count, d_head_keys);
        PISA_CHECK_LAUNCH("kde_head_keys_kernel");
        starts.resize(cap);
        hk.resize(cap);
        PISA_TRY_HIP(hipMemcpyAsync(starts.data(), d_starts, cap * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        PISA_TRY_HIP(hipMemcpyAsync(hk.data(), d_head_keys, cap * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        PISA_TRY_HIP(hipStreamSynchronize(s));
        if (n_heads <= 0 || n_heads > cap) return PISA_HIP_ERR_INVALID;
        starts.resize(n_heads);
        hk.resize(n_heads);
    } else {
        PISA_TRY_HIP(hipStreamSynchronize(s));
        if (n_heads <= 0) return PISA_HIP_ERR_INVALID;
        hipLaunchKernelGGL(kde_head_keys_kernel, dim3((unsigned)((n_heads + 255) / 256)), dim3(256), 0, s,
                           d_keys, d_starts, d_count, d_head_keys);
        PISA_CHECK_LAUNCH("kde_head_keys_kernel");
        starts.resize(n_heads);
        hk.resize(n_heads);
        PISA_TRY_HIP(hipMemcpyAsync(starts.data(), d_starts, n_heads * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        PISA_TRY_HIP(hipMemcpyAsync(hk.data(), d_head_keys, n_heads * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        PISA_TRY_HIP(hipStreamSynchronize(s));
    }
    blocks.clear();
    for (int32_t h = 0; h < n_heads; h++) {
        const int64_t begin = starts[h], end = h + 1 < n_heads ? starts[h + 1] : m;
        KdeBlock b;
        b.head = h;
        for (int d = 0; d < 3; d++) {
            const int64_t t = (int64_t)((hk[h] >> (21 * d)) & 0x1FFFFF) - KEY_OFF;
            b.c0[d] = (int32_t)(t * tile);
            b.c1[d] = (int32_t)(t * tile + tile - 1);
        }
        // equal shares: a tile of 300 queries becomes 150 + 150, not 256 + 44
        const int64_t parts = (end - begin + chunk - 1) / chunk;
        for (int64_t p = 0; p < parts; p++) {
            const int64_t q0 = begin + (end - begin) * p / parts, q1 = begin + (end - begin) * (p + 1) / parts;
            b.q_begin = (int32_t)q0;
            b.q_count = (int32_t)(q1 - q0);
            blocks.push_back(b);
        }
    }
    if (starts_out) starts_out->swap(starts);
    if (keys_out) keys_out->swap(hk);
    return PISA_HIP_OK;
}

template <bool VAR_BW, int QPT>
static int launch_pairs(const pisa_hip_kde *k, const KdeBlock *d_blocks, int n_blocks, int n_split,
                        const double *qy, int64_t m, double *partial, hipStream_t s) {
    dim3 grid((unsigned)n_blocks, (unsigned)n_split), block(KDE_THREADS);
#define KDE_PAIRS(DD) hipLaunchKernelGGL((kde_pairs_kernel<DD, VAR_BW, QPT>), grid, block, 0, s, k->g, d_blocks, qy, m, k->ys, k->n, k->coef, k->s2, k->cell_start, k->cell_s2min, k->scalars + 1, n_split, partial, k->pair_count)
    if (k->dim == 1) KDE_PAIRS(1);
    else if (k->dim == 2) KDE_PAIRS(2);
    else KDE_PAIRS(3);
#undef KDE_PAIRS
    PISA_CHECK_LAUNCH("kde_pairs_kernel");
    return PISA_HIP_OK;
}

static int pick_split(int n_blocks) {
    int n_split = 1;
    if (n_blocks < 1024) n_split = std::min(32, (1024 + n_blocks - 1) / n_blocks);
    return n_split;
}

// bytes of the create-time workspace that stay in use for the lifetime of the estimator
static size_t resident_bytes(int dim, int64_t n, int64_t n_cells) {
    auto r = [](size_t b) { return (b + 255) & ~(size_t)255; };
    return r((size_t)dim * n * 8) + 3 * r((size_t)n * 8) + r((size_t)(n_cells + 1) * 4) +
           r((size_t)n_cells * 8) + r(64) + r(64);
}

}  // namespace pisa

// split partial sums exist only below 1024 workgroups, i.e. below 2^19 queries
static size_t split_bytes(int64_t m) { return (size_t)32 * 8 * (size_t)std::min<int64_t>(m, 1 << 19); }

PISA_API int64_t pisa_hip_kde_workspace_bytes(int32_t dim, int64_t n_src) {
    if (dim < 1 || dim > 3 || n_src < 1 || n_src > 0x7FFFFFF0LL) return -1;
    const size_t n = (size_t)n_src;
    size_t total = resident_bytes(dim, n_src, cells_cap(n_src)) + (size_t)RED_BLOCKS * 16 * 8;
    total += 4 * n * 8 + n * 8 + 2 * n * 4;              // (y, weight) records, flat keys x2, idx x2
    total += n + n * 4 + n * 8;                          // flags, starts, head keys
    total += n * 8 + split_bytes(n_src);                 // pilot, split partials
    total += sort_temp_bytes(n_src) + (n / Q_CHUNK + (size_t)cells_cap(n_src)) * sizeof(KdeBlock);
    if (dim == 2)   // cell -> slot map, lists of dense / non-empty cells (the coefficients live in library scratch)
        total += (n / hermite_min() + 1) * 4 + (size_t)cells_cap(n_src) * 12 + 4096;   // dense list: one entry per cell at most
    return (int64_t)(total + 64 * 256);
}

PISA_API int pisa_hip_kde_release_scratch(void) {
    // frees the calling host thread's grow-only scratch (per thread: call it from the thread that built
    // / evaluated the estimators, after its stream has drained)
    if (g_kde_scratch) {
        if (g_kde_scratch_used) (void)hipStreamSynchronize(g_kde_scratch_stream);
        (void)hipFree(g_kde_scratch);
    }
    g_kde_scratch = nullptr;
    g_kde_scratch_bytes = 0;
    g_kde_scratch_used = false;
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_kde_configure(int32_t use_expansion) {
    const int old = g_kde_expansion;
    if (use_expansion >= 0) g_kde_expansion = use_expansion > 2 ? 2 : use_expansion;
    return old;
}

PISA_API int pisa_hip_kde_destroy(pisa_hip_kde *k) {
    if (k) free(k);
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_kde_create(int32_t dim, const double *d_x, const double *d_w, int64_t n,
                                 int32_t bw_method, int32_t adaptive, double alpha, double tol,
                                 void *d_work, int64_t work_bytes, pisa_hip_kde **out, void *stream) {
    if (!out) return PISA_HIP_ERR_INVALID;
    *out = nullptr;
    if (dim < 1 || dim > 3 || n < 2 || n > 0x7FFFFFF0LL || !d_x || !d_work || work_bytes <= 0 ||
        (bw_method != 0 && bw_method != 1) || !(tol >= 0.0) || tol >= 1.0)
        return PISA_HIP_ERR_INVALID;
    hipStream_t s = as_stream(stream);
    Arena ar(d_work, (size_t)work_bytes);
    pisa_hip_kde *k = (pisa_hip_kde *)calloc(1, sizeof(pisa_hip_kde));
    if (!k) return PISA_HIP_ERR_NOMEM;
    k->dim = dim; k->n = n; k->adaptive = adaptive; k->alpha = alpha; k->tol = tol;
#define KDE_FAIL(rc) do { free(k); return (rc); } while (0)
#define KDE_TRY(expr) do { int _rc = (expr); if (_rc != PISA_HIP_OK) KDE_FAIL(_rc); } while (0)
#define KDE_TRY_HIP(expr) do { int _rc = ::pisa::check_hip((expr), #expr); if (_rc != PISA_HIP_OK) KDE_FAIL(_rc); } while (0)
    // ---- resident part of the workspace (a first guess of the cell count is fixed up below)
    double *partial = ar.take<double>((size_t)RED_BLOCKS * 16);
    (void)ar.take<double>(16);   // (was the device-side join of the moment partials: the workspace layout is unchanged)
    if (!ar.ok) KDE_FAIL(PISA_HIP_ERR_NOMEM);
    const unsigned nb = (unsigned)((n + 255) / 256);
    // ---- moments
    double h1[10], h2[7];
#define KDE_D(KERNEL, ...) do { if (dim == 1) hipLaunchKernelGGL(KERNEL<1>, __VA_ARGS__); else if (dim == 2) hipLaunchKernelGGL(KERNEL<2>, __VA_ARGS__); else hipLaunchKernelGGL(KERNEL<3>, __VA_ARGS__); } while (0)
    // (the per-workgroup partial results come back and are joined here, in kde_final_reduce_kernel's order: one
    //  launch less per pass on a chain that is launch-latency bound for small samples)
    double hp[RED_BLOCKS * 10];
    KDE_D(kde_moments1_kernel, dim3(RED_BLOCKS), dim3(RED_THREADS), 0, s, d_x, d_w, n, partial);
    KDE_TRY_HIP(hipMemcpyAsync(hp, partial, (size_t)RED_BLOCKS * (1 + 3 * dim) * sizeof(double), hipMemcpyDeviceToHost, s));
    KDE_TRY_HIP(hipStreamSynchronize(s));
    final_reduce_host(hp, 1 + 3 * dim, 1 + dim, dim, h1);
    const double sw = h1[0];
    if (!(sw > 0.0) || !std::isfinite(sw)) KDE_FAIL(PISA_HIP_ERR_INVALID);
    double xmin[3] = {0, 0, 0}, xmax[3] = {0, 0, 0};
    for (int d = 0; d < dim; d++) {
        k->mean[d] = h1[1 + d] / sw;
        xmin[d] = h1[1 + dim + d];
        xmax[d] = h1[1 + 2 * dim + d];
    }
    KDE_D(kde_moments2_kernel, dim3(RED_BLOCKS), dim3(RED_THREADS), 0, s, d_x, d_w, n, k->mean[0],
          k->mean[1], k->mean[2], partial);
    KDE_TRY_HIP(hipMemcpyAsync(hp, partial, (size_t)RED_BLOCKS * 7 * sizeof(double), hipMemcpyDeviceToHost, s));
    KDE_TRY_HIP(hipStreamSynchronize(s));
    final_reduce_host(hp, 7, 7, 0, h2);
    // ---- bandwidth matrix (unbiased weighted covariance x factor^2), its inverse, whitening
    k->sum_w = sw;
    const double denom = 1.0 - h2[0] / (sw * sw);
    double cov[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    {
        int idx = 1;
        for (int d = 0; d < dim; d++)
            for (int e = d; e < dim; e++) {
                cov[d][e] = cov[e][d] = h2[idx] / sw / denom;
                idx++;
            }
    }
    k->factor = bw_method == 0 ? pow((double)n * (dim + 2) / 4.0, -1.0 / (dim + 4))   // silverman
                               : pow((double)n, -1.0 / (dim + 4));                      // scott
    double H[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int d = 0; d < dim; d++)
        for (int e = 0; e < dim; e++) H[d][e] = cov[d][e] * k->factor * k->factor;
    const double det = H[0][0] * (H[1][1] * H[2][2] - H[1][2] * H[2][1]) -
                       H[0][1] * (H[1][0] * H[2][2] - H[1][2] * H[2][0]) +
                       H[0][2] * (H[1][0] * H[2][1] - H[1][1] * H[2][0]);
    if (!(det > 0.0) || !std::isfinite(det)) KDE_FAIL(PISA_HIP_ERR_INVALID);
    double inv[3][3];
    inv[0][0] = (H[1][1] * H[2][2] - H[1][2] * H[2][1]) / det;
    inv[0][1] = (H[0][2] * H[2][1] - H[0][1] * H[2][2]) / det;
    inv[0][2] = (H[0][1] * H[1][2] - H[0][2] * H[1][1]) / det;
    inv[1][0] = (H[1][2] * H[2][0] - H[1][0] * H[2][2]) / det;
    inv[1][1] = (H[0][0] * H[2][2] - H[0][2] * H[2][0]) / det;
    inv[1][2] = (H[0][2] * H[1][0] - H[0][0] * H[1][2]) / det;
    inv[2][0] = (H[1][0] * H[2][1] - H[1][1] * H[2][0]) / det;
    inv[2][1] = (H[0][1] * H[2][0] - H[0][0] * H[2][1]) / det;
    inv[2][2] = (H[0][0] * H[1][1] - H[0][1] * H[1][0]) / det;
    k->norm = sqrt(pow(2.0 * M_PI, dim) * det);
    for (int d = 0; d < 3; d++)
        for (int e = 0; e < 3; e++) {
            k->cov[d * 3 + e] = (d < dim && e < dim) ? H[d][e] : 0.0;
            k->inv_cov[d * 3 + e] = (d < dim && e < dim) ? inv[d][e] : 0.0;
        }
    // Cholesky inv = L L^T, U = L^T  =>  |U v|^2 = v^T inv v
    double L[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int i = 0; i < dim; i++)
        for (int j = 0; j <= i; j++) {
            double sum = inv[i][j];
            for (int p = 0; p < j; p++) sum -= L[i][p] * L[j][p];
            if (i == j) {
                if (!(sum > 0.0)) KDE_FAIL(PISA_HIP_ERR_INVALID);
                L[i][i] = sqrt(sum);
            } else L[i][j] = sum / L[j][j];
        }
    KdeGeom &g = k->g;
    memset(&g, 0, sizeof(g));
    g.dim = dim;
    for (int d = 0; d < 3; d++) {
        g.mean[d] = k->mean[d];
        for (int e = 0; e < 3; e++) g.U[d * 3 + e] = (d < dim && e < dim) ? L[e][d] : 0.0;
    }
    // ---- cell grid over the whitened bounding box (image of the corners of the x box)
    double ylo[3] = {0, 0, 0}, yhi[3] = {0, 0, 0};
    for (int d = 0; d < dim; d++) { ylo[d] = INFINITY; yhi[d] = -INFINITY; }
    for (int corner = 0; corner < (1 << dim); corner++) {
        double xc[3] = {0, 0, 0};
        for (int d = 0; d < dim; d++) xc[d] = ((corner >> d) & 1 ? xmax[d] : xmin[d]) - k->mean[d];
        for (int d = 0; d < dim; d++) {
            double a = 0.0;
            for (int e = d; e < dim; e++) a += g.U[d * 3 + e] * xc[e];
            ylo[d] = std::min(ylo[d], a);
            yhi[d] = std::max(yhi[d], a);
        }
    }
    const bool cut = tol > 0.0;
    g.rcut2 = cut ? 2.0 * log(1.0 / tol) : 0.0;
    k->r_cut = cut ? sqrt(g.rcut2) : INFINITY;
    double extent = 0.0;
    for (int d = 0; d < dim; d++) extent = std::max(extent, yhi[d] - ylo[d]);
    if (!std::isfinite(extent)) KDE_FAIL(PISA_HIP_ERR_INVALID);
    double cell = cut ? k->r_cut / (dim == 3 ? 4.0 : 8.0) : (extent > 0 ? 2.0 * extent : 1.0);
    for (;;) {   // keep the grid below MAX_CELLS and every coordinate below 2^20
        double total = 1.0;
        bool ok = true;
        for (int d = 0; d < dim; d++) {
            const double c = floor((yhi[d] - ylo[d]) / cell) + 1.0;
            total *= c;
            ok = ok && c < (double)(KEY_OFF / 16);
        }
        if (ok && total <= (double)cells_cap(n)) break;
        cell *= 1.25;
    }
    g.cell = cell;
    g.inv_cell = 1.0 / cell;
    k->n_cells = 1;
    for (int d = 0; d < 3; d++) {
        g.nc[d] = d < dim ? (int)(floor((yhi[d] - ylo[d]) / cell) + 1.0) : 1;
        g.ylo[d] = d < dim ? ylo[d] : 0.0;
        k->n_cells *= g.nc[d];
    }
    // ---- resident arrays
    k->ys = ar.take<double>((size_t)dim * n);
    k->wn = ar.take<double>(n);
    k->coef = ar.take<double>(n);
    k->s2 = ar.take<double>(n);
    k->cell_start = ar.take<int32_t>(k->n_cells + 1);
    k->cell_s2min = ar.take<double>(k->n_cells);
    k->scalars = ar.take<double>(8);
    k->pair_count = ar.take<unsigned long long>(8);
    // ---- transient arrays
    double *rec = ar.take<double>((size_t)4 * n);      // (y, weight) per source, 32 bytes
    uint64_t *keys_a = ar.take<uint64_t>(n);
    uint32_t *idx_a = ar.take<uint32_t>(n), *idx_b = ar.take<uint32_t>(n);
    const size_t temp_bytes = sort_temp_bytes(n);
    char *temp = ar.take<char>(temp_bytes);
    if (!ar.ok) KDE_FAIL(PISA_HIP_ERR_NOMEM);
    KDE_TRY_HIP(hipMemsetAsync(k->pair_count, 0, 64, s));
    // ---- whiten, sort by cell, cell table
    uint32_t *flat_a = (uint32_t *)keys_a, *flat_b = flat_a + n;   // (the key arrays of the general form hold two 32-bit ones)
    unsigned bits = 1;
    while (bits < 32 && ((int64_t)1 << bits) < k->n_cells) bits++;
    // cell and source index in one 32-bit word where both fit ((cell << pack) | index): the sort moves keys only -- half the
    // bytes per pass --, sorted on the cell's bits; the order is the stable order of the pair sort (the indices ascend)
    static const int pack_ok = PISA_DEV_INT("KDE_SORT_PACK", 1);
    const int pack = (pack_ok && bits < 32 && n <= ((int64_t)1 << (32 - bits))) ? (int)(32 - bits) : 0;
    KDE_D(kde_whiten_flat_kernel, dim3(nb), dim3(256), 0, s, d_x, d_w, n, g, rec, flat_a, idx_a, pack);
#ifdef PISA_DEV_PROBES
    {   // development (EXPERIMENTS R6-7): N empty launches per estimator -- is the evaluation bound by the number of runtime calls?
        static const int extra = PISA_DEV_INT("KDE_EXTRA_LAUNCHES", 0);
        for (int e = 0; e < extra; e++) hipLaunchKernelGGL(kde_noop_kernel, dim3(1), dim3(64), 0, s, k->scalars);
    }
#endif
    size_t tb = temp_bytes;
    {
        static const int twice = PISA_DEV_INT("KDE_TWICE", 0);   // development: marginal cost of a phase = wall time with it run twice
        for (int rep = 0; rep < ((twice & 4) ? 2 : 1); rep++) {
            if (pack)
                KDE_TRY_HIP(rocprim::radix_sort_keys<FlatSortConfig>(temp, tb, flat_a, flat_b, (size_t)n, (unsigned)pack, 32u, s));
            else
                KDE_TRY_HIP(rocprim::radix_sort_pairs<FlatSortConfig>(temp, tb, flat_a, flat_b, idx_a, idx_b, (size_t)n, 0u, bits, s));
        }
    }
    KDE_D(kde_gather_sources_kernel, dim3(nb), dim3(256), 0, s, rec, 1.0 / sw, 1.0 / k->norm, pack ? flat_b : idx_b,
          pack ? (uint32_t)(((uint64_t)1 << pack) - 1) : 0xFFFFFFFFu, n, k->ys, k->wn,
          k->coef, adaptive ? (double *)nullptr : k->s2);
    hipLaunchKernelGGL(kde_cell_start_flat_kernel, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, s, flat_b, n,
                       k->n_cells, k->cell_start, pack);
    KDE_TRY(check_hip(hipGetLastError(), "kde setup kernels"));
    if (!adaptive) {
        k->s2_range[0] = k->s2_range[1] = 1.0;
        KDE_TRY_HIP(hipMemcpyAsync(k->scalars + 1, k->s2_range, 2 * sizeof(double), hipMemcpyHostToDevice, s));
    } else {
        // pilot estimate at the sources themselves: queries = sorted sources, tiles = cells.  The non-empty cells
        // ("heads") and the workgroups over their sources come from the cell table, read back once.
        std::vector<KdeBlock> blocks;
        std::vector<int32_t> h_starts;
        std::vector<uint64_t> h_keys;
        {
            std::vector<int32_t> cs((size_t)k->n_cells + 1);
            KDE_TRY_HIP(hipMemcpyAsync(cs.data(), k->cell_start, cs.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
            KDE_TRY_HIP(hipStreamSynchronize(s));
            const int64_t nx = g.nc[0], nxy = (int64_t)g.nc[0] * g.nc[1];
            for (int64_t c = 0; c < k->n_cells; c++) {
                const int64_t begin = cs[c], end = cs[c + 1];
                if (end <= begin) continue;
                const int64_t cz = c / nxy, cy = (c - cz * nxy) / nx, cx = c - cz * nxy - cy * nx;
                KdeBlock b;
                b.head = (int32_t)h_starts.size();
                b.c0[0] = b.c1[0] = (int32_t)cx; b.c0[1] = b.c1[1] = (int32_t)cy; b.c0[2] = b.c1[2] = (int32_t)cz;
                h_starts.push_back((int32_t)begin);
                h_keys.push_back(((uint64_t)(cz + KEY_OFF) << 42) | ((uint64_t)(cy + KEY_OFF) << 21) | (uint64_t)(cx + KEY_OFF));
                // equal shares: a cell of 600 sources becomes 300 + 300, not 512 + 88
                const int64_t parts = (end - begin + Q_CHUNK - 1) / Q_CHUNK;
                for (int64_t pp = 0; pp < parts; pp++) {
                    const int64_t q0 = begin + (end - begin) * pp / parts, q1 = begin + (end - begin) * (pp + 1) / parts;
                    b.q_begin = (int32_t)q0;
                    b.q_count = (int32_t)(q1 - q0);
                    blocks.push_back(b);
                }
            }
            if (h_starts.empty()) KDE_FAIL(PISA_HIP_ERR_INVALID);
        }
        const int n_blocks = (int)blocks.size();
        const int n_split = pick_split(n_blocks);
        unsigned long long pilot_host_pairs = 0;   // work counted here instead of by the kernel (kde_local_pilot_wave_kernel)
        KdeBlock *d_blocks = ar.take<KdeBlock>(blocks.size());
        double *pilot = ar.take<double>(n);
        double *part = n_split > 1 ? ar.take<double>((size_t)n_split * n) : pilot;
        if (!ar.ok) KDE_FAIL(PISA_HIP_ERR_NOMEM);
        KDE_TRY_HIP(hipMemcpyAsync(d_blocks, blocks.data(), blocks.size() * sizeof(KdeBlock),
                                   hipMemcpyHostToDevice, s));
        // dense cells get a Hermite series (2-D, with a cut-off, enough sources to pay)
        // host tables uploaded asynchronously below: they live until the synchronisation at the end
        std::vector<int32_t> dense, tcells;
        std::vector<double> hankel;
        // Order of the Hermite / local series: the smallest of 14, 16, 18, 20 whose truncation bound (see above: 2.3 K^2
        // (cell / 2)^P / sqrt(P!) of a cell's weight, all of it at a corner of the cell) is within 4 x tol.  The cells are
        // r_cut / 8 wide, so the bound depends on tol through the cell size as well: 20 at 1e-14 (1.8e-15), 16 at 1e-12
        // (2.8e-12; round 4 took 18 there: 3.5e-14, 28 x finer than the cut-off it sits beside), 14 at 1e-10.  The pilot's
        // error reaches a density only through lambda = (pilot / g)^-alpha, i.e. scaled by alpha (<= 1).
        int P = 20;
        for (int cand : {14, 16, 18}) {
            double bound = 2.3 * 1.09 * 1.09, fact = 1.0;
            for (int i = 1; i <= cand; i++) { bound *= 0.5 * g.cell; fact *= (double)i; }
            if (bound / sqrt(fact) <= 4.0 * tol) { P = cand; break; }
        }
        // local expansions need the intermediate V of every cell of the grid: bounded
        const bool local_ok = g_kde_expansion >= 2 && ceil(sqrt(g.rcut2) * g.inv_cell) <= (double)H2L_MAX_REACH &&
                              (double)k->n_cells * (P * P) * 8.0 < 2.0e9;
        // with local expansions a series costs its cell 400 multiply-adds per source and nothing per
        // target, so every non-empty cell gets one; evaluated target by target (no local expansions) a
        // series pays from ~24 sources
        const int dense_min = local_ok ? hermite_min() : std::max(hermite_min(), HERMITE_MIN_SERIES);
        static const int64_t expansion_min_n = (int64_t)PISA_DEV_LL("KDE_EXPANSION_MIN_N", 1000);
        // (1 000: C3-shaped evaluations of 1e5 / 3e5 events take 11.5 / 27.5 ms with the round-2 threshold of 20 000 sources per
        //  estimator -- direct pair sums below it --, 5.7 / 5.9 ms with this one)
        if (g_kde_expansion && dim == 2 && cut && n >= expansion_min_n) {
            for (size_t h = 0; h < h_starts.size(); h++) {
                const int64_t end = h + 1 < h_starts.size() ? h_starts[h + 1] : n;
                if (end - h_starts[h] < dense_min) continue;
                const int64_t cx = (int64_t)(h_keys[h] & 0x1FFFFF) - KEY_OFF;
                const int64_t cy = (int64_t)((h_keys[h] >> 21) & 0x1FFFFF) - KEY_OFF;
                dense.push_back((int32_t)(cy * g.nc[0] + cx));
            }
        }
        if (!dense.empty()) {
            const int nd = (int)dense.size();
            const int n_heads = (int)h_starts.size();
            const int reach = (int)ceil(sqrt(g.rcut2) * g.inv_cell);
            const bool local_exp = local_ok;
            // Hermite / local coefficients live in a grow-only scratch of the library (up to
            // 3.2 KB per cell: sized by what this call needs, not by the workspace's worst case)
            const size_t pp = (size_t)(P * P);
            // translation passes in H2L_SPLIT parts of the source positions (see kde_h2l4_kernel): V, its flags and the local
            // expansions once per part
            // translation passes: 2 = on the matrix cores where the series order allows (<= 16), 1 = four targets per
            // workgroup on the vector units, 0 = one target per workgroup
            unsigned long long *pstamps = nullptr;
#ifdef PISA_DEV_PROBES
            const size_t n_pstamps = (size_t)n_blocks * (Q_CHUNK / 64);
            static const char *pstamp_path = PISA_DEV_STR("KDE_PILOT_STAMPS");   // development: per-wavefront (start, inputs, end, targets of the block)
            if (pstamp_path) {
                KDE_TRY_HIP(hipMalloc((void **)&pstamps, n_pstamps * 32));
                KDE_TRY_HIP(hipMemsetAsync(pstamps, 0, n_pstamps * 32, s));
            }
#endif
            static const int pilot_form = PISA_DEV_INT("KDE_PILOT_FORM", 1);   // 1 = the local expansions a wavefront per block (where they are all there is)
            static const int h2l_form_cfg = PISA_DEV_INT("KDE_H2L_FORM", 2);
            const int h2l_form = h2l_form_cfg == 2 && P > 16 ? 1 : h2l_form_cfg;
            static const int h2l_split_cfg = PISA_DEV_INT("KDE_H2L_SPLIT", 2);
            const int h2l_split = h2l_form == 2 ? 1 : (h2l_split_cfg >= 1 && h2l_split_cfg <= 8 ? h2l_split_cfg : 2);
            const size_t need = (nd * pp + (local_exp ? h2l_split * (n_heads + (size_t)k->n_cells) * pp + (2 * reach + 1) * (2 * P - 1) : 0))
                                * sizeof(double) + (local_exp ? (size_t)h2l_split * k->n_cells : 0) + 8192;
            double *herm = nullptr;
            KDE_TRY(kde_scratch(need > g_kde_scratch_bytes ? need + need / 2 : need, s, &herm));
            double *local = herm + nd * pp;
            double *d_hankel = local + (local_exp ? h2l_split * n_heads * pp : 0);
            double *d_V = d_hankel + (2 * reach + 1) * (2 * P - 1);
            uint8_t *d_vflag = (uint8_t *)(d_V + (local_exp ? (size_t)h2l_split * k->n_cells * pp : 0));
            int32_t *d_dense = ar.take<int32_t>(dense.size());
            int32_t *slot = ar.take<int32_t>(k->n_cells);
            int32_t *d_tcells = ar.take<int32_t>(n_heads);
            int32_t *hslot = ar.take<int32_t>(k->n_cells);   // cell -> index among the non-empty cells
            if (!ar.ok) KDE_FAIL(PISA_HIP_ERR_NOMEM);
            KDE_TRY_HIP(hipMemcpyAsync(d_dense, dense.data(), dense.size() * sizeof(int32_t),
                                       hipMemcpyHostToDevice, s));
            KDE_TRY_HIP(hipMemsetAsync(slot, 0xFF, (size_t)k->n_cells * sizeof(int32_t), s));
            hipLaunchKernelGGL(kde_slot_scatter_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, s,
                               d_dense, nd, slot);
            tcells.assign(n_heads, 0);
            hankel.clear();
            if (local_exp) {
                for (int h = 0; h < n_heads; h++) {
                    const int64_t cx = (int64_t)(h_keys[h] & 0x1FFFFF) - KEY_OFF;
                    const int64_t cy = (int64_t)((h_keys[h] >> 21) & 0x1FFFFF) - KEY_OFF;
                    tcells[h] = (int32_t)(cy * g.nc[0] + cx);
                }
                const int nh = 2 * P - 1;
                hankel.resize((size_t)(2 * reach + 1) * nh);
                for (int j = -reach; j <= reach; j++) {
                    // h_n(d), d = j * cell / sqrt 2: h_0 = exp(-d^2), h_1 = 2 d h_0, h_{n+1} = 2 d h_n - 2 n h_{n-1}
                    const long double d = (long double)j * (long double)g.cell * 0.70710678118654752440084436210485L;
                    long double h0 = expl(-d * d), h1 = 2.0L * d * h0;
                    double *row = hankel.data() + (size_t)(j + reach) * nh;
                    row[0] = (double)h0;
                    row[1] = (double)h1;
                    for (int m = 1; m + 1 < nh; m++) {
                        const long double h2 = 2.0L * d * h1 - 2.0L * m * h0;
                        row[m + 1] = (double)h2;
                        h0 = h1;
                        h1 = h2;
                    }
                }
                KDE_TRY_HIP(hipMemcpyAsync(d_hankel, hankel.data(), hankel.size() * sizeof(double),
                                           hipMemcpyHostToDevice, s));
                if (nd == n_heads) {
                    // every non-empty cell has a series: its slot IS its index among the non-empty cells
                    // (both lists are in cell order), and the list of target cells is the list of series
                    hslot = slot;
                    d_tcells = d_dense;
                } else {
                    KDE_TRY_HIP(hipMemcpyAsync(d_tcells, tcells.data(), tcells.size() * sizeof(int32_t),
                                               hipMemcpyHostToDevice, s));
                    KDE_TRY_HIP(hipMemsetAsync(hslot, 0xFF, (size_t)k->n_cells * sizeof(int32_t), s));
                    hipLaunchKernelGGL(kde_slot_scatter_kernel, dim3((unsigned)((n_heads + 255) / 256)), dim3(256), 0, s,
                                       d_tcells, n_heads, hslot);
                }
            }
            dim3 grid((unsigned)n_blocks, (unsigned)n_split);
#define KDE_FGT(PP) do { \
                if (local_exp && h2l_form == 2) /* (the matrix-core translation reads the coefficients transposed: the two go together) */ \
                    hipLaunchKernelGGL(kde_hermite_coef_mfma_kernel, dim3((unsigned)nd), dim3(64 * FGT_WAVES), 0, s, g, d_dense, nd, \
                                       k->cell_start, k->ys, n, k->coef, PP, herm); \
                else \
                    hipLaunchKernelGGL(kde_hermite_coef_kernel<PP>, dim3((unsigned)nd), dim3(HC_THREADS), 0, s, g, d_dense, \
                                       k->cell_start, k->ys, n, k->coef, herm); \
                if (local_exp) { \
                    if (h2l_form == 2) { \
                        hipLaunchKernelGGL(kde_h2l_mfma_kernel<0>, dim3((unsigned)g.nc[0], (unsigned)((g.nc[1] + H2LM_T - 1) / H2LM_T)), \
                                           dim3(64 * FGT_WAVES), 0, s, g, hslot, slot, herm, d_hankel, reach, PP, d_V, d_vflag, local); \
                        hipLaunchKernelGGL(kde_h2l_mfma_kernel<1>, dim3((unsigned)((g.nc[0] + H2LM_T - 1) / H2LM_T), (unsigned)g.nc[1]), \
                                           dim3(64 * FGT_WAVES), 0, s, g, hslot, slot, herm, d_hankel, reach, PP, d_V, d_vflag, local); \
                    } else if (h2l_form == 1) { \
                        hipLaunchKernelGGL((kde_h2l4_kernel<PP, 0>), dim3((unsigned)g.nc[0], (unsigned)((g.nc[1] + H2L4_T - 1) / H2L4_T), (unsigned)h2l_split), \
                                           dim3(H2L_THREADS), 0, s, g, hslot, slot, herm, d_hankel, reach, d_V, d_vflag, local, n_heads); \
                        hipLaunchKernelGGL((kde_h2l4_kernel<PP, 1>), dim3((unsigned)((g.nc[0] + H2L4_T - 1) / H2L4_T), (unsigned)g.nc[1], (unsigned)h2l_split), \
                                           dim3(H2L_THREADS), 0, s, g, hslot, slot, herm, d_hankel, reach, d_V, d_vflag, local, n_heads); \
                    } else { \
                        hipLaunchKernelGGL((kde_h2l_kernel<PP, 0>), dim3((unsigned)k->n_cells), dim3(H2L_THREADS), 0, s, g, d_tcells, \
                                           slot, herm, d_hankel, reach, d_V, d_vflag, local); \
                        hipLaunchKernelGGL((kde_h2l_kernel<PP, 1>), dim3((unsigned)n_heads), dim3(H2L_THREADS), 0, s, g, d_tcells, \
                                           slot, herm, d_hankel, reach, d_V, d_vflag, local); \
                    } \
                    if (nd == n_heads && h2l_split == 1 && pilot_form == 1) { \
                        hipLaunchKernelGGL(kde_local_pilot_wave_kernel<PP>, dim3((unsigned)n_blocks, (unsigned)(Q_CHUNK / (64 * FGT_WAVES))), dim3(64 * FGT_WAVES), 0, s, g, \
                                           d_blocks, k->ys, n, local, pilot, pstamps); \
                        pilot_host_pairs = (unsigned long long)(PP * PP / 23 + 1) * (unsigned long long)n; \
                    } else if (nd < n_heads) \
                        hipLaunchKernelGGL((kde_local_pilot_kernel<PP, true>), dim3((unsigned)n_blocks), dim3(KDE_THREADS), 0, s, g, \
                                           d_blocks, k->ys, n, k->coef, k->cell_start, slot, local, n_heads, h2l_form >= 1 ? h2l_split : 1, pilot, k->pair_count); \
                    else \
                        hipLaunchKernelGGL((kde_local_pilot_kernel<PP, false>), dim3((unsigned)n_blocks), dim3(KDE_THREADS), 0, s, g, \
                                           d_blocks, k->ys, n, k->coef, k->cell_start, slot, local, n_heads, h2l_form >= 1 ? h2l_split : 1, pilot, k->pair_count); \
                } else { \
                    hipLaunchKernelGGL(kde_hermite_pilot_kernel<PP>, grid, dim3(KDE_THREADS), 0, s, g, d_blocks, \
                                       k->ys, n, k->coef, k->cell_start, slot, herm, n_split, part, k->pair_count); \
                } } while (0)
            static const int twice = PISA_DEV_INT("KDE_TWICE", 0);
            for (int rep = 0; rep < ((twice & 2) ? 2 : 1); rep++) {
                if (P == 14) KDE_FGT(14); else if (P == 16) KDE_FGT(16); else if (P == 18) KDE_FGT(18); else KDE_FGT(20);
            }
#undef KDE_FGT
            KDE_TRY(check_hip(hipGetLastError(), "kde expansion kernels"));
#ifdef PISA_DEV_PROBES
            if (pstamps) {
                std::vector<unsigned long long> h(n_pstamps * 4);
                KDE_TRY_HIP(hipStreamSynchronize(s));
                KDE_TRY_HIP(hipMemcpy(h.data(), pstamps, h.size() * 8, hipMemcpyDeviceToHost));
                (void)hipFree(pstamps);
                if (FILE *f = fopen(pstamp_path, "wb")) {
                    fwrite(h.data(), 8, h.size(), f);
                    fclose(f);
                }
            }
#endif
            k->n_dense = nd;
            if (local_exp) part = pilot;   // complete (no split)
        } else {
            KDE_TRY((launch_pairs<false, Q_PER_THREAD>(k, d_blocks, n_blocks, n_split, k->ys, n, part, s)));
        }
        if (n_split > 1 && part != pilot)
            hipLaunchKernelGGL(kde_combine_kernel, dim3(nb), dim3(256), 0, s, part, n_split, n,
                               (const uint32_t *)nullptr, pilot);
        hipLaunchKernelGGL(kde_logsum_kernel, dim3(RED_BLOCKS), dim3(RED_THREADS), 0, s, pilot, k->wn, n, partial);
        double *partial_mm = partial + 2 * RED_BLOCKS;   // (its own region: the logsum partials are still being read)
        hipLaunchKernelGGL(kde_bandwidth_kernel, dim3(RED_BLOCKS), dim3(RED_THREADS), 0, s, pilot, k->wn, n,
                           partial, alpha, dim, 1.0 / k->norm, k->coef, k->s2, partial_mm);
        hipLaunchKernelGGL(kde_final_reduce_kernel, dim3(1), dim3(RED_THREADS), 0, s, partial_mm, 2, 0, 1,
                           k->scalars + 1);   // [1] min s2, [2] max s2
        KDE_TRY(check_hip(hipGetLastError(), "kde pilot kernels"));
        KDE_TRY_HIP(hipMemcpyAsync(k->s2_range, k->scalars + 1, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
        // the host copy of the blocks must outlive the asynchronous upload
        KDE_TRY_HIP(hipMemcpyAsync(&k->pairs_pilot, k->pair_count, sizeof(unsigned long long),
                                   hipMemcpyDeviceToHost, s));
        KDE_TRY_HIP(hipMemsetAsync(k->pair_count, 0, 64, s));
        // one synchronisation for the uploads of this block, the two read-backs and the reset
        KDE_TRY_HIP(hipStreamSynchronize(s));
        k->pairs_pilot += pilot_host_pairs;
    }
    if (!adaptive) KDE_TRY_HIP(hipStreamSynchronize(s));
    k->cell_s2min_valid = 0;   // per-cell widest kernel: only the point evaluation needs it
#undef KDE_FAIL
#undef KDE_TRY
#undef KDE_TRY_HIP
    *out = k;
    return PISA_HIP_OK;
}

PISA_API int64_t pisa_hip_kde_resident_bytes(const pisa_hip_kde *k) {
    if (!k) return -1;
    return (int64_t)(resident_bytes(k->dim, k->n, k->n_cells) + (size_t)RED_BLOCKS * 16 * 8 + 4096);
}

PISA_API int64_t pisa_hip_kde_eval_workspace_bytes(const pisa_hip_kde *k, int64_t n_qry) {
    if (!k || n_qry < 1 || n_qry > 0x7FFFFFF0LL) return -1;
    const int64_t m = n_qry;
    size_t total = 2 * (size_t)k->dim * m * 8 + 2 * (size_t)m * 8 + 2 * (size_t)m * 4 + (size_t)m +
                   (size_t)m * 4 + (size_t)m * 8 + split_bytes(m) + (size_t)m * 8 + sort_temp_bytes(m) +
                   ((size_t)m / 128 + (size_t)std::min<int64_t>(m, 1 << 22) + 16) * sizeof(KdeBlock);
    return (int64_t)(total + 64 * 256);
}

PISA_API int pisa_hip_kde_evaluate(pisa_hip_kde *k, const double *d_qry, int64_t m, void *d_work,
                                   int64_t work_bytes, double *d_out, void *stream) {
    if (!k || m < 0) return PISA_HIP_ERR_INVALID;
    if (m == 0) return PISA_HIP_OK;
    if (!d_qry || !d_out || !d_work || m > 0x7FFFFFF0LL) return PISA_HIP_ERR_INVALID;
    hipStream_t s = as_stream(stream);
    Arena ar(d_work, (size_t)work_bytes);
    const int dim = k->dim;
    const KdeGeom &g = k->g;
    // Tile size (cells per side).  A tile's queries are cut into equal workgroups of <= 256 (one
    // query per thread); per query the cost is ~ (cells within reach of the tile) / (share of the
    // 256 lanes in use).  Assumes the queries cover the source grid evenly (a map's bin centres).
    int tile = 1;
    if (g.rcut2 > 0.0) {
        const double per_cell = (double)m / (double)k->n_cells;
        const double reach = 1.5 * sqrt(g.rcut2) * g.inv_cell;
        double best = INFINITY;
        for (int t = 1; t <= 64; t++) {
            const double cnt = per_cell * pow((double)t, dim);
            const double util = cnt / (KDE_THREADS * ceil(cnt / KDE_THREADS));
            const double cost = pow(t + 2.0 * reach, dim) / util;
            if (cost < best) { best = cost; tile = t; }
        }
    }
    double *qy = ar.take<double>((size_t)dim * m), *qys = ar.take<double>((size_t)dim * m);
    uint64_t *keys_a = ar.take<uint64_t>(m), *keys_b = ar.take<uint64_t>(m);
    uint32_t *idx_a = ar.take<uint32_t>(m), *idx_b = ar.take<uint32_t>(m);
    uint8_t *flags = ar.take<uint8_t>(m);
    int32_t *starts = ar.take<int32_t>(m);
    uint64_t *head_keys = ar.take<uint64_t>(m);
    int32_t *d_count = ar.take<int32_t>(8);
    const size_t temp_bytes = sort_temp_bytes(m);
    char *temp = ar.take<char>(temp_bytes);
    if (!ar.ok) return PISA_HIP_ERR_NOMEM;
    const unsigned nb = (unsigned)((m + 255) / 256);
    KDE_D(kde_whiten_key_kernel, dim3(nb), dim3(256), 0, s, d_qry, m, g, tile, 0, qy, keys_a, idx_a);
    size_t tb = temp_bytes;
    PISA_TRY_HIP(hipcub::DeviceRadixSort::SortPairs(temp, tb, keys_a, keys_b, idx_a, idx_b, (int)m, 0, 63, s));
    KDE_D(kde_gather_kernel, dim3(nb), dim3(256), 0, s, qy, (const double *)nullptr, 1.0, idx_b, m, qys,
          (double *)nullptr);
    PISA_CHECK_LAUNCH("kde evaluate setup");
    std::vector<KdeBlock> blocks;
    int rc = query_blocks(keys_b, m, tile, KDE_THREADS, temp, temp_bytes, flags, starts, d_count, head_keys, blocks, s);
    if (rc != PISA_HIP_OK) return rc;
    const int n_blocks = (int)blocks.size();
    const int n_split = pick_split(n_blocks);
    KdeBlock *d_blocks = ar.take<KdeBlock>(blocks.size());
    double *part = ar.take<double>((size_t)n_split * m);
    if (!ar.ok) return PISA_HIP_ERR_NOMEM;
    PISA_TRY_HIP(hipMemcpyAsync(d_blocks, blocks.data(), blocks.size() * sizeof(KdeBlock),
                                hipMemcpyHostToDevice, s));
    if (!k->cell_s2min_valid) {
        hipLaunchKernelGGL(kde_cell_s2min_kernel, dim3((unsigned)((k->n_cells + 255) / 256)), dim3(256), 0, s,
                           k->s2, k->cell_start, k->n_cells, k->cell_s2min);
        PISA_CHECK_LAUNCH("kde_cell_s2min_kernel");
        k->cell_s2min_valid = 1;
    }
    rc = launch_pairs<true, 1>(k, d_blocks, n_blocks, n_split, qys, m, part, s);
    if (rc != PISA_HIP_OK) return rc;
    hipLaunchKernelGGL(kde_combine_kernel, dim3(nb), dim3(256), 0, s, part, n_split, m, idx_b, d_out);
    PISA_CHECK_LAUNCH("kde_combine_kernel");
    PISA_TRY_HIP(hipMemcpyAsync(&k->pairs_eval, k->pair_count, sizeof(unsigned long long),
                                hipMemcpyDeviceToHost, s));
    PISA_TRY_HIP(hipStreamSynchronize(s));   // `blocks` (host) is read by the upload above
    PISA_TRY_HIP(hipMemsetAsync(k->pair_count, 0, 64, s));
    return PISA_HIP_OK;
}
#undef KDE_D

// ---- evaluation on a lattice of points  x[d] = origin[d] + i_d step[d],  0 <= i_d < count[d],
//      out[(i_0 n_1 + i_1) n_2 + i_2]  (numpy.meshgrid(indexing="ij") order)
static int lattice_strip(const pisa_hip_kde *k, const double *step, const int64_t *count) {
    // (rcut2 <= 138, i.e. tol >= 1e-30: the strip's middle value must stay a normal number, see the kernel)
    if (k->dim != 2 || !(k->g.rcut2 > 0.0) || k->g.rcut2 > 138.0 || count[0] * count[1] > 0x7FFFFFF0LL) return 0;
    const double da = k->g.U[0] * step[0];
    if (!(da > 0.0) || !std::isfinite(da) || !(k->s2_range[1] > 0.0)) return 0;
    static const int forced = PISA_DEV_INT("KDE_LATTICE_R", -1);
    if (forced == 0) return 0;
    const double lim = 50.0 / (da * sqrt(k->s2_range[1]));
    for (int R : {32, 16, 8})
        if ((double)R <= lim && (forced < 0 || R <= forced)) return R;
    return 0;
}

// Sub-patch of a wavefront: sw strips of lpw consecutive lines, sw lpw = LG lanes (the wavefront's 64 / LG lane groups
// work different shares on the same sub-patch).  A source costs one pass per sub-patch within its reach, whatever the
// number of strips it reaches there, so the sub-patch should be as compact as the kernel discs: the expected number
// of sub-patches a unit-bandwidth source touches (sources spread evenly over the lattice and its margin) picks sw for
// a given LG; LG = 8 (eight shares side by side: scripts/dev/kde_pass_model.py) unless the lattice then has more
// than 4 096 sub-patches (every sub-patch has a wavefront, partial sums and a share list of its own).
static int64_t lattice_patches(int R, int sw, int lpw, const int64_t *count) {
    const int64_t strips_a = (count[0] + R - 1) / R;
    return ((strips_a + sw - 1) / sw) * ((count[1] + lpw - 1) / lpw);
}

static void lattice_shape(const pisa_hip_kde *k, int R, const double *step, const int64_t *count, int &sw_out, int &lg_out) {
    const int strips_a = (int)((count[0] + R - 1) / R);
    static const int forced_sw = PISA_DEV_INT("KDE_LATTICE_SW", 0);
    static const int forced_lg = PISA_DEV_INT("KDE_LATTICE_LG", 0);
    const double rp = sqrt(k->g.rcut2) / fabs(k->g.U[0] * step[0]), rl = sqrt(k->g.rcut2) / fabs(k->g.U[4] * step[1]);
    const double n0 = (double)count[0], n1 = (double)count[1];
    for (int lg : {8, 16, 32, 64}) {
        if (forced_lg > 0 && lg != forced_lg) continue;   // (development build; one of 8, 16, 32, 64)
        int best = 1;
        double best_cost = INFINITY;
        for (int sw = 1; sw <= lg; sw *= 2) {
            if (forced_sw > 0 ? sw != std::min(forced_sw, lg) : (sw > 1 && sw / 2 >= strips_a)) continue;
            const int lpw = lg / sw;
            double rows = 0.0, cols = 0.0;
            for (int64_t j = 0; j < count[1]; j += lpw)
                rows += std::min(1.0, (2.0 * rl + (double)std::min<int64_t>(lpw, count[1] - j)) / (n1 + 2.0 * rl));
            for (int64_t i = 0; i < count[0]; i += (int64_t)sw * R)
                cols += std::min(1.0, (2.0 * rp + (double)std::min<int64_t>((int64_t)sw * R, count[0] - i)) / (n0 + 2.0 * rp));
            const double cost = rows * cols;
            if (cost < best_cost * (1.0 - 1e-9)) { best_cost = cost; best = sw; }
        }
        sw_out = best;
        lg_out = lg;
        // (every sub-patch also has a list of the shares within reach of it, n_shares entries at most: 256 MB in all)
        const int64_t n_shares = k->n / LAT_SHARE + 1;
        const int64_t cap = std::min<int64_t>(4096, std::max<int64_t>(1, ((int64_t)64 << 20) / n_shares));
        if (lattice_patches(R, best, lg / best, count) <= cap || forced_lg > 0 || lg == 64) return;
    }
}

// number of wavefronts of the lattice kernel (>= one per patch)
static int64_t lattice_waves(int R, int sw, int lpw, const int64_t *count, int64_t n) {
    const int64_t patches = lattice_patches(R, sw, lpw, count);
    // 6 144 = TWICE the wavefronts the chip holds of this kernel (3 per SIMD): with exactly one resident set every
    // SIMD's three wavefronts have equal work, the oldest is served first and the youngest runs the last third of the
    // launch alone at ~40 % issue rate; with half-size wavefronts the second set fills in as the first finishes
    // (round 5: 278 -> 226 us per estimator; 8 192: the same)
    static const int waves = PISA_DEV_INT("KDE_LATTICE_WAVES", 6144);
    int64_t w = std::max<int64_t>(patches, waves);
    w = std::min<int64_t>(w, std::max<int64_t>(patches, (int64_t)(128 << 20) / (R * sw * lpw * 8)));     // partial sums <= 128 MB
    (void)n;   // (the plan gives a patch no more wavefronts than it has shares within reach)
    return w;
}

PISA_API int64_t pisa_hip_kde_lattice_workspace_bytes(const pisa_hip_kde *k, const double *h_step,
                                                      const int64_t *h_count) {
    if (!k || !h_step || !h_count) return -1;
    int64_t m = 1;
    for (int d = 0; d < k->dim; d++) {
        if (h_count[d] < 1 || h_count[d] > 0x7FFFFFF0LL / m) return -1;
        m *= h_count[d];
    }
    const int R = lattice_strip(k, h_step, h_count);
    if (R)
    {
        int sw = 1, lg = 64;
        lattice_shape(k, R, h_step, h_count, sw, lg);
        const size_t waves = (size_t)lattice_waves(R, sw, lg / sw, h_count, k->n);
        const size_t patches = (size_t)lattice_patches(R, sw, lg / sw, h_count);
        return (int64_t)(((size_t)k->n + LAT_SHARE) * LAT_GREC * 8 + waves * R * lg * 8 + ((size_t)k->n / LAT_SHARE + 1) * 32 +
                         patches * ((size_t)k->n / LAT_SHARE + 1) * 4 +   /* lists of the shares within reach of each sub-patch */
                         (patches + 1) * 8 + 4096);
    }
    const int64_t general = pisa_hip_kde_eval_workspace_bytes(k, m);
    return general < 0 ? -1 : general + (int64_t)k->dim * m * 8 + 4096;
}

PISA_API int pisa_hip_kde_evaluate_lattice(pisa_hip_kde *k, const double *h_origin, const double *h_step,
                                           const int64_t *h_count, void *d_work, int64_t work_bytes,
                                           double *d_out, void *stream) {
    if (!k || !h_origin || !h_step || !h_count || !d_out || !d_work) return PISA_HIP_ERR_INVALID;
    const int dim = k->dim;
    int64_t m = 1;
    for (int d = 0; d < dim; d++) {
        if (h_count[d] < 1 || h_count[d] > 0x7FFFFFF0LL / m || !std::isfinite(h_origin[d]) ||
            !std::isfinite(h_step[d]))
            return PISA_HIP_ERR_INVALID;
        m *= h_count[d];
    }
    hipStream_t s = as_stream(stream);
    Arena ar(d_work, (size_t)work_bytes);
    const int R = lattice_strip(k, h_step, h_count);
    if (!R) {   // not a 2-D lattice this form covers: the points written out, general evaluation
        double *x = ar.take<double>((size_t)dim * m);
        if (!ar.ok) return PISA_HIP_ERR_NOMEM;
        hipLaunchKernelGGL(kde_lattice_points_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, dim,
                           h_origin[0], dim > 1 ? h_origin[1] : 0.0, dim > 2 ? h_origin[2] : 0.0, h_step[0],
                           dim > 1 ? h_step[1] : 0.0, dim > 2 ? h_step[2] : 0.0,
                           dim > 1 ? (int)h_count[1] : 1, dim > 2 ? (int)h_count[2] : 1, m, x);
        PISA_CHECK_LAUNCH("kde_lattice_points_kernel");
        const size_t used = ((size_t)dim * m * 8 + 255) & ~(size_t)255;
        return pisa_hip_kde_evaluate(k, x, m, (char *)d_work + used, work_bytes - (int64_t)used, d_out, stream);
    }
    const KdeGeom &g = k->g;
    KdeLattice L;
    const double dx0 = h_origin[0] - g.mean[0], dx1 = h_origin[1] - g.mean[1];
    L.ya0 = g.U[0] * dx0 + g.U[1] * dx1;
    L.yb0 = g.U[4] * dx1;
    L.da = g.U[0] * h_step[0];
    L.sa = g.U[1] * h_step[1];
    L.db = g.U[4] * h_step[1];
    L.n0 = (int32_t)h_count[0];
    L.n1 = (int32_t)h_count[1];
    L.strips_a = (L.n0 + R - 1) / R;
    int sw = 1, lg = 64;
    lattice_shape(k, R, h_step, h_count, sw, lg);
    L.sw = sw;
    L.lpw = lg / sw;
    L.n_colblk = (L.strips_a + L.sw - 1) / L.sw;
    const int n_patches = (int)lattice_patches(R, L.sw, L.lpw, h_count);
    const int n_waves = (int)lattice_waves(R, L.sw, L.lpw, h_count, k->n);
    const int64_t n_shares = (k->n + LAT_SHARE - 1) / LAT_SHARE;
    double *rec = ar.take<double>(((size_t)k->n + LAT_SHARE) * LAT_GREC);   // padded by a whole share of records
    double *part = ar.take<double>((size_t)n_waves * R * lg);
    double *box = ar.take<double>((size_t)n_shares * 4);
    unsigned int *load = ar.take<unsigned int>((size_t)n_patches);
    int32_t *lists = ar.take<int32_t>((size_t)n_patches * (size_t)n_shares);
    int32_t *wstart = ar.take<int32_t>((size_t)n_patches + 1);
    if (!ar.ok) return PISA_HIP_ERR_NOMEM;
    hipLaunchKernelGGL(kde_lattice_prep_kernel, dim3((unsigned)((k->n + 63) / 64)), dim3(64), 0, s, k->ys,
                       k->coef, k->s2, k->n, L.da, g.rcut2, rec, box);
    // (a wavefront's list is cut evenly among its lane groups whatever its length: the floor only bounds the fixed cost per
    // wavefront -- scan of the share boxes, 16 KB of partial sums -- against its work)
    static const int min_shares = PISA_DEV_INT("KDE_LATTICE_MIN_SHARES", 8);
    hipLaunchKernelGGL(kde_lattice_load_kernel, dim3((unsigned)n_patches), dim3(LOAD_THREADS), 0, s, L, R, box, n_shares, load, lists,
                       n_patches, n_waves, min_shares, wstart, k->pair_count + 4);
    unsigned long long *stamps = nullptr;
#ifdef PISA_DEV_PROBES
    static const char *stamp_path = PISA_DEV_STR("KDE_LATTICE_STAMPS");   // development: per-wavefront (start, end, steps, passes | patch)
    if (stamp_path) {
        PISA_TRY_HIP(hipMalloc((void **)&stamps, (size_t)n_waves * 32));
        PISA_TRY_HIP(hipMemsetAsync(stamps, 0, (size_t)n_waves * 32, s));
    }
#endif
    static const int no_count = PISA_DEV_INT("KDE_LATTICE_NO_COUNT", 0);   // development: the launch without its per-wavefront atomic
    unsigned long long *lat_count = no_count ? nullptr : k->pair_count;
#define KDE_LAT(RR, LL) hipLaunchKernelGGL((kde_lattice_kernel<RR, LL>), dim3((unsigned)n_waves), dim3(64), 0, s, L, g.rcut2, rec, k->n, lists, load, wstart, n_patches, part, lat_count, stamps)
#define KDE_LAT_R(RR) do { if (lg == 8) KDE_LAT(RR, 8); else if (lg == 16) KDE_LAT(RR, 16); else if (lg == 32) KDE_LAT(RR, 32); else KDE_LAT(RR, 64); } while (0)
    static const int twice = PISA_DEV_INT("KDE_TWICE", 0);
    for (int rep = 0; rep < ((twice & 1) ? 2 : 1); rep++) {
        if (R == 32) KDE_LAT_R(32); else if (R == 16) KDE_LAT_R(16); else KDE_LAT_R(8);
    }
#undef KDE_LAT_R
#undef KDE_LAT
    hipLaunchKernelGGL(kde_lattice_combine_kernel, dim3((unsigned)(n_patches * R)), dim3(256), 0, s, part, L, R, wstart, d_out);
    PISA_CHECK_LAUNCH("kde lattice kernels");
    PISA_TRY_HIP(hipMemcpyAsync(&k->pairs_eval, k->pair_count, sizeof(unsigned long long),
                                hipMemcpyDeviceToHost, s));
    PISA_TRY_HIP(hipStreamSynchronize(s));
    PISA_TRY_HIP(hipMemsetAsync(k->pair_count, 0, 64, s));
#ifdef PISA_DEV_PROBES
    if (stamps) {
        std::vector<unsigned long long> h((size_t)n_waves * 4);
        PISA_TRY_HIP(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
        (void)hipFree(stamps);
        if (FILE *f = fopen(stamp_path, "ab")) {
            const long long hdr[4] = {(long long)n_waves, (long long)n_patches, (long long)k->n, (long long)lg};
            fwrite(hdr, 8, 4, f);
            fwrite(h.data(), 8, h.size(), f);
            fclose(f);
        }
    }
#endif
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_kde_info(const pisa_hip_kde *k, pisa_hip_kde_info_t *info) {
    if (!k || !info) return PISA_HIP_ERR_INVALID;
    memset(info, 0, sizeof(*info));
    info->dim = k->dim;
    info->n_src = k->n;
    info->factor = k->factor;
    info->norm = k->norm;
    info->sum_w = k->sum_w;
    info->r_cut = k->r_cut;
    info->cell = k->g.cell;
    info->n_cells = k->n_cells;
    for (int d = 0; d < 3; d++) { info->mean[d] = k->mean[d]; info->cells[d] = k->g.nc[d]; }
    for (int i = 0; i < 9; i++) { info->covariance[i] = k->cov[i]; info->inv_cov[i] = k->inv_cov[i]; }
    info->pairs_pilot = (int64_t)k->pairs_pilot;
    info->pairs_eval = (int64_t)k->pairs_eval;
    info->n_dense = k->n_dense;
    return PISA_HIP_OK;
}

/* device pointers into the estimator's resident workspace (sorted source order): for tests and
 * for callers that need the local bandwidths */
PISA_API int pisa_hip_kde_arrays(const pisa_hip_kde *k, const double **d_ys, const double **d_coef,
                                 const double **d_s2) {
    if (!k) return PISA_HIP_ERR_INVALID;
    if (d_ys) *d_ys = k->ys;
    if (d_coef) *d_coef = k->coef;
    if (d_s2) *d_s2 = k->s2;
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_kde_eval(int32_t dim, const double *d_src, const double *d_coef,
                               const double *d_s2, int64_t n_src, const double *d_qry,
                               int64_t n_qry, const double *h_inv_cov, double *d_out,
                               void *stream) {
    if (dim < 1 || dim > 3 || n_src < 0 || n_qry < 0 || !h_inv_cov) return PISA_HIP_ERR_INVALID;
    if (n_qry == 0) return PISA_HIP_OK;
    if (!d_qry || !d_out || (n_src > 0 && (!d_src || !d_coef || !d_s2))) return PISA_HIP_ERR_INVALID;
    const double *ic = h_inv_cov;  // row-major dim x dim, symmetric
    double c00 = ic[0], c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
    if (dim >= 2) { c01 = ic[1]; c11 = ic[dim + 1]; }
    if (dim == 3) { c02 = ic[2]; c12 = ic[dim + 2]; c22 = ic[2 * dim + 2]; }
    const unsigned qblocks = (unsigned)((n_qry + KDE_THREADS - 1) / KDE_THREADS);
    // enough workgroups for 256 CUs x 4: split the sources when there are few query points
    int n_split = 1;
    if (qblocks < 1024 && n_src > 4 * KDE_TILE) {
        n_split = (int)((1024 + qblocks - 1) / qblocks);
        const int64_t max_split = n_src / (2 * KDE_TILE);
        if (n_split > max_split) n_split = (int)max_split;
        if (n_split > 64) n_split = 64;
        if (n_split < 1) n_split = 1;
    }
    int64_t src_chunk = n_src;
    double *dst = d_out;
    hipStream_t s = as_stream(stream);
    if (n_split > 1) {
        src_chunk = ((n_src + n_split - 1) / n_split + KDE_TILE - 1) / KDE_TILE * KDE_TILE;
        n_split = (int)((n_src + src_chunk - 1) / src_chunk);
        const size_t need = (size_t)n_split * n_qry * sizeof(double);
        int rc_s = kde_scratch(need, s, &dst);
        if (rc_s) return rc_s;
    }
    dim3 block(KDE_THREADS), grid(qblocks, (unsigned)n_split);
#define KDE_LAUNCH(DD) hipLaunchKernelGGL(kde_eval_kernel<DD>, grid, block, 0, s, d_src, d_coef, d_s2, n_src, d_qry, n_qry, c00, c01, c02, c11, c12, c22, src_chunk, dst)
    if (dim == 1) KDE_LAUNCH(1);
    else if (dim == 2) KDE_LAUNCH(2);
    else KDE_LAUNCH(3);
#undef KDE_LAUNCH
    PISA_CHECK_LAUNCH("kde_eval_kernel");
    if (n_split > 1) {
        hipLaunchKernelGGL(kde_reduce_kernel, dim3(qblocks), dim3(256), 0, s, dst, n_split, n_qry, d_out);
        PISA_CHECK_LAUNCH("kde_reduce_kernel");
    }
    return PISA_HIP_OK;
}
