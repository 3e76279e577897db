// kde.hip -- all-pairs Gaussian kernel sums for the KDE smoothing stage
// (pisa/stages/utils/kde.py -> pisa/utils/kde_hist.py:35-217 -> external
// `kde.gaussian_kde`, whose source is NOT in the reference tree: parity of the
// KDE core is UNPINNED).
//
//   out[j] = sum_i coef[i] * exp(-0.5 * s2[i] * (q_j - x_i)^T inv_cov (q_j - x_i))
//
// with per-source squared inverse local bandwidth s2[i] (1 for the pilot
// estimate) and coef[i] = w_i * s_i^d / norm.  One thread owns one query point
// and keeps its partial sum in a register; sources are streamed through LDS in
// tiles (every lane reads the same source -> LDS broadcast, no bank conflicts).
// With few query points (the evaluation grid of a map is ~10^4 points, 40 workgroups) the
// sources are additionally split over blockIdx.y so that the launch fills the chip; the
// per-split partial sums are added in split order by a second kernel.
// Summation order per query point is fixed => bit-reproducible.
// Roofline: FP64 VALU + one exp per pair (compute bound; N*M pairs, 40 B/source).
#include "common.hpp"

namespace pisa {

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

// scratch for the per-split partial sums (grown on demand; one per process, like the rest
// of the library not meant for concurrent calls from several host threads)
static double *g_kde_scratch = nullptr;
static size_t g_kde_scratch_bytes = 0;

}  // namespace pisa

using namespace pisa;

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
        if (need > g_kde_scratch_bytes) {
            PISA_TRY_HIP(hipStreamSynchronize(s));
            if (g_kde_scratch) (void)hipFree(g_kde_scratch);
            g_kde_scratch = nullptr;
            g_kde_scratch_bytes = 0;
            PISA_TRY_HIP(hipMalloc(&g_kde_scratch, need));
            g_kde_scratch_bytes = need;
        }
        dst = g_kde_scratch;
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
