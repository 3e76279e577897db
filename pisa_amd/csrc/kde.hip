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
                double ic22, double *__restrict__ out) {
    __shared__ double t_x[D][KDE_TILE];
    __shared__ double t_c[KDE_TILE];
    __shared__ double t_s[KDE_TILE];
    const int64_t j = (int64_t)blockIdx.x * KDE_THREADS + threadIdx.x;
    double q[D];
#pragma unroll
    for (int d = 0; d < D; d++) q[d] = j < n_qry ? qry[(int64_t)d * n_qry + j] : 0.0;
    double acc = 0.0;
    for (int64_t base = 0; base < n_src; base += KDE_TILE) {
        const int cnt = (int)((n_src - base) < KDE_TILE ? (n_src - base) : KDE_TILE);
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
    dim3 block(KDE_THREADS), grid((unsigned)((n_qry + KDE_THREADS - 1) / KDE_THREADS));
    hipStream_t s = as_stream(stream);
#define KDE_LAUNCH(DD) hipLaunchKernelGGL(kde_eval_kernel<DD>, grid, block, 0, s, d_src, d_coef, d_s2, n_src, d_qry, n_qry, c00, c01, c02, c11, c12, c22, d_out)
    if (dim == 1) KDE_LAUNCH(1);
    else if (dim == 2) KDE_LAUNCH(2);
    else KDE_LAUNCH(3);
#undef KDE_LAUNCH
    PISA_CHECK_LAUNCH("kde_eval_kernel");
    return PISA_HIP_OK;
}
