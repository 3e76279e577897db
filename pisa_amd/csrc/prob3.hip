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
#include "prob3_paths.hpp"

namespace pisa {

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

__global__ void fill_probs_kernel(const double *__restrict__ prob, int init_flav, int flav,
                                  int64_t n, double *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = prob[9 * i + 3 * init_flav + flav];
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

