// stages.hip -- the small per-event services around the hot path (gfx950).  All HBM bound, one pass over the
// columns, one thread per event:
//   lookup_indices   flat bin number of every event by the bin EDGES, underflow -1, overflow n_bins
//                    (pisa/core/bin_indexing.py:46-101; the rule of translation.find_index :504-553)
//   two_nu_osc       weights *= flux x two-flavour vacuum probability     (pisa/stages/osc/two_nu_osc.py:66-127)
//   power_law        out = norm * nominal * (E / pivot)^index             (pisa/stages/flux/astrophysical.py:69-149)
//   shift_toward     out = clip(x + (target - x) * fraction)              (pisa/stages/reco/resolutions.py:74-96)
//   poly_scale       weights *= max(0, prod_k (1 + (lin_k + quad_k p_k) p_k))
//                                                                          (pisa/stages/xsec/genie_sys.py:103-113,
//                                                                           pisa/stages/xsec/dis_sys.py:196-206)
#include <math.h>

#include "common.hpp"

namespace pisa {

constexpr int POLY_MAX = 8;

struct EdgeSet {
    const double *x[3];
    const double *edges[3];
    int32_t n_edges[3];
    int32_t ndim;
};

// translation.py:504-553: [ bin 0 ) [ bin 1 ) ... [ last bin ]; -1 below the first edge and for NaN, n_bins above
__device__ __forceinline__ int find_index(double v, const double *__restrict__ e, int n_edges) {
    if (!(v >= e[0])) return -1;
    const int nb = n_edges - 1;
    if (v > e[nb]) return nb;
    if (v == e[nb]) return nb - 1;
    int lo = 0, hi = nb;            // e[lo] <= v < e[hi]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (v >= e[mid]) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ void __launch_bounds__(256)
lookup_indices_kernel(const EdgeSet s, int64_t n, int64_t *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool under = false, over = false;
    int64_t flat = 0, total = 1;
    for (int d = 0; d < s.ndim; d++) {
        const int nb = s.n_edges[d] - 1;
        const int k = find_index(s.x[d][i], s.edges[d], s.n_edges[d]);
        under = under || k == -1;
        over = over || k == nb;
        flat = flat * nb + k;
        total *= nb;
    }
    // bin_indexing.py:66-74, 93-101: any dimension below -> -1, else any dimension above -> n_bins
    out[i] = under ? -1 : (over ? total : flat);
}

// two_nu_osc.py:101-110 (path length through the Earth from the production height) and :122-127
__global__ void __launch_bounds__(256)
two_nu_osc_kernel(const double2 *__restrict__ flux, double t23, double dm31, const double *__restrict__ energy,
                  const double *__restrict__ coszen, int flav, int64_t n, double *__restrict__ weights) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double2 f = flux[i];
    if (flav == 0) {
        weights[i] *= f.x;
        return;
    }
    const double L1 = 19.;
    const double R = 6378.2 + L1;
    const double zen = acos(coszen[i]);
    const double phi = asin((1 - L1 / R) * sin(zen));
    const double psi = zen - phi;
    const double propdist = sqrt((R - L1) * (R - L1) + R * R - (2 * (R - L1) * R * cos(psi)));
    const double s = sin(1.267 * dm31 * propdist / energy[i]);
    const double p = t23 * (s * s);
    weights[i] *= f.y * (flav == 1 ? 1.0 - p : p);
}

__global__ void __launch_bounds__(256)
power_law_kernel(const double *__restrict__ energy, double pivot, double index, double norm,
                 const double *__restrict__ nominal, int64_t n, double *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double scale = pow(energy[i] / pivot, index);
    out[i] = nominal ? norm * nominal[i] * scale : norm * scale;
}

__global__ void __launch_bounds__(256)
shift_toward_kernel(const double *__restrict__ x, const double *__restrict__ target, double target_value,
                    double fraction, int has_clip, double lo, double hi, int64_t n, double *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v0 = x[i];
    double v = v0 + ((target ? target[i] : target_value) - v0) * fraction;
    if (has_clip) v = v < lo ? lo : (v > hi ? hi : v);      // np.clip: NaN stays NaN
    out[i] = v;
}

struct PolySet {
    const double *lin[POLY_MAX];
    const double *quad[POLY_MAX];
    double p[POLY_MAX];
    int32_t k;
};

__global__ void __launch_bounds__(256)
poly_scale_kernel(const PolySet s, int64_t n, double *__restrict__ weights) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double factor = 1.0;
    for (int k = 0; k < s.k; k++) {
        const double q = s.quad[k] ? s.quad[k][i] * s.p[k] : 0.0;
        factor *= 1. + (s.lin[k][i] + q) * s.p[k];
    }
    // np.maximum(0, factor) hands a NaN on; Python's max(0, NaN) of dis_sys.py:206 gives 0 -- only genie's form is
    // reachable with finite columns, and both agree there
    weights[i] *= factor < 0 ? 0.0 : factor;
}

}  // namespace pisa

using namespace pisa;

static inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

PISA_API int pisa_hip_lookup_indices(const double *const *h_d_sample, const double *const *h_d_edges,
                                     const int32_t *h_n_edges, int32_t ndim, int64_t n, int64_t *d_index,
                                     void *stream) {
    if (ndim < 1 || ndim > 3 || n < 0 || !h_d_sample || !h_d_edges || !h_n_edges) return PISA_HIP_ERR_INVALID;
    EdgeSet s;
    s.ndim = ndim;
    int64_t total = 1;
    for (int d = 0; d < 3; d++) {
        s.x[d] = nullptr; s.edges[d] = nullptr; s.n_edges[d] = 2;
    }
    for (int d = 0; d < ndim; d++) {
        if (h_n_edges[d] < 2 || !h_d_edges[d] || (n > 0 && !h_d_sample[d])) return PISA_HIP_ERR_INVALID;
        s.x[d] = h_d_sample[d]; s.edges[d] = h_d_edges[d]; s.n_edges[d] = h_n_edges[d];
        total *= h_n_edges[d] - 1;
        if (total > (1LL << 40)) return PISA_HIP_ERR_INVALID;
    }
    if (n == 0) return PISA_HIP_OK;
    if (!d_index) return PISA_HIP_ERR_INVALID;
    hipLaunchKernelGGL(lookup_indices_kernel, grid_for(n), dim3(256), 0, as_stream(stream), s, n, d_index);
    PISA_CHECK_LAUNCH("lookup_indices_kernel");
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_two_nu_osc(const double *d_nu_flux, double theta, double deltam31, const double *d_energy,
                                 const double *d_coszen, int32_t flav, int64_t n, double *d_weights, void *stream) {
    if (n < 0 || flav < 0 || flav > 2) return PISA_HIP_ERR_INVALID;
    if (n == 0) return PISA_HIP_OK;
    if (!d_nu_flux || !d_energy || !d_coszen || !d_weights) return PISA_HIP_ERR_INVALID;
    hipLaunchKernelGGL(two_nu_osc_kernel, grid_for(n), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const double2 *>(d_nu_flux), theta, deltam31, d_energy, d_coszen, (int)flav, n,
                       d_weights);
    PISA_CHECK_LAUNCH("two_nu_osc_kernel");
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_power_law(const double *d_energy, double pivot, double index, double norm,
                                const double *d_nominal, int64_t n, double *d_out, void *stream) {
    if (n < 0 || !(pivot > 0)) return PISA_HIP_ERR_INVALID;
    if (n == 0) return PISA_HIP_OK;
    if (!d_energy || !d_out) return PISA_HIP_ERR_INVALID;
    hipLaunchKernelGGL(power_law_kernel, grid_for(n), dim3(256), 0, as_stream(stream), d_energy, pivot, index, norm,
                       d_nominal, n, d_out);
    PISA_CHECK_LAUNCH("power_law_kernel");
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_shift_toward(const double *d_x, const double *d_target, double target_value, double fraction,
                                   int32_t has_clip, double lo, double hi, int64_t n, double *d_out, void *stream) {
    if (n < 0 || (has_clip && !(lo <= hi))) return PISA_HIP_ERR_INVALID;
    if (n == 0) return PISA_HIP_OK;
    if (!d_x || !d_out) return PISA_HIP_ERR_INVALID;
    hipLaunchKernelGGL(shift_toward_kernel, grid_for(n), dim3(256), 0, as_stream(stream), d_x, d_target, target_value,
                       fraction, (int)has_clip, lo, hi, n, d_out);
    PISA_CHECK_LAUNCH("shift_toward_kernel");
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_poly_scale(const double *const *h_d_linear, const double *const *h_d_quad,
                                 const double *h_params, int32_t n_terms, int64_t n, double *d_weights, void *stream) {
    if (n_terms < 0 || n_terms > POLY_MAX || n < 0 || (n_terms && (!h_d_linear || !h_params)))
        return PISA_HIP_ERR_INVALID;
    PolySet s;
    s.k = n_terms;
    for (int k = 0; k < POLY_MAX; k++) {
        s.lin[k] = nullptr; s.quad[k] = nullptr; s.p[k] = 0;
    }
    for (int k = 0; k < n_terms; k++) {
        if (n > 0 && !h_d_linear[k]) return PISA_HIP_ERR_INVALID;
        s.lin[k] = h_d_linear[k];
        s.quad[k] = h_d_quad ? h_d_quad[k] : nullptr;
        s.p[k] = h_params[k];
    }
    if (n == 0) return PISA_HIP_OK;
    if (!d_weights) return PISA_HIP_ERR_INVALID;
    hipLaunchKernelGGL(poly_scale_kernel, grid_for(n), dim3(256), 0, as_stream(stream), s, n, d_weights);
    PISA_CHECK_LAUNCH("poly_scale_kernel");
    return PISA_HIP_OK;
}
