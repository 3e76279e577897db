// stages.hip -- the small per-event services around the hot path (gfx950).  All HBM bound, one pass over the
// columns, one thread per event:
//   lookup_indices   flat bin number of every event by the bin EDGES, underflow -1, overflow n_bins
//                    (pisa/core/bin_indexing.py:46-101; the rule of translation.find_index :504-553)
//   two_nu_osc       weights *= flux x two-flavour vacuum probability     (pisa/stages/osc/two_nu_osc.py:66-127)
//   power_law        out = norm * nominal * (E / pivot)^index             (pisa/stages/flux/astrophysical.py:69-149)
//   shift_toward     out = clip(x + (target - x) * fraction)              (pisa/stages/reco/resolutions.py:74-96)
//   poly_scale       weights *= max(0, scale * prod_k (1 + (lin_k + quad_k p_k) p_k))
//                                                                          (pisa/stages/xsec/genie_sys.py:103-113,
//                                                                           pisa/stages/xsec/dis_sys.py:196-206,
//                                                                           pisa/stages/background/atm_muons.py:95-101)
//   interp_linear    numpy.interp between knots                            (atm_muons.py:82-87, 159-164)
//   column_combination  exp / 1 + / plain sum_g c_g col_g                 (pisa/stages/discr_sys/ultrasurfaces.py:339-356)
//   vector_op        scale / mul / imul / imul_and_scale / itruediv / assign / pow / sqrt / replace_where_counts_gt
//                                                                          (pisa/utils/vectorizer.py:44-209)
//   decoherence      P[n][3][3] of the vacuum decoherence model            (pisa/stages/osc/decoherence.py:66-269)
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
    double scale;
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
    factor *= s.scale;      // atm_muons.py:100-101: clip(weight_mod * atm_muon_scale, 0, inf); 1.0 elsewhere (exact)
    // np.maximum(0, factor) hands a NaN on; Python's max(0, NaN) of dis_sys.py:206 gives 0 -- only genie's form is
    // reachable with finite columns, and both agree there
    weights[i] *= factor < 0 ? 0.0 : factor;
}

// numpy's `interp` (what scipy's interp1d(kind='linear') calls for 1-D float data): linear between the knots, the
// knot value AT a knot; outside [x_0, x_last] the status flag is raised (interp1d's bounds_error) and NaN written
__global__ void __launch_bounds__(256)
interp_linear_kernel(const double *__restrict__ xk, const double *__restrict__ yk, int nk,
                     const double *__restrict__ x, int64_t n, double *__restrict__ out, int32_t *__restrict__ status) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    if (!(v >= xk[0] && v <= xk[nk - 1])) {
        if (status && v == v) atomicOr(status, 1);      // NaN passes through, as in numpy
        out[i] = nan("");
        return;
    }
    int lo = 0, hi = nk - 1;                            // xk[lo] <= v <= xk[hi]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (v >= xk[mid]) lo = mid; else hi = mid;
    }
    if (v == xk[hi]) lo = hi;                           // numpy's search lands on the knot itself
    if (lo == nk - 1 || v == xk[lo]) {
        out[i] = yk[lo];
        return;
    }
    const double slope = (yk[lo + 1] - yk[lo]) / (xk[lo + 1] - xk[lo]);
    double r = slope * (v - xk[lo]) + yk[lo];
    if (r != r) {
        r = slope * (v - xk[lo + 1]) + yk[lo + 1];
        if (r != r && yk[lo] == yk[lo + 1]) r = yk[lo];
    }
    out[i] = r;
}

// pisa/utils/vectorizer.py:44-209: the element-wise helpers stages are written with
__global__ void __launch_bounds__(256)
vector_op_kernel(int op, const double *__restrict__ a, const double *__restrict__ b, double s, int64_t n,
                 double *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    switch (op) {
    case PISA_HIP_VEC_SCALE: out[i] = a[i] * s; break;
    case PISA_HIP_VEC_MUL: out[i] = a[i] * b[i]; break;
    case PISA_HIP_VEC_IMUL: out[i] *= a[i]; break;
    case PISA_HIP_VEC_IMUL_AND_SCALE: out[i] *= a[i] * s; break;
    case PISA_HIP_VEC_ITRUEDIV: out[i] = a[i] == 0.0 ? 0.0 : out[i] / a[i]; break;
    case PISA_HIP_VEC_ASSIGN: out[i] = a[i]; break;
    case PISA_HIP_VEC_POW: out[i] = pow(a[i], s); break;
    case PISA_HIP_VEC_SQRT: out[i] = sqrt(a[i]); break;
    default: if (b[i] > s) out[i] = a[i];      // replace_where_counts_gt
    }
}

// ultrasurfaces.py:339-356: out = exp(sum_g shift_g * grad_g[i]) or 1 + that sum, the sum in the order of the columns
constexpr int COMBO_MAX = 64;
struct ComboSet {
    const double *col[COMBO_MAX];
    double coef[COMBO_MAX];
    int32_t k, mode;
};

__global__ void __launch_bounds__(256)
column_combination_kernel(const ComboSet s, int64_t n, double *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double acc = 0.0;
    for (int g = 0; g < s.k; g++) acc += s.coef[g] * s.col[g][i];
    out[i] = s.mode == 0 ? exp(acc) : (s.mode == 1 ? 1 + acc : acc);
}

// decoherence.py:66-106, 229-269, 449-466: P[i][3][3] of the vacuum decoherence model.  nue stays nue; the numu
// row is (0, 1 - D, D), the nutau row its mirror, D the numu disappearance in the 3-flavour or 2-flavour form
struct DecohArgs {
    double coef[3], gamma[3], delta[3];
    int32_t two_flavor;
};

__global__ void __launch_bounds__(256)
decoherence_kernel(const DecohArgs a, const double *__restrict__ energy, const double *__restrict__ baseline,
                   int64_t n, double *__restrict__ prob) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double E = energy[i], L = baseline[i];
    double dis;
    if (a.two_flavor) {
        // :135-139  gamma in eV, L in m / 1.97e-7; cos(2 * 1.27 dm32 L / E)
        const double decoh = exp(-a.gamma[0] * (L * 1000.0 / 1.97e-7));
        const double osc = cos((2. * 1.27 * a.delta[0] * L) / E);
        dis = a.coef[0] * (1. - (decoh * osc));
    } else {
        double acc = 0.0;
        for (int k = 0; k < 3; k++)        // pairs (1,0), (2,0), (2,1) in the order of :262-265
            acc += a.coef[k] * (1.0 - exp(-a.gamma[k] * L * 5.07e+18) *
                                          cos(a.delta[k] * 1.0e-18 / (2.0 * E) * L * 5.07e+18));
        dis = 2.0 * acc;
    }
    const double surv = 1. - dis;
    double *p = prob + 9 * i;
    p[0] = 1.; p[1] = 0.; p[2] = 1. - 1. - 0.;
    p[3] = 0.; p[4] = surv; p[5] = 1. - 0. - surv;
    p[6] = 0.; p[7] = p[5]; p[8] = surv;
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
                                 const double *h_params, int32_t n_terms, double scale, int64_t n, double *d_weights,
                                 void *stream) {
    if (n_terms < 0 || n_terms > POLY_MAX || n < 0 || (n_terms && (!h_d_linear || !h_params)))
        return PISA_HIP_ERR_INVALID;
    PolySet s;
    s.k = n_terms;
    s.scale = scale;
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

PISA_API int pisa_hip_interp_linear(const double *d_x_knots, const double *d_y_knots, int32_t n_knots, const double *d_x,
                                    int64_t n, double *d_out, int32_t *d_status, void *stream) {
    if (n_knots < 2 || n < 0 || !d_x_knots || !d_y_knots) return PISA_HIP_ERR_INVALID;
    if (n == 0) return PISA_HIP_OK;
    if (!d_x || !d_out) return PISA_HIP_ERR_INVALID;
    hipLaunchKernelGGL(interp_linear_kernel, grid_for(n), dim3(256), 0, as_stream(stream), d_x_knots, d_y_knots,
                       (int)n_knots, d_x, n, d_out, d_status);
    PISA_CHECK_LAUNCH("interp_linear_kernel");
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_decoherence_probs(const double *h_coef, const double *h_gamma, const double *h_delta,
                                        int32_t two_flavor, const double *d_energy, const double *d_baseline, int64_t n,
                                        double *d_probability, void *stream) {
    if (!h_coef || !h_gamma || !h_delta || n < 0) return PISA_HIP_ERR_INVALID;
    if (n == 0) return PISA_HIP_OK;
    if (!d_energy || !d_baseline || !d_probability) return PISA_HIP_ERR_INVALID;
    DecohArgs a;
    for (int k = 0; k < 3; k++) {
        a.coef[k] = h_coef[k]; a.gamma[k] = h_gamma[k]; a.delta[k] = h_delta[k];
    }
    a.two_flavor = two_flavor ? 1 : 0;
    hipLaunchKernelGGL(decoherence_kernel, grid_for(n), dim3(256), 0, as_stream(stream), a, d_energy, d_baseline, n,
                       d_probability);
    PISA_CHECK_LAUNCH("decoherence_kernel");
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_column_combination(const double *const *h_d_columns, const double *h_coef, int32_t n_columns,
                                         int32_t mode, int64_t n, double *d_out, void *stream) {
    if (n_columns < 0 || n_columns > COMBO_MAX || mode < 0 || mode > 2 || n < 0 || (n_columns && (!h_d_columns || !h_coef)))
        return PISA_HIP_ERR_INVALID;
    ComboSet s;
    s.k = n_columns;
    s.mode = mode;
    for (int g = 0; g < COMBO_MAX; g++) {
        s.col[g] = nullptr; s.coef[g] = 0;
    }
    for (int g = 0; g < n_columns; g++) {
        if (n > 0 && !h_d_columns[g]) return PISA_HIP_ERR_INVALID;
        s.col[g] = h_d_columns[g];
        s.coef[g] = h_coef[g];
    }
    if (n == 0) return PISA_HIP_OK;
    if (!d_out) return PISA_HIP_ERR_INVALID;
    hipLaunchKernelGGL(column_combination_kernel, grid_for(n), dim3(256), 0, as_stream(stream), s, n, d_out);
    PISA_CHECK_LAUNCH("column_combination_kernel");
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_vector_op(int32_t op, const double *d_a, const double *d_b, double scalar, int64_t n,
                                double *d_out, void *stream) {
    if (op < PISA_HIP_VEC_SCALE || op > PISA_HIP_VEC_REPLACE_WHERE_COUNTS_GT || n < 0) return PISA_HIP_ERR_INVALID;
    if (n == 0) return PISA_HIP_OK;
    const bool needs_b = op == PISA_HIP_VEC_MUL || op == PISA_HIP_VEC_REPLACE_WHERE_COUNTS_GT;
    if (!d_a || !d_out || (needs_b && !d_b)) return PISA_HIP_ERR_INVALID;
    hipLaunchKernelGGL(vector_op_kernel, grid_for(n), dim3(256), 0, as_stream(stream), (int)op, d_a, d_b, scalar, n, d_out);
    PISA_CHECK_LAUNCH("vector_op_kernel");
    return PISA_HIP_OK;
}
