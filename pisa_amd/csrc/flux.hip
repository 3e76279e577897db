// flux.hip -- nominal atmospheric flux from a 2-D (azimuth-averaged) Honda table,
// `calculate_2d_flux_weights` (pisa/utils/flux_weights.py:267-349) for
// flux.honda_ip (pisa/stages/flux/honda_ip.py:59-107).
//
// The reference walks the events in a Python loop: 20 FITPACK spline-derivative
// evaluations in log10(E), a running sum, a NEW interpolating cubic spline through
// the 21 coszen knots (scipy splrep) and its derivative at the event's coszen.
// Here one thread does one event (or grid node) for all four primaries:
//   * the band splines' coefficients are prepared once on the host (scipy splrep,
//     exactly the reference's construction); the kernel differentiates and
//     evaluates them the way FITPACK's splder does (coefficient differences
//     k*(c[i+1]-c[i])/(t[i+k+1]-t[i+1]), then the non-zero B-splines of degree k-1
//     from fpbspl's recurrence, summed in the same order);
//   * interpolation through fixed knots is linear in the data, so the per-event
//     splrep becomes c = Cmat . y with the cardinal-spline coefficient matrix Cmat
//     [21][21] computed once by scipy; only the four coefficients around coszen are
//     formed.  Equal to the reference's QR solve up to rounding (~1e-15).
// HBM: 16 B in, 32 B out per event; the tables (4 x 20 x 102 coefficients = 65 KB)
// stay in L2.  Compute bound at ~2.5 kflop per event, far off the hot loop: fluxes
// change only when the table does.
#include "common.hpp"

namespace pisa {

constexpr int FLUX_MAX_BANDS = 32;

struct FluxDev {
    int n_band;            // coszen bands of the table (20)
    int n_te;              // knots of a band spline (106)
    const double *t_e;     // [n_te]
    const double *c_e;     // [4][n_band][n_te - 4] band-spline coefficients, primaries (nue, numu, nuebar, numubar)
    int n_tcz;             // knots of the coszen spline (25)
    const double *t_cz;    // [n_tcz]
    const double *cmat;    // [n_tcz - 4][n_band + 1] cardinal-spline coefficients
    double cz_step;        // 0.1: width of a coszen band (flux_weights.py:343)
    int enpow;
};

// fpbspl: the k+1 non-zero B-splines of degree k at x for the knot interval
// t[l] <= x < t[l+1] (0-based l); k <= 3
__device__ __forceinline__ void fpbspl(const double *__restrict__ t, int k, double x, int l,
                                       double (&h)[4]) {
    double hh[3];
    h[0] = 1.0;
    for (int j = 1; j <= k; j++) {
        for (int i = 0; i < j; i++) hh[i] = h[i];
        h[0] = 0.0;
        for (int i = 1; i <= j; i++) {
            const double tli = t[l + i], tlj = t[l + i - j];
            if (tli == tlj) {
                h[i] = 0.0;
                continue;
            }
            const double f = hh[i - 1] / (tli - tlj);
            h[i - 1] = h[i - 1] + f * (tli - x);
            h[i] = f * (x - tlj);
        }
    }
}

// knot interval of x, clamped to [k, n-k-2] (splder's search; x outside the knot
// range extrapolates the end polynomial, scipy's default ext=0)
__device__ __forceinline__ int find_interval(const double *__restrict__ t, int n, int k, double x) {
    int lo = k, hi = n - k - 2;
    while (lo < hi) {  // largest l in [lo, hi] with t[l] <= x
        const int mid = (lo + hi + 1) >> 1;
        if (x >= t[mid]) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// first derivative at x of the cubic spline (t, c) whose interval l is known:
// splder with nu = 1
__device__ __forceinline__ double cubic_derivative(const double *__restrict__ t, const double (&c)[4],
                                                   int l, double x) {
    // c[m] = coefficient l-3+m; derivative coefficients d[l-3 .. l-1]
    double d[3];
#pragma unroll
    for (int m = 0; m < 3; m++) {
        const int i = l - 3 + m;
        const double fac = t[i + 4] - t[i + 1];
        d[m] = fac > 0.0 ? 3.0 * (c[m + 1] - c[m]) / fac : c[m];
    }
    double h[4];
    fpbspl(t, 2, x, l, h);
    double sp = 0.0;
#pragma unroll
    for (int j = 0; j < 3; j++) sp = sp + d[j] * h[j];
    return sp;
}

__global__ void __launch_bounds__(256)
flux_2d_kernel(const FluxDev f, const double *__restrict__ energy,
               const double *__restrict__ coszen, int64_t n, double *__restrict__ nu_flux,
               double *__restrict__ nubar_flux, int32_t *__restrict__ status) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double e = energy[i], cz = coszen[i];
    if (!(cz >= -1.0 && cz <= 1.0)) {  // flux_weights.py:318-319 raises
        if (status) atomicOr(status, 1);
        return;
    }
    const double x = log10(e);
    const int le = find_interval(f.t_e, f.n_te, 3, x);
    const int lc = find_interval(f.t_cz, f.n_tcz, 3, cz);
    const int n_ce = f.n_te - 4, n_y = f.n_band + 1;
    double div = 1.0;
    for (int p = 0; p < f.enpow; p++) div = div * e;
    double out[4];
    for (int prim = 0; prim < 4; prim++) {
        // running integral over coszen of the bands' dN/dlog10E at x (flux_weights.py:337-343)
        double y[FLUX_MAX_BANDS + 1];
        double run = 0.0;
        y[0] = 0.0 * f.cz_step;
        for (int b = 0; b < f.n_band; b++) {
            const double *cb = f.c_e + ((int64_t)prim * f.n_band + b) * n_ce + (le - 3);
            const double c4[4] = {cb[0], cb[1], cb[2], cb[3]};
            run = run + cubic_derivative(f.t_e, c4, le, x);
            y[b + 1] = run * f.cz_step;
        }
        // the four coefficients of the interpolating coszen spline around lc
        double c4[4];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const double *row = f.cmat + (int64_t)(lc - 3 + m) * n_y;
            double acc = 0.0;
            for (int k = 0; k < n_y; k++) acc = acc + row[k] * y[k];
            c4[m] = acc;
        }
        out[prim] = cubic_derivative(f.t_cz, c4, lc, cz) / div;
    }
    nu_flux[2 * i] = out[0];
    nu_flux[2 * i + 1] = out[1];
    nubar_flux[2 * i] = out[2];
    nubar_flux[2 * i + 1] = out[3];
}

}  // namespace pisa

using namespace pisa;

PISA_API int pisa_hip_flux_2d(const pisa_hip_flux_table *h_table, const double *d_true_energy,
                              const double *d_true_coszen, int64_t n, double *d_nu_flux,
                              double *d_nubar_flux, int32_t *d_status, void *stream) {
    if (!h_table || n < 0) return PISA_HIP_ERR_INVALID;
    if (h_table->n_bands < 1 || h_table->n_bands > FLUX_MAX_BANDS || h_table->n_knots_e < 8 ||
        h_table->n_knots_cz != h_table->n_bands + 1 + 4 || !h_table->d_knots_e ||
        !h_table->d_coef_e || !h_table->d_knots_cz || !h_table->d_cardinal ||
        h_table->enpow < 0 || h_table->enpow > 8)
        return PISA_HIP_ERR_INVALID;
    if (n == 0) return PISA_HIP_OK;
    if (!d_true_energy || !d_true_coszen || !d_nu_flux || !d_nubar_flux) return PISA_HIP_ERR_INVALID;
    FluxDev f;
    f.n_band = h_table->n_bands;
    f.n_te = h_table->n_knots_e;
    f.t_e = h_table->d_knots_e;
    f.c_e = h_table->d_coef_e;
    f.n_tcz = h_table->n_knots_cz;
    f.t_cz = h_table->d_knots_cz;
    f.cmat = h_table->d_cardinal;
    f.cz_step = h_table->cz_step;
    f.enpow = h_table->enpow;
    dim3 block(256), grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(flux_2d_kernel, grid, block, 0, as_stream(stream), f, d_true_energy,
                       d_true_coszen, n, d_nu_flux, d_nubar_flux, d_status);
    PISA_CHECK_LAUNCH("flux_2d_kernel");
    return PISA_HIP_OK;
}
