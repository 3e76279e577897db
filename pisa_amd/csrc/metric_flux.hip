// metric_flux.hip -- bin-sized metric reduction and the Barr flux systematics.
//
//   metric_kernel       Map.metric + np.nansum (map.py:1572-1604) for
//                       llh / poisson_llh / chi2 / mod_chi2 (stats.py); a single
//                       workgroup with a fixed reduction tree => deterministic
//   barr_simple_kernel  apply_sys_vectorized (flux/barr_simple.py:147-233)
#include "common.hpp"
#include "metric_device.hpp"

namespace pisa {

// The metrics beyond the four of the fused tail (hist.hip), per bin, from the raw expectation and its variance:
//   correct_chi2          (k - l)^2 / (s2 + l) + ln(s2 + l)                        stats.py:697-730
//   signed_sqrt_mod_chi2  (k - l) / sqrt(s2 + l)                                   stats.py:762-786
//   mcllh_mean / _eff     Poisson-gamma mixture with a = 0 / 1, b = 0              stats.py:328-438,
//                                                                                   likelihood_functions.py:22-63
//   conv_llh              Poisson smeared with a normal of width sigma, 101 steps   stats.py:440-596
__device__ __forceinline__ double log_poisson_(double k, double l) { return k * log(l) - l - lgamma(k + 1); }

__device__ double conv_poisson_(double k, double l, double s) {
    l = l > SMALL_POS ? l : SMALL_POS;      // Python's max(SMALL_POS, x): NaN -> SMALL_POS
    k = k > SMALL_POS ? k : SMALL_POS;
    s = s > SMALL_POS ? s : SMALL_POS;
    const int st = 2 * (50 + 1);
    const double start = -3 * s, stop = 3 * s;
    const double step = (stop - start) / (st - 1);
    const double shift = 3 * s / (st - 1.);
    const double log_s = log(s), half_log_2pi = 0.5 * log(2 * 3.141592653589793);
    double conv = 0.0, norm = 0.0;
    bool open_ = false;
    for (int j = 0; j < st - 1; j++) {
        const double x = (j * step + start) + shift;
        const double cy = -log_s - half_log_2pi - x * x / (2 * (s * s));
        norm += exp(cy);
        const double fx = x + l;
        open_ = open_ || fx > 0;            // idx = argmax(f_x > 0); f_x ascends
        if (open_) {
            double fy = log_poisson_(k, fx);
            if (fy != fy) fy = 0.0;         // np.nan_to_num
            conv += exp(cy + fy);
        }
    }
    return conv / norm;
}

__device__ double norm_conv_poisson_(double k, double l, double s) {
    const double cp = conv_poisson_(k, l, s);
    const double n1 = exp(log_poisson_(l, l));
    const double n2 = conv_poisson_(l, l, s);
    return cp * n1 / n2;
}

__device__ double metric_bin_wide(int kind, double k, double lam, double s2) {
    if (kind == PISA_HIP_METRIC_CONV_LLH) {
        const double s = sqrt(s2);
        const double a = norm_conv_poisson_(k, lam, s), b = norm_conv_poisson_(k, k, s);
        return log(a > SMALL_POS ? a : SMALL_POS) - log(b > SMALL_POS ? b : SMALL_POS);
    }
    if (lam < SMALL_POS) lam = SMALL_POS;
    if (kind == PISA_HIP_METRIC_CORRECT_CHI2) {
        const double tv = s2 + lam, d = k - lam;
        return (d * d) / tv + log(tv);
    }
    if (kind == PISA_HIP_METRIC_SIGNED_SQRT_MOD_CHI2) return (k - lam) / sqrt(s2 + lam);
    // poisson_gamma(data=k, sum_w=lam, sum_w2=s2, a, b=0)
    const double a = kind == PISA_HIP_METRIC_MCLLH_EFF ? 1.0 : 0.0;
    if (lam <= 0 || s2 < 0) return k == 0 ? 0.0 : -HUGE_VAL;
    if (s2 == 0) return k * log(lam) - lam - lgamma(k + 1);
    const double alpha = (lam * lam) / s2 + a;
    const double beta = lam / s2 + 0.0;
    return alpha * log(beta) + lgamma(k + alpha) - lgamma(k + 1.0) - (k + alpha) * log1p(beta) - lgamma(alpha);
}

__device__ __forceinline__ double metric_any(int kind, double k, double lam, double s2) {
    return kind <= PISA_HIP_METRIC_MOD_CHI2 ? metric_bin(kind, k, lam, s2) : metric_bin_wide(kind, k, lam, s2);
}

// which kinds refuse negative counts / expectations (stats.py:231-240 and the like); the chi2 family beyond
// Pearson's and conv_llh only clip
__device__ __forceinline__ bool kind_checks_sign(int kind) {
    return kind <= PISA_HIP_METRIC_MOD_CHI2 || kind == PISA_HIP_METRIC_MCLLH_MEAN || kind == PISA_HIP_METRIC_MCLLH_EFF;
}

__global__ void __launch_bounds__(256)
metric_kernel(int kind, const double *__restrict__ actual, const double *__restrict__ expected,
              const double *__restrict__ sigma2, int n_maps, int64_t n_bins,
              double *__restrict__ per_bin, double *__restrict__ total,
              int32_t *__restrict__ status) {
    __shared__ double s_sum[256];
    __shared__ int s_flag[2];  // [0] negative input, [1] some |delta| >= 5 eps (chi2)
    if (threadIdx.x < 2) s_flag[threadIdx.x] = 0;
    __syncthreads();
    double acc = 0.0;
    for (int64_t b = threadIdx.x; b < n_bins; b += blockDim.x) {
        double k = actual[b];
        // sum of the maps in index order (distribution_maker.py:274-281)
        double lam = 0.0, s2 = 0.0;
        for (int m = 0; m < n_maps; m++) {
            lam = (m == 0) ? expected[b] : lam + expected[(int64_t)m * n_bins + b];
            if (sigma2) s2 = (m == 0) ? sigma2[b] : s2 + sigma2[(int64_t)m * n_bins + b];
        }
        double v;
        bool finite = (k == k) && (lam == lam) && !isinf(k) && !isinf(lam);
        if (!finite) {
            v = __longlong_as_double(0x7ff8000000000000LL);
        } else {
            if ((k < 0.0 || lam < 0.0) && kind_checks_sign(kind)) atomicOr(&s_flag[0], 1);
            if (kind == PISA_HIP_METRIC_CHI2) {
                double lc = lam < SMALL_POS ? SMALL_POS : lam;
                if (!(fabs(k - lc) < 5 * FTYPE_PREC)) atomicOr(&s_flag[1], 1);
            }
            v = metric_any(kind, k, lam, s2);
        }
        if (per_bin) per_bin[b] = v;
        if (v == v) acc += v;  // np.nansum
    }
    s_sum[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) s_sum[threadIdx.x] += s_sum[threadIdx.x + off];
        __syncthreads();
    }
    bool chi2_zero = (kind == PISA_HIP_METRIC_CHI2) && s_flag[1] == 0;  // stats.py:160-161
    if (chi2_zero && per_bin) {
        for (int64_t b = threadIdx.x; b < n_bins; b += blockDim.x) per_bin[b] = 0.0;
    }
    if (threadIdx.x == 0) {
        total[0] = chi2_zero ? 0.0 : s_sum[0];
        if (status && s_flag[0]) status[0] = PISA_HIP_ERR_NEGATIVE;
    }
}

// Two-stage form for maps too large for one workgroup to walk (> METRIC_ONE_WG_MAX bins): one
// bin per thread, the per-workgroup sums (the same tree as above) go to scratch, and a second
// single-workgroup kernel adds them in index order with the same tree.  Deterministic for a
// given n_bins; the association of the total differs from the one-workgroup form, which is
// why small maps keep that one (it is what the fused tail kernel of hist.hip reproduces).
constexpr int64_t METRIC_ONE_WG_MAX = 4096;  // = PISA_HIP_FINALIZE_METRIC_MAX: every map the fused tail can take

__global__ void __launch_bounds__(256)
metric_partial_kernel(int kind, const double *__restrict__ actual, const double *__restrict__ expected,
                      const double *__restrict__ sigma2, int n_maps, int64_t n_bins,
                      double *__restrict__ per_bin, double *__restrict__ partial,
                      int32_t *__restrict__ flags) {
    __shared__ double s_sum[256];
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double acc = 0.0;
    if (b < n_bins) {
        double k = actual[b];
        double lam = 0.0, s2 = 0.0;
        for (int m = 0; m < n_maps; m++) {
            lam = (m == 0) ? expected[b] : lam + expected[(int64_t)m * n_bins + b];
            if (sigma2) s2 = (m == 0) ? sigma2[b] : s2 + sigma2[(int64_t)m * n_bins + b];
        }
        double v;
        bool finite = (k == k) && (lam == lam) && !isinf(k) && !isinf(lam);
        if (!finite) {
            v = __longlong_as_double(0x7ff8000000000000LL);
        } else {
            if ((k < 0.0 || lam < 0.0) && kind_checks_sign(kind)) atomicOr(&flags[0], 1);
            if (kind == PISA_HIP_METRIC_CHI2) {
                double lc = lam < SMALL_POS ? SMALL_POS : lam;
                if (!(fabs(k - lc) < 5 * FTYPE_PREC)) atomicOr(&flags[1], 1);
            }
            v = metric_any(kind, k, lam, s2);
        }
        if (per_bin) per_bin[b] = v;
        if (v == v) acc = v;  // np.nansum
    }
    s_sum[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) s_sum[threadIdx.x] += s_sum[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = s_sum[0];
}

__global__ void __launch_bounds__(256)
metric_final_kernel(int kind, const double *__restrict__ partial, int n_part,
                    int32_t *__restrict__ flags, double *__restrict__ per_bin, int64_t n_bins,
                    double *__restrict__ total, int32_t *__restrict__ status) {
    __shared__ double s_sum[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n_part; i += 256) acc += partial[i];
    s_sum[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) s_sum[threadIdx.x] += s_sum[threadIdx.x + off];
        __syncthreads();
    }
    const bool chi2_zero = (kind == PISA_HIP_METRIC_CHI2) && flags[1] == 0;  // stats.py:160-161
    if (chi2_zero && per_bin) {
        for (int64_t b = threadIdx.x; b < n_bins; b += 256) per_bin[b] = 0.0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        total[0] = chi2_zero ? 0.0 : s_sum[0];
        if (status && flags[0]) status[0] = PISA_HIP_ERR_NEGATIVE;
        flags[0] = flags[1] = 0;  // ready for the next call
    }
}

// scratch of the two-stage form (grown on demand; one per process, like the rest of the state)
static double *g_metric_partial = nullptr;
static int32_t *g_metric_flags = nullptr;
static int64_t g_metric_parts = 0;

// ------------------------------------------------ binned post-histogram stages
// out = x * scale[i] * scalar, optionally floored (`v < floor ? floor : v`, so NaN
// stays NaN like np.clip / apply_floor_gufunc): hypersurfaces.py:243-257,
// set_variance.py:88-104, csv_icc_hist.py:83
__global__ void __launch_bounds__(256)
bin_scale_kernel(const double *__restrict__ x, const double *__restrict__ scale, double scalar,
                 int has_floor, double floor_, int64_t n, double *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    if (scale) v = v * scale[i];
    v = v * scalar;
    if (has_floor) v = v < floor_ ? floor_ : v;
    out[i] = v;
}

__global__ void __launch_bounds__(256)
bin_sqrt_kernel(const double *__restrict__ x, int64_t n, double *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = sqrt(x[i]);  // set_variance.py:84-86
}

// ------------------------------------------------------- Barr flux systematics
// pisa/utils/barr_parameterization.py:17-113
__device__ __forceinline__ double sign_(double v) {
    if (v == 0) return 0.0;
    return v >= 0 ? 1.0 : -1.0;
}
__device__ __forceinline__ double LogLogParam(double true_energy, double y1, double y2, double x1,
                                              double x2, bool use_cutoff, double cutoff_value) {
    double nu_nubar = sign_(y2);
    y1 = sign_(y1) * log10(fabs(y1) + 0.0001);
    y2 = log10(fabs(y2 + 0.0001));
    // 10^t through exp10 (the reference's `10. ** t` is libm's pow: the same value to an ulp, a third
    // of the instructions; five of them per event made this kernel compute bound)
    double modification =
        nu_nubar * exp10((((y2 - y1) / (x2 - x1)) * (log10(true_energy) - x1) + y1 - 2.));
    if (use_cutoff) modification *= exp(-1. * true_energy / cutoff_value);
    return modification;
}
__device__ __forceinline__ double norm_fcn(double x, double A, double sigma) {
    const double pi = 3.14159265358979323846;
    return A / sqrt(2 * pi * (sigma * sigma)) * exp(-(x * x) / (2 * (sigma * sigma)));
}
__device__ __forceinline__ double ModFlux(int flav, double e, double cz) {
    // all eight e*/z* scale arguments are 1 at the only call site (modRatioNuBar)
    const double e1max_mu = 3., e2max_mu = 43, e1max_e = 2.5, e2max_e = 10;
    const double x1e = 0.5, x2e = 3.;
    const double z1max_mu = 0.6, z2max_mu = 5., z1max_e = 0.3, z2max_e = 5.;
    const double nue_cutoff = 650., numu_cutoff = 1000.;
    const double x1z = 0.5, x2z = 2.;
    if (flav == 1) {
        double A_ave = LogLogParam(e, e1max_mu * 1., e2max_mu * 1., x1e, x2e, false, 0);
        double A_shape = 2.5 * LogLogParam(e, z1max_mu * 1., z2max_mu * 1., x1z, x2z, true, numu_cutoff);
        return A_ave - (norm_fcn(cz, A_shape, 0.36) - 0.6 * A_shape);
    }
    double A_ave = LogLogParam(e, e1max_mu * 1. + e1max_e * 1., e2max_mu * 1. + e2max_e * 1., x1e, x2e, false, 0);
    double A_shape = 1. * LogLogParam(e, z1max_mu * 1. + z1max_e * 1., z2max_mu * 1. + z2max_e * 1., x1z, x2z, true, nue_cutoff);
    return A_ave - (1.5 * norm_fcn(cz, A_shape, 0.36) - 0.7 * A_shape);
}
__device__ __forceinline__ double modRatioUpHor(int flav, double e, double cz, double uphor) {
    const double z1max_mu = 0.6, z2max_mu = 5., z1max_e = 0.3, z2max_e = 5.;
    const double nue_cutoff = 650.;
    const double x1z = 0.5, x2z = 2.;
    if (flav == 0) {
        double A_shape = 1. * fabs(uphor) *
                         LogLogParam(e, (z1max_e + z1max_mu), (z2max_e + z2max_mu), x1z, x2z, true, nue_cutoff);
        return 1 - 0.3 * sign_(uphor) * norm_fcn(cz, A_shape, 0.35);
    }
    return 1.;
}
__device__ __forceinline__ double modRatioNuBar(int nubar, int flav, double e, double cz,
                                                double nubar_sys) {
    double modfactor = nubar_sys * ModFlux(flav, e, cz);
    if (nubar < 0) return fmax(0., 1. / (1 + 0.5 * modfactor));
    return fmax(0., 1. + 0.5 * modfactor);
}
// barr_simple.py:107-138
__device__ __forceinline__ void apply_ratio_scale(double ratio_scale, double in1, double in2,
                                                  double &o0, double &o1) {
    if (in1 == 0. && in2 == 0.) { o0 = 0.; o1 = 0.; return; }
    double orig_ratio = in1 / in2;
    double orig_sum = in1 + in2;
    double nw = orig_sum / (1. + ratio_scale * orig_ratio);
    o0 = ratio_scale * orig_ratio * nw;
    o1 = nw;
}

// The parts of apply_sys_vectorized (barr_simple.py:147-233) that depend on the event only, not on the
// systematics: ModFlux of both flavours (modRatioNuBar = max(0, 1 +- 0.5 sys ModFlux)), the energy and
// the zenith factor of modRatioUpHor's Gaussian (A_shape = |uphor| LL, norm = A_shape / sqrt(2 pi s^2) *
// EX) and log(E / E_pivot) of the spectral tilt -- ten of the kernel's eleven transcendentals.
struct BarrFactors {
    double mf0, mf1;   // ModFlux(0 / 1, E, cz)
    double ll_uh;      // LogLogParam(E, 0.9, 10, 0.5, 2, cutoff 650)
    double ex_uh;      // exp(-cz^2 / (2 * 0.35^2))
    double lx;         // log(E / 24.0900951261)
};
__device__ __forceinline__ BarrFactors barr_factors(double e, double cz) {
    const double z1max_mu = 0.6, z2max_mu = 5., z1max_e = 0.3, z2max_e = 5.;
    const double nue_cutoff = 650.;
    const double x1z = 0.5, x2z = 2.;
    BarrFactors f;
    f.mf0 = ModFlux(0, e, cz);
    f.mf1 = ModFlux(1, e, cz);
    f.ll_uh = LogLogParam(e, (z1max_e + z1max_mu), (z2max_e + z2max_mu), x1z, x2z, true, nue_cutoff);
    const double sigma = 0.35;
    f.ex_uh = exp(-(cz * cz) / (2 * (sigma * sigma)));
    f.lx = log(e / 24.0900951261);
    return f;
}
// apply_sys_vectorized from the event's factors: operation for operation what the reference does once
// the factors are there (flux stage output; tests/test_gpu_flux.py against the reference's goldens)
__device__ __forceinline__ double2 barr_from_factors(const BarrFactors &F, double x_piv, double2 fn, double2 fb,
                                                     int nubar, double nue_numu_ratio, double nu_nubar_ratio,
                                                     double delta_index, double uphor, double barr_nu_nubar) {
    double nu0, nu1, nb0, nb1;
    apply_ratio_scale(nue_numu_ratio, fn.x, fn.y, nu0, nu1);
    apply_ratio_scale(nue_numu_ratio, fb.x, fb.y, nb0, nb1);
    // (E / E_pivot)^delta_index (barr_simple.py:39-45): exp(delta log x) for the energies that exist
    // (same value to two ulp, a third cheaper than the general pow); pow keeps the reference's answers
    // for E <= 0 and NaN
    double idx_scale = x_piv > 0.0 ? exp(delta_index * F.lx) : pow(x_piv, delta_index);
    nu0 *= idx_scale; nu1 *= idx_scale; nb0 *= idx_scale; nb1 *= idx_scale;
    double e0, e1, m0, m1;
    apply_ratio_scale(nu_nubar_ratio, nu0, nb0, e0, e1);  // nue: (nu, nubar)
    apply_ratio_scale(nu_nubar_ratio, nu1, nb1, m0, m1);  // numu
    double o0 = nubar < 0 ? e1 : e0;
    double o1 = nubar < 0 ? m1 : m0;
    // modRatioNuBar (barr_parameterization.py:106-113)
    const double mod0 = barr_nu_nubar * F.mf0, mod1 = barr_nu_nubar * F.mf1;
    o0 *= nubar < 0 ? fmax(0., 1. / (1 + 0.5 * mod0)) : fmax(0., 1. + 0.5 * mod0);
    o1 *= nubar < 0 ? fmax(0., 1. / (1 + 0.5 * mod1)) : fmax(0., 1. + 0.5 * mod1);
    // modRatioUpHor (:84-103): nue only; norm_fcn(cz, A_shape, 0.35) with its exponential from the factors
    {
        const double pi = 3.14159265358979323846;
        const double sigma = 0.35;
        const double A_shape = 1. * fabs(uphor) * F.ll_uh;
        const double norm = A_shape / sqrt(2 * pi * (sigma * sigma)) * F.ex_uh;
        o0 *= 1 - 0.3 * sign_(uphor) * norm;
    }
    o1 *= 1.;
    return make_double2(o0, o1);
}

__device__ __forceinline__ double2 barr_one(double e, double cz, double2 fn, double2 fb, int nubar,
                                            double nue_numu_ratio, double nu_nubar_ratio, double delta_index,
                                            double uphor, double barr_nu_nubar) {
    const BarrFactors F = barr_factors(e, cz);
    return barr_from_factors(F, e / 24.0900951261, fn, fb, nubar, nue_numu_ratio, nu_nubar_ratio, delta_index,
                             uphor, barr_nu_nubar);
}

// ONE pass for a moved flux systematic when the flux is held per event: every input in the fused
// kernel's own resident order and column layout (the engine keeps such copies), the event's
// parameter-free factors from `barr_factors_kernel`, and the static factor folded in on the way out:
//   out[p] = static_w[p] * apply_sys(...)[p]      -- pisa_hip_barr_simple_multi followed by
// pisa_hip_fold_flux_multi, bit for bit, without the second pass and its gather.
__global__ void __launch_bounds__(256)
barr_factors_kernel(const double *__restrict__ e, const double *__restrict__ cz, int64_t n,
                    double *__restrict__ out, int32_t *__restrict__ status) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double ei = e[i];
    if (!(ei > 0.0) && status) atomicOr(status, 1);   // the one-pass form is for energies that exist
    const BarrFactors f = barr_factors(ei, cz[i]);
    out[i] = f.mf0; out[n + i] = f.mf1; out[2 * n + i] = f.ll_uh; out[3 * n + i] = f.ex_uh; out[4 * n + i] = f.lx;
}

constexpr int BARR_FOLD_MAX_SETS = 16;
struct BarrFoldSets {
    pisa_hip_barr_fold_set s[BARR_FOLD_MAX_SETS];
};
__global__ void __launch_bounds__(256)
barr_fold_multi_kernel(const BarrFoldSets sets, double nue_numu_ratio, double nu_nubar_ratio,
                       double delta_index, double uphor, double barr_nu_nubar) {
    const pisa_hip_barr_fold_set &S = sets.s[blockIdx.y];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = S.n;
    if (i >= n) return;
    BarrFactors F;
    const double *f = S.d_factors;
    F.mf0 = f[i]; F.mf1 = f[n + i]; F.ll_uh = f[2 * n + i]; F.ex_uh = f[3 * n + i]; F.lx = f[4 * n + i];
    const double2 r = barr_from_factors(F, 1.0 /* energies checked positive when the factors were made */,
                                        reinterpret_cast<const double2 *>(S.d_nu_flux_nominal)[i],
                                        reinterpret_cast<const double2 *>(S.d_nubar_flux_nominal)[i], S.nubar,
                                        nue_numu_ratio, nu_nubar_ratio, delta_index, uphor, barr_nu_nubar);
    const double w = S.d_static_w[i];
    reinterpret_cast<double2 *>(S.d_out)[i] = make_double2(w * r.x, w * r.y);
}

__global__ void __launch_bounds__(256)
barr_simple_kernel(const double *__restrict__ true_energy, const double *__restrict__ true_coszen,
                   const double *__restrict__ nu_nom, const double *__restrict__ nubar_nom,
                   int nubar, double nue_numu_ratio, double nu_nubar_ratio, double delta_index,
                   double uphor, double barr_nu_nubar, int64_t n, double *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    reinterpret_cast<double2 *>(out)[i] =
        barr_one(true_energy[i], true_coszen[i], reinterpret_cast<const double2 *>(nu_nom)[i],
                 reinterpret_cast<const double2 *>(nubar_nom)[i], nubar, nue_numu_ratio, nu_nubar_ratio,
                 delta_index, uphor, barr_nu_nubar);
}

// all containers of a pipeline in one launch (blockIdx.y = container): on the oscillation grid a
// container is 2*10^4 nodes -- twelve launches of a 4 us kernel cost more than the arithmetic
constexpr int BARR_MAX_SETS = 16;
struct BarrSets {
    pisa_hip_barr_set s[BARR_MAX_SETS];
};
__global__ void __launch_bounds__(256)
barr_simple_multi_kernel(const BarrSets sets, double nue_numu_ratio, double nu_nubar_ratio,
                         double delta_index, double uphor, double barr_nu_nubar) {
    const pisa_hip_barr_set &S = sets.s[blockIdx.y];
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S.n) return;
    reinterpret_cast<double2 *>(S.d_out)[i] =
        barr_one(S.d_true_energy[i], S.d_true_coszen[i],
                 reinterpret_cast<const double2 *>(S.d_nu_flux_nominal)[i],
                 reinterpret_cast<const double2 *>(S.d_nubar_flux_nominal)[i], S.nubar, nue_numu_ratio,
                 nu_nubar_ratio, delta_index, uphor, barr_nu_nubar);
}

}  // namespace pisa

using namespace pisa;

PISA_API int pisa_hip_metric(int32_t kind, const double *d_actual, const double *d_expected,
                             const double *d_sigma2, int32_t n_maps, int64_t n_bins,
                             double *d_per_bin, double *d_total, int32_t *d_status, void *stream) {
    if (kind < 0 || kind > PISA_HIP_METRIC_CONV_LLH || n_maps < 1 || n_bins < 0 || !d_total) return PISA_HIP_ERR_INVALID;
    if (n_bins > 0 && (!d_actual || !d_expected)) return PISA_HIP_ERR_INVALID;
    if (n_bins > METRIC_ONE_WG_MAX) {
        const int64_t n_part = (n_bins + 255) / 256;
        if (n_part > g_metric_parts || !g_metric_flags) {
            if (g_metric_partial) (void)hipFree(g_metric_partial);
            g_metric_partial = nullptr;
            g_metric_parts = 0;
            PISA_TRY_HIP(hipMalloc(&g_metric_partial, (size_t)n_part * sizeof(double)));
            g_metric_parts = n_part;
            if (!g_metric_flags) {
                PISA_TRY_HIP(hipMalloc(&g_metric_flags, 2 * sizeof(int32_t)));
                PISA_TRY_HIP(hipMemsetAsync(g_metric_flags, 0, 2 * sizeof(int32_t), as_stream(stream)));
            }
        }
        hipLaunchKernelGGL(metric_partial_kernel, dim3((unsigned)n_part), dim3(256), 0, as_stream(stream),
                           (int)kind, d_actual, d_expected, d_sigma2, (int)n_maps, n_bins, d_per_bin,
                           g_metric_partial, g_metric_flags);
        PISA_CHECK_LAUNCH("metric_partial_kernel");
        hipLaunchKernelGGL(metric_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), (int)kind,
                           g_metric_partial, (int)n_part, g_metric_flags, d_per_bin, n_bins, d_total, d_status);
        PISA_CHECK_LAUNCH("metric_final_kernel");
        return PISA_HIP_OK;
    }
    hipLaunchKernelGGL(metric_kernel, dim3(1), dim3(256), 0, as_stream(stream), (int)kind, d_actual,
                       d_expected, d_sigma2, (int)n_maps, n_bins, d_per_bin, d_total, d_status);
    PISA_CHECK_LAUNCH("metric_kernel");
    return PISA_HIP_OK;
}


// Folds the static per-event factors into a new flux column, in the fused kernel's resident
// event order and column layout:  out[e] = static_w[e] * flux[perm ? perm[e] : e][0..1].
// layout 0: out[n][2];  layout 1: quad-blocked out[n_pad/256][4][64][2], event 4q+k at
// [q/64][k][q%64] (hist.hip, 16-bit index form).  Runs whenever a flux systematic moved
// (`HotPathEngine.update_flux`): 16 B gathered + 8 B + 8 B index read, 16 B written per event.
namespace pisa {
__global__ void __launch_bounds__(256)
fold_flux_kernel(const double2 *__restrict__ flux, const int64_t *__restrict__ perm,
                 const double *__restrict__ static_w, int64_t n, int layout, double2 *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const double2 f = flux[perm ? perm[e] : e];
    const double w = static_w[e];
    double2 r;
    r.x = w * f.x;
    r.y = w * f.y;
    if (layout == 0) {
        out[e] = r;
    } else {
        const int64_t q = e >> 2;
        out[(((q >> 6) * 4 + (e & 3)) << 6) + (q & 63)] = r;
    }
}

// all containers in one launch (blockIdx.y = container): what a flux systematic costs per evaluation
// when the flux is held per event is two passes over the events (Barr, fold), not 24 launches
constexpr int FOLD_MAX_SETS = 16;
struct FoldSets {
    pisa_hip_fold_set s[FOLD_MAX_SETS];
};
__global__ void __launch_bounds__(256)
fold_flux_multi_kernel(const FoldSets sets) {
    const pisa_hip_fold_set &S = sets.s[blockIdx.y];
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= S.n) return;
    const double2 f = reinterpret_cast<const double2 *>(S.d_flux)[S.d_perm ? S.d_perm[e] : e];
    const double w = S.d_static_w[e];
    double2 r;
    r.x = w * f.x;
    r.y = w * f.y;
    double2 *out = reinterpret_cast<double2 *>(S.d_out);
    if (S.layout == 0) {
        out[e] = r;
    } else {
        const int64_t q = e >> 2;
        out[(((q >> 6) * 4 + (e & 3)) << 6) + (q & 63)] = r;
    }
}
}  // namespace pisa


namespace pisa {
// hist.apply_function with a binned calc_mode (pisa/stages/utils/hist.py:132-160):
//   hist = (unc*w) @ T,  sumw2 = (unc*w)^2 @ T,  bin_unc2 = (unc^2*w) @ T
// with T[i][j] = number of events in calc bin i and output bin j (`hist_transform`, :69-84).  T is
// kept as its non-zeros, grouped by OUTPUT bin (CSR over j): one workgroup per output bin, every
// thread sums the entries t, t+256, ... in index order, fixed reduction tree => reproducible.
__global__ void __launch_bounds__(256)
transform_apply_kernel(const double *__restrict__ w, const double *__restrict__ unc,
                       const int32_t *__restrict__ ptr, const int32_t *__restrict__ col,
                       const double *__restrict__ val, double *__restrict__ hist,
                       double *__restrict__ sumw2, double *__restrict__ bin_unc2) {
    __shared__ double s[3][256];
    const int j = blockIdx.x;
    double a = 0.0, b = 0.0, c = 0.0;
    for (int k = ptr[j] + threadIdx.x; k < ptr[j + 1]; k += 256) {
        const int i = col[k];
        const double u = unc ? unc[i] : 1.0, wi = w[i], t = val[k];
        const double uw = u * wi;
        a += uw * t;
        b += (uw * uw) * t;
        c += ((u * u) * wi) * t;
    }
    s[0][threadIdx.x] = a; s[1][threadIdx.x] = b; s[2][threadIdx.x] = c;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            s[0][threadIdx.x] += s[0][threadIdx.x + off];
            s[1][threadIdx.x] += s[1][threadIdx.x + off];
            s[2][threadIdx.x] += s[2][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (hist) hist[j] = s[0][0];
        if (sumw2) sumw2[j] = s[1][0];
        if (bin_unc2) bin_unc2[j] = s[2][0];
    }
}
}  // namespace pisa

PISA_API int pisa_hip_transform_apply(const double *d_weights, const double *d_unc_weights,
                                      const int32_t *d_ptr, const int32_t *d_col, const double *d_val,
                                      int64_t n_out, double *d_hist, double *d_sumw2, double *d_bin_unc2,
                                      void *stream) {
    if (n_out < 0 || n_out > 0x7FFFFFFF) return PISA_HIP_ERR_INVALID;
    if (n_out == 0) return PISA_HIP_OK;
    if (!d_weights || !d_ptr || !d_col || !d_val) return PISA_HIP_ERR_INVALID;
    hipLaunchKernelGGL(transform_apply_kernel, dim3((unsigned)n_out), dim3(256), 0, as_stream(stream),
                       d_weights, d_unc_weights, d_ptr, d_col, d_val, d_hist, d_sumw2, d_bin_unc2);
    PISA_CHECK_LAUNCH("transform_apply_kernel");
    return PISA_HIP_OK;
}


namespace pisa {
// Where the flux lives on the oscillation grid (flux stages with the calc_mode of osc.prob3, as in
// the IceCube 3-year cfgs) an event's  f_e P_e + f_mu P_mu  (prob3.py:621-622 after two nearest-node
// lookups, container.py:981-1012) depends on its node only.  The products are formed per NODE,
//   out[c][node] = (f_e[c][node] * P_e[side_c][flav_c][node],  f_mu[c][node] * P_mu[...][node]),
// and handed to the fused kernel as the per-container gather table; the events then carry only the
// static pair (w0*aeff, w0*aeff) and a flux systematic costs this launch instead of a pass over
// the events.
struct FluxProbArgs {
    int32_t n_cont;
    int32_t side[16], flav[16];
    const double2 *flux[16];
};
__global__ void __launch_bounds__(256)
flux_prob_tables_kernel(const FluxProbArgs a, const double2 *__restrict__ pepmu, int64_t n_nodes,
                        double2 *__restrict__ out) {
    const int64_t node = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (node >= n_nodes) return;
    const double2 f = a.flux[c][node];
    const double2 p = pepmu[((int64_t)a.side[c] * 3 + a.flav[c]) * n_nodes + node];
    out[(int64_t)c * n_nodes + node] = make_double2(f.x * p.x, f.y * p.y);
}
}  // namespace pisa

PISA_API int pisa_hip_flux_prob_tables(const double *const *h_d_flux_nodes, const int32_t *h_nubar,
                                       const int32_t *h_flav, int32_t n_containers, const double *d_pepmu,
                                       int64_t n_nodes, double *d_out, void *stream) {
    if (n_containers < 0 || n_nodes < 0 || !h_d_flux_nodes || !h_nubar || !h_flav) return PISA_HIP_ERR_INVALID;
    if (n_containers == 0 || n_nodes == 0) return PISA_HIP_OK;
    if (!d_pepmu || !d_out) return PISA_HIP_ERR_INVALID;
    for (int base = 0; base < n_containers; base += 16) {
        FluxProbArgs a;
        a.n_cont = n_containers - base < 16 ? n_containers - base : 16;
        for (int k = 0; k < a.n_cont; k++) {
            const int c = base + k;
            if (!h_d_flux_nodes[c] || (h_nubar[c] != 1 && h_nubar[c] != -1) || h_flav[c] < 0 || h_flav[c] > 2)
                return PISA_HIP_ERR_INVALID;
            a.flux[k] = reinterpret_cast<const double2 *>(h_d_flux_nodes[c]);
            a.side[k] = h_nubar[c] > 0 ? 0 : 1;
            a.flav[k] = h_flav[c];
        }
        hipLaunchKernelGGL(flux_prob_tables_kernel, dim3((unsigned)((n_nodes + 255) / 256), (unsigned)a.n_cont),
                           dim3(256), 0, as_stream(stream), a, reinterpret_cast<const double2 *>(d_pepmu), n_nodes,
                           reinterpret_cast<double2 *>(d_out) + (int64_t)base * n_nodes);
        PISA_CHECK_LAUNCH("flux_prob_tables_kernel");
    }
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_fold_flux(const double *d_flux, const int64_t *d_perm, const double *d_static_w,
                                int64_t n, int32_t layout, double *d_out, void *stream) {
    if (n < 0 || (layout != 0 && layout != 1)) return PISA_HIP_ERR_INVALID;
    if (n == 0) return PISA_HIP_OK;
    if (!d_flux || !d_static_w || !d_out) return PISA_HIP_ERR_INVALID;
    hipLaunchKernelGGL(fold_flux_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                       (const double2 *)d_flux, d_perm, d_static_w, n, (int)layout, (double2 *)d_out);
    PISA_CHECK_LAUNCH("fold_flux_kernel");
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_fold_flux_multi(const pisa_hip_fold_set *h_sets, int32_t n_sets, void *stream) {
    if (n_sets < 0 || (n_sets > 0 && !h_sets)) return PISA_HIP_ERR_INVALID;
    for (int k = 0; k < n_sets; k++) {
        const pisa_hip_fold_set &h = h_sets[k];
        if (h.n < 0 || (h.layout != 0 && h.layout != 1)) return PISA_HIP_ERR_INVALID;
        if (h.n > 0 && (!h.d_flux || !h.d_static_w || !h.d_out)) return PISA_HIP_ERR_INVALID;
    }
    for (int base = 0; base < n_sets; base += FOLD_MAX_SETS) {
        const int nc = n_sets - base < FOLD_MAX_SETS ? n_sets - base : FOLD_MAX_SETS;
        FoldSets sets;
        int64_t n_max = 0;
        for (int k = 0; k < nc; k++) {
            sets.s[k] = h_sets[base + k];
            n_max = h_sets[base + k].n > n_max ? h_sets[base + k].n : n_max;
        }
        if (n_max == 0) continue;
        hipLaunchKernelGGL(fold_flux_multi_kernel, dim3((unsigned)((n_max + 255) / 256), (unsigned)nc), dim3(256), 0,
                           as_stream(stream), sets);
        PISA_CHECK_LAUNCH("fold_flux_multi_kernel");
    }
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_barr_simple(const double *d_true_energy, const double *d_true_coszen,
                                  const double *d_nu_flux_nominal,
                                  const double *d_nubar_flux_nominal, int64_t nubar,
                                  double nue_numu_ratio, double nu_nubar_ratio, double delta_index,
                                  double Barr_uphor_ratio, double Barr_nu_nubar_ratio, int64_t n,
                                  double *d_out, void *stream) {
    if (n < 0 || (nubar != 1 && nubar != -1)) return PISA_HIP_ERR_INVALID;
    if (n == 0) return PISA_HIP_OK;
    if (!d_true_energy || !d_true_coszen || !d_nu_flux_nominal || !d_nubar_flux_nominal || !d_out)
        return PISA_HIP_ERR_INVALID;
    dim3 block(256), grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(barr_simple_kernel, grid, block, 0, as_stream(stream), d_true_energy,
                       d_true_coszen, d_nu_flux_nominal, d_nubar_flux_nominal, (int)nubar,
                       nue_numu_ratio, nu_nubar_ratio, delta_index, Barr_uphor_ratio,
                       Barr_nu_nubar_ratio, n, d_out);
    PISA_CHECK_LAUNCH("barr_simple_kernel");
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_barr_factors(const double *d_true_energy, const double *d_true_coszen, int64_t n,
                                   double *d_factors, int32_t *d_status, void *stream) {
    if (n < 0) return PISA_HIP_ERR_INVALID;
    if (n == 0) return PISA_HIP_OK;
    if (!d_true_energy || !d_true_coszen || !d_factors) return PISA_HIP_ERR_INVALID;
    hipLaunchKernelGGL(barr_factors_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                       d_true_energy, d_true_coszen, n, d_factors, d_status);
    PISA_CHECK_LAUNCH("barr_factors_kernel");
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_barr_fold_multi(const pisa_hip_barr_fold_set *h_sets, int32_t n_sets,
                                      double nue_numu_ratio, double nu_nubar_ratio, double delta_index,
                                      double Barr_uphor_ratio, double Barr_nu_nubar_ratio, void *stream) {
    if (n_sets < 0 || (n_sets > 0 && !h_sets)) return PISA_HIP_ERR_INVALID;
    for (int k = 0; k < n_sets; k++) {
        const pisa_hip_barr_fold_set &h = h_sets[k];
        if (h.n < 0 || (h.nubar != 1 && h.nubar != -1)) return PISA_HIP_ERR_INVALID;
        if (h.n > 0 && (!h.d_nu_flux_nominal || !h.d_nubar_flux_nominal || !h.d_factors || !h.d_static_w || !h.d_out))
            return PISA_HIP_ERR_INVALID;
    }
    for (int base = 0; base < n_sets; base += BARR_FOLD_MAX_SETS) {
        const int nc = n_sets - base < BARR_FOLD_MAX_SETS ? n_sets - base : BARR_FOLD_MAX_SETS;
        BarrFoldSets sets;
        int64_t n_max = 0;
        for (int k = 0; k < nc; k++) {
            sets.s[k] = h_sets[base + k];
            if (sets.s[k].n > n_max) n_max = sets.s[k].n;
        }
        if (n_max == 0) continue;
        hipLaunchKernelGGL(barr_fold_multi_kernel, dim3((unsigned)((n_max + 255) / 256), (unsigned)nc), dim3(256), 0,
                           as_stream(stream), sets, nue_numu_ratio, nu_nubar_ratio, delta_index, Barr_uphor_ratio,
                           Barr_nu_nubar_ratio);
        PISA_CHECK_LAUNCH("barr_fold_multi_kernel");
    }
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_barr_simple_multi(const pisa_hip_barr_set *h_sets, int32_t n_sets,
                                        double nue_numu_ratio, double nu_nubar_ratio, double delta_index,
                                        double Barr_uphor_ratio, double Barr_nu_nubar_ratio, void *stream) {
    if (n_sets < 0 || (n_sets > 0 && !h_sets)) return PISA_HIP_ERR_INVALID;
    for (int k = 0; k < n_sets; k++) {
        const pisa_hip_barr_set &h = h_sets[k];
        if (h.n < 0 || (h.nubar != 1 && h.nubar != -1)) return PISA_HIP_ERR_INVALID;
        if (h.n > 0 && (!h.d_true_energy || !h.d_true_coszen || !h.d_nu_flux_nominal ||
                        !h.d_nubar_flux_nominal || !h.d_out))
            return PISA_HIP_ERR_INVALID;
    }
    for (int base = 0; base < n_sets; base += BARR_MAX_SETS) {
        const int nc = n_sets - base < BARR_MAX_SETS ? n_sets - base : BARR_MAX_SETS;
        BarrSets sets;
        int64_t n_max = 0;
        for (int k = 0; k < nc; k++) {
            sets.s[k] = h_sets[base + k];
            n_max = h_sets[base + k].n > n_max ? h_sets[base + k].n : n_max;
        }
        if (n_max == 0) continue;
        dim3 block(256), grid((unsigned)((n_max + 255) / 256), (unsigned)nc);
        hipLaunchKernelGGL(barr_simple_multi_kernel, grid, block, 0, as_stream(stream), sets, nue_numu_ratio,
                           nu_nubar_ratio, delta_index, Barr_uphor_ratio, Barr_nu_nubar_ratio);
        PISA_CHECK_LAUNCH("barr_simple_multi_kernel");
    }
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_bin_scale(const double *d_x, const double *d_scale, double scalar,
                                int32_t has_floor, double floor_value, int64_t n, double *d_out,
                                void *stream) {
    if (n < 0) return PISA_HIP_ERR_INVALID;
    if (n == 0) return PISA_HIP_OK;
    if (!d_x || !d_out) return PISA_HIP_ERR_INVALID;
    dim3 block(256), grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(bin_scale_kernel, grid, block, 0, as_stream(stream), d_x, d_scale, scalar,
                       (int)has_floor, floor_value, n, d_out);
    PISA_CHECK_LAUNCH("bin_scale_kernel");
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_bin_sqrt(const double *d_x, int64_t n, double *d_out, void *stream) {
    if (n < 0) return PISA_HIP_ERR_INVALID;
    if (n == 0) return PISA_HIP_OK;
    if (!d_x || !d_out) return PISA_HIP_ERR_INVALID;
    dim3 block(256), grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(bin_sqrt_kernel, grid, block, 0, as_stream(stream), d_x, n, d_out);
    PISA_CHECK_LAUNCH("bin_sqrt_kernel");
    return PISA_HIP_OK;
}
