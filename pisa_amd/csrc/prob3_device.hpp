// prob3_device.hpp -- three-flavour matter oscillation core for gfx950.
//
// Device-side implementation of the per-element work of PISA's prob3
// (pisa/stages/osc/prob3numba/numba_osc_kernels.py:121-872).  One thread owns
// one (energy, path) element and keeps its 3x3 complex state in VGPRs; every
// element-invariant quantity (PMNS, H_vac, H_decay, matter potential, mass
// splittings) is prepared once on the host per nu / nubar sign
// (prob3_make_consts) and arrives through the kernel-argument segment, so it is
// read with scalar loads and lives in SGPRs.
//
// The arithmetic keeps the reference's operation order (the library is built
// with -ffp-contract=off) so that results track the reference to a few ulp;
// the algebraic reformulations are confined to (a) hoisting the invariant
// prologue, (b) never materialising product[3][3][3] / H_minus_M[3][3][3]
// (each (i,j) term is formed and consumed in registers), (c) resolving the
// layer-matrix cache by index instead of storing 120 matrices per thread.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pisa {

struct cplx {
    double re, im;
};

#define PISA_HD __host__ __device__ __forceinline__

PISA_HD cplx cmake(double re, double im) { return cplx{re, im}; }
PISA_HD cplx cadd(cplx a, cplx b) { return cplx{a.re + b.re, a.im + b.im}; }
PISA_HD cplx csub(cplx a, cplx b) { return cplx{a.re - b.re, a.im - b.im}; }
PISA_HD cplx cconj(cplx a) { return cplx{a.re, -a.im}; }
PISA_HD cplx cmul(cplx a, cplx b) {
    return cplx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
PISA_HD cplx cscale(double x, cplx a) { return cplx{x * a.re, x * a.im}; }
PISA_HD double cabs2(cplx a) { return a.re * a.re + a.im * a.im; }
// Smith division (what numba / CPython use for complex / complex)
PISA_HD cplx cdiv(cplx a, cplx b) {
    cplx r;
    if (fabs(b.re) >= fabs(b.im)) {
        double ratio = b.im / b.re;
        double denom = b.re + b.im * ratio;
        r.re = (a.re + a.im * ratio) / denom;
        r.im = (a.im - a.re * ratio) / denom;
    } else {
        double ratio = b.re / b.im;
        double denom = b.re * ratio + b.im;
        r.re = (a.re * ratio + a.im) / denom;
        r.im = (a.im * ratio - a.re) / denom;
    }
    return r;
}
// 1 / b with ONE real division (cdiv above spends three): for operands whose squared modulus stays inside the
// double range -- the scaled cubic of eigvals3_general and the eigenvalue differences of a layer
PISA_HD cplx crecip(cplx b) {
    const double n = 1.0 / (b.re * b.re + b.im * b.im);
    cplx r;
    r.re = b.re * n;
    r.im = -b.im * n;
    return r;
}

struct mat3 {
    cplx m[3][3];
};

// C = A.B with the reference's accumulation order (numba_tools.py:278-289)
PISA_HD void mat_mul(const mat3 &A, const mat3 &B, mat3 &C) {
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++) {
            cplx acc = cmul(A.m[i][0], B.m[0][j]);
            acc = cadd(acc, cmul(A.m[i][1], B.m[1][j]));
            acc = cadd(acc, cmul(A.m[i][2], B.m[2][j]));
            C.m[i][j] = acc;
        }
}

// Element-invariant quantities for ONE nu/nubar sign
// (numba_osc_kernels.py:208-221 and the sign handling of :435-440, :647-653).
struct Prob3Side {
    mat3 U;        // mix_nubar
    mat3 Ud;       // conjugate transpose
    mat3 Hvd;      // H_vac (+ H_decay if decay_flag == 1), without 1/2E
    mat3 V;        // mat_pot (nu) or conj(mat_pot) (nubar)
    double lri[3][3];  // +-lri_pot*1e9 added to Re(H_mat)
    double a_sign;     // +1 (nu) / -1 (nubar): H_mat = a_sign*a*V
    // mass-basis images used by the planned grid form only (eigen_terms):
    // 2E.U^dagger.H.U = X0 + (2E.a_sign.a).XV + 2E.XL
    mat3 X0;       // U^dagger . Hvd . U
    mat3 XV;       // U^dagger . V . U
    mat3 XL;       // U^dagger . lri . U
};

struct Prob3Consts {
    Prob3Side side[2];  // [0] nu, [1] nubar
    double dm[3][3];
    int32_t decay;      // 1: decay branch (complex eigenvalues)
    int32_t pad;
    // planned grid form only: which cubic root follows which vacuum mass state.  get_dms orders
    // the matter eigenvalues by the vacuum ones (:816-825), M_i = mu[argmin_j |dm[i][0] - mv_j|];
    // the vacuum eigenvalues mv_j = 2E.eig(H_vac/2E) do not depend on the energy (only their
    // rounding does), so the assignment is found once per evaluation on the host.
    int32_t vac_order[3];
    int32_t pad2;
};

// Host-side prologue: get_H_vac (:534-569), get_H_decay (:571-603) and the
// nubar conjugations (:208-217), evaluated once per call instead of per element.
inline void prob3_make_consts(const double *dm, const double *mix, const double *mat_pot,
                              const double *mat_decay, const double *lri_pot, int64_t decay_flag,
                              Prob3Consts &c) {
    const cplx *mixc = reinterpret_cast<const cplx *>(mix);
    const cplx *potc = reinterpret_cast<const cplx *>(mat_pot);
    const cplx *decc = reinterpret_cast<const cplx *>(mat_decay);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) c.dm[i][j] = dm[3 * i + j];
    c.decay = (decay_flag == 1) ? 1 : 0;
    c.pad = c.pad2 = 0;
    {
        // vacuum half of get_dms (:754-789) at E = 1 and its ordering rule (:816-825)
        const double x = c.dm[1][0], y = c.dm[2][0];
        const double h = 0.5, one_third = 1.0 / 3.0, two_third = 2.0 / 3.0;
        const double c2_v = -h * (x + y);
        const double p_v = (h * h) * (x * x + y * y - x * y);
        const double q_v = (h * (h * h)) * (x + y) * ((x + y) * (x + y) - 4.5 * x * y);
        const double tmp_v = p_v * (p_v * p_v) - q_v * q_v;
        const double a = two_third * 3.14159265358979323846;
        const double res_v = ::atan2(::sqrt(tmp_v), q_v) * one_third;
        const double b_v = two_third * ::sqrt(p_v);
        const double thv[3] = {res_v + a, res_v - a, res_v};
        double mv[3];
        for (int i = 0; i < 3; i++) mv[i] = 2.0 * (b_v * ::cos(thv[i]) - c2_v * one_third + c.dm[0][0]);
        for (int i = 0; i < 3; i++) {
            double best = ::fabs(c.dm[i][0] - mv[0]);
            int sel = 0;
            for (int j = 1; j < 3; j++) {
                const double t = ::fabs(c.dm[i][0] - mv[j]);
                if (t < best) { best = t; sel = j; }
            }
            c.vac_order[i] = sel;
        }
    }
    for (int s = 0; s < 2; s++) {
        Prob3Side &S = c.side[s];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                cplx u = mixc[3 * i + j];
                S.U.m[i][j] = (s == 0) ? u : cconj(u);
            }
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) S.Ud.m[j][i] = cconj(S.U.m[i][j]);
        // H_vac = U . diag(0, dm21, dm31) . U^dagger
        mat3 diag, tmp, Hvac, Hdec, D;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) diag.m[i][j] = cmake(0.0, 0.0);
        diag.m[1][1] = cmake(c.dm[1][0] + 0.0, 0.0);
        diag.m[2][2] = cmake(c.dm[2][0] + 0.0, 0.0);
        // the reference accumulates from an explicit 0.0: 0 + x == x
        mat_mul(diag, S.Ud, tmp);
        mat_mul(S.U, tmp, Hvac);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) D.m[i][j] = decc[3 * i + j];
        mat_mul(D, S.Ud, tmp);
        mat_mul(S.U, tmp, Hdec);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                S.Hvd.m[i][j] = c.decay ? cadd(Hvac.m[i][j], Hdec.m[i][j]) : Hvac.m[i][j];
                cplx v = potc[3 * i + j];
                S.V.m[i][j] = (s == 0) ? v : cconj(v);
                double l = lri_pot[3 * i + j] * 1e9;
                S.lri[i][j] = (s == 0) ? l : -l;
            }
        S.a_sign = (s == 0) ? 1.0 : -1.0;
        mat3 Lc;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Lc.m[i][j] = cmake(S.lri[i][j], 0.0);
        mat_mul(S.Hvd, S.U, tmp); mat_mul(S.Ud, tmp, S.X0);
        mat_mul(S.V, S.U, tmp);   mat_mul(S.Ud, tmp, S.XV);
        mat_mul(Lc, S.U, tmp);    mat_mul(S.Ud, tmp, S.XL);
    }
}

// ---------------------------------------------------------------- eigenvalues

// get_dms (numba_osc_kernels.py:687-831): real eigenvalues of the Hermitian
// H_full via the trigonometric cubic solution, re-ordered to follow the vacuum
// ordering.  M[k] = 2E * lambda_k.
// vacuum half of get_dms: eigenvalues of the vacuum Hamiltonian, used only to ORDER the
// matter eigenvalues (:816-825).  Depends on the energy alone, so a kernel that walks many
// layers at one energy (event mode) evaluates it once.
__device__ __forceinline__ void get_dms_vacuum(double energy, const double (&dm)[3][3],
                                               double (&mv)[3]) {
    double one_over_two_e = 0.5 / energy;
    const double one_third = 1.0 / 3.0;
    const double two_third = 2.0 / 3.0;
    double x = dm[1][0];
    double y = dm[2][0];
    double c2_v = -one_over_two_e * (x + y);
    double p_v = (one_over_two_e * one_over_two_e) * (x * x + y * y - x * y);
    double q_v = (one_over_two_e * (one_over_two_e * one_over_two_e)) * (x + y) *
                 ((x + y) * (x + y) - 4.5 * x * y);
    double tmp_v = p_v * (p_v * p_v) - q_v * q_v;
    const double a = two_third * 3.14159265358979323846;
    double res_v = atan2(sqrt(tmp_v), q_v) * one_third;
    double b_v = two_third * sqrt(p_v);
    double thv[3] = {res_v + a, res_v - a, res_v};
#pragma unroll
    for (int i = 0; i < 3; i++)
        mv[i] = 2.0 * energy * (b_v * cos(thv[i]) - c2_v * one_third + dm[0][0]);
}

// Reciprocal and square root without the range scaling of the IEEE-complete library forms (12 and
// 18 instructions): hardware estimate (23 bits) + Newton steps in fused arithmetic, <= 1 ulp for
// arguments in the normal range; fast_sqrt(0) = 0.  Event mode only (eigen_terms<false, true>), where
// the kernel is bound by instruction issue.
__device__ __forceinline__ double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double r = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, r, y);
    r = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(y, r, y);
}
__device__ __forceinline__ double fast_sqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return x == 0.0 ? 0.0 : g;
}

// fdlibm kernel polynomials on |r| <= pi/4 with the low part rl of the reduced argument
__device__ __forceinline__ void sincos_kernel(double r, double rl, double *sn_, double *cs_) {
    const double z = r * r;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double v = z * r;
    double ps = __builtin_fma(z, S6, S5);
    ps = __builtin_fma(z, ps, S4);
    ps = __builtin_fma(z, ps, S3);
    ps = __builtin_fma(z, ps, S2);
    *sn_ = r - ((z * (0.5 * rl - v * ps) - rl) - v * S1);
    double pc = __builtin_fma(z, C6, C5);
    pc = __builtin_fma(z, pc, C4);
    pc = __builtin_fma(z, pc, C3);
    pc = __builtin_fma(z, pc, C2);
    pc = __builtin_fma(z, pc, C1);
    pc = z * pc;
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    *cs_ = w + (((1.0 - w) - hz) + (z * pc - r * rl));
}

// sin and cos of a phase of moderate size.  The phases of the reduced form are
// (M_k - Mbar) L/E 2.534, a few hundred radians at most; the library sincos carries the machinery
// for arguments up to 1e308 and is a fifth of the chain kernel.  Three-term Cody-Waite reduction
// with fused multiply-adds (exact products) and the fdlibm kernel polynomials: < 0.8 ulp for
// |x| <= 1e9 (checked against long double over 4e7 random arguments between 1e-6 and 1e9);
// the quadrant is taken from a 64-bit integer, so the reduction keeps working beyond (absolute
// error <= 1e-16 up to 1e12, 1e-14 up to 1e15 -- smaller than the rounding of such a phase
// itself); NaN and infinities give NaN.
__device__ __forceinline__ void sincos_phase(double x, double *s, double *c) {
    const double two_over_pi = 6.36619772367581382433e-01;
    const double p1 = 1.57079632679489655800e+00;  // pi/2 rounded to 53 bits
    const double p2 = 6.12323399573676603587e-17;  // next 53 bits
    const double p3 = -1.49738490485916983e-33;    // remainder
    const double k = rint(x * two_over_pi);
    const double r0 = __builtin_fma(-k, p1, x);
    const double r = __builtin_fma(-k, p2, r0);
    double rl = __builtin_fma(-k, p2, r0 - r);     // what the second step rounded away
    rl = __builtin_fma(-k, p3, rl);
    double sn, cs;
    sincos_kernel(r, rl, &sn, &cs);
    const int q = (int)((long long)k & 3);
    double ss = (q & 1) ? cs : sn, cc = (q & 1) ? sn : cs;
    if (q == 1 || q == 2) cc = -cc;
    if (q >= 2) ss = -ss;
    *s = ss;
    *c = cc;
}

// sin and cos of atan2(y, q) / 3 for y >= 0 (the angle of the trigonometric cubic solution, in
// [0, pi/3]) without the library's atan2 (105 instructions: signs, infinities, an IEEE division) and
// the general range reduction: fdlibm's atan on [0, 1] (one of two argument reductions, 11-term
// polynomial, 0.7 ulp) on min/max of (|q|, y) with Newton reciprocals, and the kernel polynomials
// on the angle itself or, beyond pi/4, on angle - pi/2.  The angle is within 2 ulp of the
// library's.  Event mode only (eigen_terms<false, true>).
__device__ __forceinline__ void sincos_third_angle(double y, double q, double *s, double *c) {
    const double ax = fabs(q);
    const double big = fmax(ax, y), small = fmin(ax, y);
    double t = small * fast_rcp(big);
    t = big > 0.0 ? t : 0.0;                    // atan2(0, 0) = 0
    const bool id1 = t >= 0.6875, red = t >= 0.4375;
    const double num = id1 ? t - 1.0 : __builtin_fma(2.0, t, -1.0), den = id1 ? t + 1.0 : 2.0 + t;
    const double x = red ? num * fast_rcp(den) : t;
    const double z = x * x, w = z * z;
    const double a0 = 3.33333333333329318027e-01, a1 = -1.99999999998764832476e-01,
                 a2 = 1.42857142725034663711e-01, a3 = -1.11111104054623557880e-01,
                 a4 = 9.09088713343650656196e-02, a5 = -7.69187620504482999495e-02,
                 a6 = 6.66107313738753120669e-02, a7 = -5.83357013379057348645e-02,
                 a8 = 4.97687799461593236017e-02, a9 = -3.65315727442169155270e-02,
                 a10 = 1.62858201153657823623e-02;
    double s1 = __builtin_fma(w, a10, a8);
    s1 = __builtin_fma(w, s1, a6);
    s1 = __builtin_fma(w, s1, a4);
    s1 = __builtin_fma(w, s1, a2);
    s1 = __builtin_fma(w, s1, a0);
    s1 = z * s1;
    double s2 = __builtin_fma(w, a9, a7);
    s2 = __builtin_fma(w, s2, a5);
    s2 = __builtin_fma(w, s2, a3);
    s2 = __builtin_fma(w, s2, a1);
    s2 = w * s2;
    const double hi = id1 ? 7.85398163397448278999e-01 : 4.63647609000806093515e-01;   // atan(1), atan(1/2)
    const double lo = id1 ? 3.06161699786838301793e-17 : 2.26987774529616870924e-17;
    const double xs = x * (s1 + s2);
    double ang = red ? hi - ((xs - lo) - x) : x - xs;
    const double p1 = 1.57079632679489655800e+00, p2 = 6.12323399573676603587e-17;     // pi/2
    ang = ax >= y ? ang : (p1 - ang) + p2;
    ang = q < 0.0 ? (2.0 * p1 - ang) + 2.0 * p2 : ang;
    const double res = ang * (1.0 / 3.0);
    const bool far = res > 0.78539816339744830962;
    const double r0 = res - p1;                 // exact (res >= pi/4)
    const double rr = r0 - p2;
    const double r = far ? rr : res, rl = far ? (r0 - rr) - p2 : 0.0;
    double sn, cs;
    sincos_kernel(r, rl, &sn, &cs);
    *s = far ? cs : sn;                         // sin(r + pi/2) = cos r
    *c = far ? -sn : cs;                        // cos(r + pi/2) = -sin r
}

// matter half of get_dms: the three roots 2E.lambda of the characteristic cubic of H, in the
// order of the trigonometric solution (before the vacuum ordering)
// FAST (reduced form only): the three cosines cos(res), cos(res +- 2pi/3) from ONE sincos and
// the addition theorem instead of three library calls (differs by rounding).
template <bool FAST = false>
__device__ __forceinline__ void get_dms_matter_roots(double energy, const mat3 &H,
                                                     const double (&dm)[3][3], double (&mu)[3]) {
    const cplx h01 = H.m[0][1], h12 = H.m[1][2], h20 = H.m[2][0];
    const cplx h00 = H.m[0][0], h11 = H.m[1][1], h22 = H.m[2][2], h02 = H.m[0][2];
    double real_product_a = cmul(cmul(h01, h12), h20).re;
    double real_product_b = cmul(cmul(h00, h11), h22).re;
    double n_emu = h01.re * h01.re + h01.im * h01.im;
    double n_etau = h02.re * h02.re + h02.im * h02.im;
    double n_mutau = h12.re * h12.re + h12.im * h12.im;
    cplx s12 = cadd(h11, h22);
    double c1 = (h00.re * s12.re) - (h00.im * s12.im) + (h11.re * h22.re) - (h11.im * h22.im) -
                n_emu - n_mutau - n_etau;
    double c0 = h00.re * n_mutau + h11.re * n_etau + h22.re * n_emu - 2.0 * real_product_a -
                real_product_b;
    double c2 = -h00.re - h11.re - h22.re;

    const double one_third = 1.0 / 3.0;
    const double two_third = 2.0 / 3.0;
    double p = c2 * c2 - 3.0 * c1;
    p = fmax(0.0, p);
    double q = -13.5 * c0 - c2 * (c2 * c2) + 4.5 * c1 * c2;
    double tmp = 27 * (0.25 * (c1 * c1) * (p - c1) + c0 * (q + 6.75 * c0));
    tmp = fmax(0.0, tmp);

    const double a = two_third * 3.14159265358979323846;
    double res = atan2(sqrt(tmp), q) * one_third;
    double b = two_third * sqrt(p);
    if (FAST) {
        double sn, cs;
        sincos_phase(res, &sn, &cs);
        const double ca = -0.5, sa = 0.86602540378443864676;  // cos, sin of 2pi/3
        const double cth[3] = {cs * ca - sn * sa, cs * ca + sn * sa, cs};
#pragma unroll
        for (int i = 0; i < 3; i++) mu[i] = 2.0 * energy * (b * cth[i] - c2 * one_third + dm[0][0]);
        return;
    }
    double th[3] = {res + a, res - a, res};
#pragma unroll
    for (int i = 0; i < 3; i++) mu[i] = 2.0 * energy * (b * cos(th[i]) - c2 * one_third + dm[0][0]);
}

// matter half of get_dms, given the vacuum eigenvalues
__device__ __forceinline__ void get_dms_matter(double energy, const mat3 &H,
                                               const double (&dm)[3][3], const double (&mv)[3],
                                               double (&M)[3]) {
    double mu[3];
    get_dms_matter_roots(energy, H, dm, mu);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double best = fabs(dm[i][0] - mv[0]);
        double sel = mu[0];
#pragma unroll
        for (int j = 1; j < 3; j++) {
            double t = fabs(dm[i][0] - mv[j]);
            bool lt = t < best;
            sel = lt ? mu[j] : sel;
            best = lt ? t : best;
        }
        M[i] = sel;
    }
}

__device__ __forceinline__ void get_dms(double energy, const mat3 &H, const double (&dm)[3][3],
                                        double (&M)[3]) {
    double mv[3];
    get_dms_vacuum(energy, dm, mv);
    get_dms_matter(energy, H, dm, mv, M);
}

// Complex helpers for the decay branch (general complex 3x3 eigenvalues;
// the reference calls LAPACK zgeev through np.linalg.eigvals,
// numba_osc_kernels.py:655-685).  Closed-form cubic + Newton polishing; the
// order of the eigenvalues is irrelevant to everything downstream.
// FAST (event-mode kernel): the operands come from the cubic of a matrix scaled to unit size, so the modulus is a
// plain square root of the sum of squares (no hypot scaling), and the cube root's angle uses sincos_third_angle
// (no library atan2 + sincos: 260 instructions -> ~90); equal to a few ulp.
template <bool FAST = false>
__device__ __forceinline__ cplx csqrt_d(cplx z) {
    double r = FAST ? sqrt(z.re * z.re + z.im * z.im) : hypot(z.re, z.im);
    if (r == 0.0) return cmake(0.0, 0.0);
    double t = sqrt(0.5 * (r + fabs(z.re)));
    if (z.re >= 0.0) return cmake(t, z.im / (2.0 * t));
    return cmake(fabs(z.im) / (2.0 * t), z.im >= 0.0 ? t : -t);
}
template <bool FAST = false>
__device__ __forceinline__ cplx ccbrt_d(cplx z) {
    double r = FAST ? sqrt(z.re * z.re + z.im * z.im) : hypot(z.re, z.im);
    if (r == 0.0) return cmake(0.0, 0.0);
    double m = cbrt(r);
    double s, c;
    if (FAST) {
        sincos_third_angle(fabs(z.im), z.re, &s, &c);   // angle of (re, |im|) / 3, in [0, pi/3]
        s = z.im < 0.0 ? -s : s;
    } else {
        double ang = atan2(z.im, z.re) / 3.0;
        sincos(ang, &s, &c);
    }
    return cmake(m * c, m * s);
}
template <bool FAST = false>
__device__ inline void eigvals3_general(const mat3 &H, cplx (&lam)[3]) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) s = fmax(s, fmax(fabs(H.m[i][j].re), fabs(H.m[i][j].im)));
    if (s == 0.0) {
        lam[0] = lam[1] = lam[2] = cmake(0.0, 0.0);
        return;
    }
    double inv = 1.0 / s;
    mat3 A;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) A.m[i][j] = cscale(inv, H.m[i][j]);
    cplx tr = cadd(cadd(A.m[0][0], A.m[1][1]), A.m[2][2]);
    cplx c2 = cscale(-1.0, tr);
    cplx m00 = csub(cmul(A.m[1][1], A.m[2][2]), cmul(A.m[1][2], A.m[2][1]));
    cplx m11 = csub(cmul(A.m[0][0], A.m[2][2]), cmul(A.m[0][2], A.m[2][0]));
    cplx m22 = csub(cmul(A.m[0][0], A.m[1][1]), cmul(A.m[0][1], A.m[1][0]));
    cplx c1 = cadd(cadd(m00, m11), m22);
    cplx det = cadd(
        csub(cmul(A.m[0][0], m00),
             cmul(A.m[0][1], csub(cmul(A.m[1][0], A.m[2][2]), cmul(A.m[1][2], A.m[2][0])))),
        cmul(A.m[0][2], csub(cmul(A.m[1][0], A.m[2][1]), cmul(A.m[1][1], A.m[2][0]))));
    cplx c0 = cscale(-1.0, det);
    cplx c2sq = cmul(c2, c2);
    cplx p = csub(c1, cscale(1.0 / 3.0, c2sq));
    cplx q = cadd(csub(cscale(2.0 / 27.0, cmul(c2sq, c2)), cscale(1.0 / 3.0, cmul(c2, c1))), c0);
    cplx halfq = cscale(0.5, q);
    cplx disc = cadd(cmul(halfq, halfq), cscale(1.0 / 27.0, cmul(cmul(p, p), p)));
    cplx sd = csqrt_d<FAST>(disc);
    cplx u1 = csub(sd, halfq), u2 = csub(cscale(-1.0, sd), halfq);
    cplx u3 = (cabs2(u1) >= cabs2(u2)) ? u1 : u2;
    cplx u = ccbrt_d<FAST>(u3);
    const cplx w1 = {-0.5, 0.86602540378443864676}, w2 = {-0.5, -0.86602540378443864676};
    cplx us[3] = {u, cmul(u, w1), cmul(u, w2)};
    cplx shift = cscale(1.0 / 3.0, c2);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        cplx t = (cabs2(us[k]) == 0.0) ? cmake(0.0, 0.0)
                                       : csub(us[k], FAST ? cmul(p, crecip(cscale(3.0, us[k]))) : cdiv(p, cscale(3.0, us[k])));
        lam[k] = csub(t, shift);
    }
    // Newton polishing of the closed-form roots: quadratic convergence from ~1e-8 (nearly degenerate roots) or
    // better, so two steps reach the conditioning of the cubic; the reference-order kernels keep four
    for (int it = 0; it < (FAST ? 2 : 4); it++)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            cplx xk = lam[k];
            cplx f = cadd(cmul(cadd(cmul(cadd(xk, c2), xk), c1), xk), c0);
            cplx df = cadd(cmul(cadd(cscale(3.0, xk), cscale(2.0, c2)), xk), c1);
            if (cabs2(df) > 0.0) lam[k] = csub(xk, FAST ? cmul(f, crecip(df)) : cdiv(f, df));
        }
#pragma unroll
    for (int k = 0; k < 3; k++) lam[k] = cscale(s, lam[k]);
}

// ------------------------------------------------------ one layer's amplitude

// get_transition_matrix (numba_osc_kernels.py:348-478) with
// get_transition_matrix_massbasis (:481-531) and get_product (:834-872) fused:
// returns A (mass basis) for a layer of electron density rho and length
// baseline.  DECAY selects the complex-eigenvalue branch at compile time.
//
// FAST (event-mode kernel): the caller passes the vacuum eigenvalues it computed once
// for this energy (`get_dms_vacuum`; identical values, just not recomputed per layer) and
// the 54 divisions by the three real denominators become multiplications by their
// reciprocals (<= 1 ulp per quotient).
template <bool DECAY, bool FAST = false>
__device__ __forceinline__ void layer_amplitude(const Prob3Side &S, const double (&dm)[3][3],
                                                double energy, double rho, double baseline,
                                                mat3 &A, const double *vacuum_dms = nullptr) {
    // get_H_mat (:605-653) + LRI (:435-440)
    const double tworttwoGf = 1.52588e-4;
    double a = 0.5 * rho * tworttwoGf;
    double sa = S.a_sign * a;  // exact sign flip
    double one_over_two_e = 0.5 / energy;
    mat3 Hf;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            cplx hm = cscale(sa, S.V.m[i][j]);
            hm.re = hm.re + S.lri[i][j];
            Hf.m[i][j] = cadd(cscale(one_over_two_e, S.Hvd.m[i][j]), hm);
        }

    // H in the mass basis, times 2E  (:466-467, :857)
    mat3 tmp, X;
    mat_mul(Hf, S.U, tmp);
    mat_mul(S.Ud, tmp, X);
    double two_e = 2.0 * energy;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) X.m[i][j] = cscale(two_e, X.m[i][j]);

    double L_over_E = baseline / energy;
    const double hbar_c_factor = 2.534;

    if (!DECAY) {
        double M[3];
        if (FAST) {
            const double mv[3] = {vacuum_dms[0], vacuum_dms[1], vacuum_dms[2]};
            get_dms_matter(energy, Hf, dm, mv, M);
        } else {
            get_dms(energy, Hf, dm, M);
        }
        // phases exp(-i M_k L/E 2.534)
        cplx ph[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            double arg = (-M[k]) * L_over_E * hbar_c_factor;
            double s, c;
            sincos(arg, &s, &c);
            ph[k] = cmake(c, s);
        }
        // real denominators (M_k - M_j)(M_k - M_l)  (:870-872)
        double den0 = (M[0] - M[1]) * (M[0] - M[2]);
        double den1 = (M[1] - M[2]) * (M[1] - M[0]);
        double den2 = (M[2] - M[0]) * (M[2] - M[1]);
        const double inv0 = FAST ? 1.0 / den0 : 0.0, inv1 = FAST ? 1.0 / den1 : 0.0,
                     inv2 = FAST ? 1.0 / den2 : 0.0;
        // diagonal entries of (2E H - M_k): Xd[k][i]
        cplx Xd[3][3];
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int i = 0; i < 3; i++) Xd[k][i] = cmake(X.m[i][i].re - M[k], X.m[i][i].im);
#define HMM(i_, j_, k_) (((i_) == (j_)) ? Xd[k_][i_] : X.m[i_][j_])
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                cplx p0 = cmul(HMM(i, 0, 1), HMM(0, j, 2));
                p0 = cadd(p0, cmul(HMM(i, 1, 1), HMM(1, j, 2)));
                p0 = cadd(p0, cmul(HMM(i, 2, 1), HMM(2, j, 2)));
                cplx p1 = cmul(HMM(i, 0, 2), HMM(0, j, 0));
                p1 = cadd(p1, cmul(HMM(i, 1, 2), HMM(1, j, 0)));
                p1 = cadd(p1, cmul(HMM(i, 2, 2), HMM(2, j, 0)));
                cplx p2 = cmul(HMM(i, 0, 0), HMM(0, j, 1));
                p2 = cadd(p2, cmul(HMM(i, 1, 0), HMM(1, j, 1)));
                p2 = cadd(p2, cmul(HMM(i, 2, 0), HMM(2, j, 1)));
                if (FAST) {
                    p0 = cmake(p0.re * inv0, p0.im * inv0);
                    p1 = cmake(p1.re * inv1, p1.im * inv1);
                    p2 = cmake(p2.re * inv2, p2.im * inv2);
                } else {
                    p0 = cmake(p0.re / den0, p0.im / den0);
                    p1 = cmake(p1.re / den1, p1.im / den1);
                    p2 = cmake(p2.re / den2, p2.im / den2);
                }
                cplx acc = cmul(ph[0], p0);
                acc = cadd(acc, cmul(ph[1], p1));
                acc = cadd(acc, cmul(ph[2], p2));
                A.m[i][j] = acc;
            }
#undef HMM
    } else {
        // FAST (event-mode kernel): the 27 complex quotients and the Newton steps of the eigenvalues multiply by
        // reciprocals with one real division each -- 18 divisions per layer instead of 126 (a complex quotient by
        // Smith's rule is three); equal to a few ulp
        cplx lam[3], M[3];
        eigvals3_general<FAST>(Hf, lam);
#pragma unroll
        for (int k = 0; k < 3; k++) M[k] = cscale(two_e, lam[k]);
        cplx ph[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            cplx arg = cscale(hbar_c_factor, cscale(L_over_E, cscale(-1.0, M[k])));
            // arg * 1j = (-arg.im, arg.re); exp
            double l = exp(-arg.im);
            double s, c;
            sincos(arg.re, &s, &c);
            ph[k] = cmake(l * c, l * s);
        }
        cplx den0 = cmul(csub(M[0], M[1]), csub(M[0], M[2]));
        cplx den1 = cmul(csub(M[1], M[2]), csub(M[1], M[0]));
        cplx den2 = cmul(csub(M[2], M[0]), csub(M[2], M[1]));
        const cplx zero_c = cmake(0.0, 0.0);
        const cplx inv0 = FAST ? crecip(den0) : zero_c, inv1 = FAST ? crecip(den1) : zero_c, inv2 = FAST ? crecip(den2) : zero_c;
        cplx Xd[3][3];
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int i = 0; i < 3; i++) Xd[k][i] = csub(X.m[i][i], M[k]);
#define HMM(i_, j_, k_) (((i_) == (j_)) ? Xd[k_][i_] : X.m[i_][j_])
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                cplx p0 = cmul(HMM(i, 0, 1), HMM(0, j, 2));
                p0 = cadd(p0, cmul(HMM(i, 1, 1), HMM(1, j, 2)));
                p0 = cadd(p0, cmul(HMM(i, 2, 1), HMM(2, j, 2)));
                cplx p1 = cmul(HMM(i, 0, 2), HMM(0, j, 0));
                p1 = cadd(p1, cmul(HMM(i, 1, 2), HMM(1, j, 0)));
                p1 = cadd(p1, cmul(HMM(i, 2, 2), HMM(2, j, 0)));
                cplx p2 = cmul(HMM(i, 0, 0), HMM(0, j, 1));
                p2 = cadd(p2, cmul(HMM(i, 1, 0), HMM(1, j, 1)));
                p2 = cadd(p2, cmul(HMM(i, 2, 0), HMM(2, j, 1)));
                if (FAST) {
                    p0 = cmul(p0, inv0);
                    p1 = cmul(p1, inv1);
                    p2 = cmul(p2, inv2);
                } else {
                    p0 = cdiv(p0, den0);
                    p1 = cdiv(p1, den1);
                    p2 = cdiv(p2, den2);
                }
                cplx acc = cmul(ph[0], p0);
                acc = cadd(acc, cmul(ph[1], p1));
                acc = cadd(acc, cmul(ph[2], p2));
                A.m[i][j] = acc;
            }
#undef HMM
    }
}

// Event-mode layer matrix WITH decay (complex eigenvalues), polynomial form.  The reference builds
//   A = sum_k e_k (X - M_a)(X - M_b) / ((M_k - M_a)(M_k - M_b)),   e_k = exp(-i M_k L/E 2.534),  X = 2E U^dagger H U
// (numba_osc_kernels.py:432-467, 834-872: three full matrix products and 27 complex quotients per layer).
// Expanded in powers of X -- the same Lagrange interpolation of exp on the spectrum, i.e. the same matrix --
//   A = c0 I + c1 X + c2 X^2,  t_k = e_k / den_k,  c2 = sum t_k,  c1 = -sum t_k (M_a + M_b),  c0 = sum t_k M_a M_b
// it costs ONE matrix product; X itself is assembled from the host-prepared mass-basis images (X0 + 2E a XV +
// 2E XL, as the planned grid form does) instead of two products with U, and its eigenvalues are taken from X
// directly (similar to 2E H: the same spectrum).  About half the instructions of layer_amplitude<true, true>
// and far fewer live registers (no three sets of shifted diagonals, no 27 partial products); the result differs
// from it by rounding (<= 1e-13 on the probabilities, tests/test_gpu_prob3_variants.py, decay goldens).
__device__ __forceinline__ void layer_amplitude_decay_poly(const Prob3Side &S, double energy, double rho,
                                                           double baseline, mat3 &A) {
    const double tworttwoGf = 1.52588e-4;
    const double two_e = 2.0 * energy;
    const double ka = two_e * (S.a_sign * (0.5 * rho * tworttwoGf));
    mat3 X;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            X.m[i][j] = cadd(S.X0.m[i][j], cadd(cscale(ka, S.XV.m[i][j]), cscale(two_e, S.XL.m[i][j])));
    cplx M[3];
    eigvals3_general<true>(X, M);
    const double lf = (baseline * fast_rcp(energy)) * 2.534;
    // exp(-i X lf) = c0 + c1 X + c2 X^2: the interpolation polynomial of exp(-i m lf) in the eigenvalues of X.
    cplx e[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        // exp(-i M_k lf) = exp(M_k.im lf) (cos(M_k.re lf) - i sin(M_k.re lf))
        const double l = exp(M[k].im * lf);
        double sn, cs;
        sincos(-M[k].re * lf, &sn, &cs);
        e[k] = cmake(l * cs, l * sn);
    }
    const cplx d01 = csub(M[0], M[1]), d02 = csub(M[0], M[2]), d12 = csub(M[1], M[2]);
    const double g01 = cabs2(d01), g02 = cabs2(d02), g12 = cabs2(d12);
    cplx c0, c1, c2;
    // LAGRANGE form (terms e_k / prod(m_k - m_j)): the cheapest, and fine while the eigenvalues are separated.  It divides by the
    // gap of a nearly degenerate pair and loses log10(scale / gap) digits in the sum of its two large terms (found by
    // scripts/dev/fuzz_prob3.py: dm21 = 0 leaves the light states 5e-9 eV^2 apart, 3e-10 on the probabilities).  A wavefront in
    // which some lane has a pair closer than 1e-3 of its largest gap takes the NEWTON form instead, the closest pair (a, b) first:
    //     f[a] + f[a,b] (X - a) + f[a,b,c] (X - a)(X - b),   f[a,b] = e_a (-i lf) phi(z),  phi(z) = (exp z - 1) / z  by its series
    // for |z| = |lf (b - a)| < 1/16, the plain quotient otherwise.
    const double gmin = fmin(g01, fmin(g02, g12)), gmax = fmax(g01, fmax(g02, g12));
    if (!__any(gmin < 1e-6 * gmax)) {
        const cplx t0 = cmul(e[0], crecip(cmul(d01, d02)));
        const cplx t1 = cmul(e[1], crecip(cscale(-1.0, cmul(d12, d01))));
        const cplx t2 = cmul(e[2], crecip(cmul(d02, d12)));
        c2 = cadd(cadd(t0, t1), t2);
        c1 = cscale(-1.0, cadd(cadd(cmul(t0, cadd(M[1], M[2])), cmul(t1, cadd(M[2], M[0]))), cmul(t2, cadd(M[0], M[1]))));
        c0 = cadd(cadd(cmul(t0, cmul(M[1], M[2])), cmul(t1, cmul(M[2], M[0]))), cmul(t2, cmul(M[0], M[1])));
    } else {
        const bool p01 = g01 <= g02 && g01 <= g12, p02 = !p01 && g02 <= g12;
        const cplx ma = p01 ? M[0] : (p02 ? M[0] : M[1]), mb = p01 ? M[1] : M[2], mc = p01 ? M[2] : (p02 ? M[1] : M[0]);
        const cplx ea = p01 ? e[0] : (p02 ? e[0] : e[1]), eb = p01 ? e[1] : e[2], ec = p01 ? e[2] : (p02 ? e[1] : e[0]);
        const cplx dab = csub(mb, ma);
        const cplx z = cmul(cmake(0.0, -lf), dab);            // e_b = e_a exp(z)
        const bool small = cabs2(z) < 1.0 / 256.0;            // nine terms of phi give 1e-17 there
        cplx phi = cmake(1.0, 0.0), term = cmake(1.0, 0.0);
#pragma unroll
        for (int n = 2; n <= 10; n++) {
            term = cscale(1.0 / (double)n, cmul(term, z));
            phi = cadd(phi, term);
        }
        const cplx quot = cmul(csub(eb, ea), crecip(dab));   // (of an exactly degenerate pair: not taken)
        const cplx f01 = small ? cmul(cmul(ea, cmake(0.0, -lf)), phi) : quot;
        const cplx f12 = cmul(csub(ec, eb), crecip(csub(mc, mb)));
        c2 = cmul(csub(f12, f01), crecip(csub(mc, ma)));
        c1 = csub(f01, cmul(c2, cadd(ma, mb)));
        c0 = cadd(csub(ea, cmul(f01, ma)), cmul(c2, cmul(ma, mb)));
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            cplx x2 = cmul(X.m[i][0], X.m[0][j]);
            x2 = cadd(x2, cmul(X.m[i][1], X.m[1][j]));
            x2 = cadd(x2, cmul(X.m[i][2], X.m[2][j]));
            cplx acc = cadd(cmul(c1, X.m[i][j]), cmul(c2, x2));
            if (i == j) acc = cadd(acc, c0);
            A.m[i][j] = acc;
        }
}

// ---------------------------------------------------------------------------
// Two-stage form used on (E x coszen) grids.  Everything in
// get_transition_matrix except the three phases exp(-i M_k L/E 2.534) depends
// only on (energy, density): the eigenvalues M_k and the 27 quotients
// product[i][j][k] / ((M_k-M_j)(M_k-M_l)) (numba_osc_kernels.py:432-467,
// 834-872).  Stage A evaluates them once per (E, shell density) -- a few
// thousand times per evaluation instead of once per node and crossed layer --
// and stage B assembles A = sum_k phase_k * Q_k per layer.  Same operations on
// the same operands as layer_amplitude() except that the quotients are formed with
// one reciprocal per eigenvalue: equal to rounding (~1e-16), not bit for bit.
constexpr int PROB3_NF = 60;  // fields per record, decay form: M[3] (re,im) + Q[3][3][3] (re,im)
constexpr int PROB3_NF_REDUCED = 18;  // without decay (see eigen_terms)

// field(f) = value callback; f in [0, PROB3_NF)
// LRI = false (event mode, chosen on the host when lri_pot is all zeros -- every configuration without a long-range
// potential): the XL terms are not formed.  2E XL = 0 added nothing (x + 0 = x, and a fused ka XV + 0 rounds once like
// the product): same bits; nine multiply-adds and eighteen scalar registers less per layer matrix.
template <bool DECAY, bool FAST = false, bool LRI = true, class StoreFn>
__device__ __forceinline__ void eigen_terms(const Prob3Side &S, const double (&dm)[3][3],
                                            const int32_t (&vac_order)[3], double energy, double rho,
                                            const StoreFn &store) {
    const double tworttwoGf = 1.52588e-4;
    double a = 0.5 * rho * tworttwoGf;
    double sa = S.a_sign * a;
    double two_e = 2.0 * energy;
    const double ka = two_e * sa;
    if (!DECAY) {
        // ---- REDUCED form.  X = 2E.U^dagger.H.U is linear in the layer's potential a and is assembled
        // from three matrices prepared once on the host (X0 + (2E a) XV + 2E XL) instead of two 3x3
        // complex products per (E, rho).  Without decay X is Hermitian:
        //     X = [[x0, u, v], [conj u, x1, w], [conj v, conj w, x2]],
        // nine real numbers, and everything below works on them.
        const double x0 = S.X0.m[0][0].re + (LRI ? ka * S.XV.m[0][0].re + two_e * S.XL.m[0][0].re : ka * S.XV.m[0][0].re);
        const double x1 = S.X0.m[1][1].re + (LRI ? ka * S.XV.m[1][1].re + two_e * S.XL.m[1][1].re : ka * S.XV.m[1][1].re);
        const double x2 = S.X0.m[2][2].re + (LRI ? ka * S.XV.m[2][2].re + two_e * S.XL.m[2][2].re : ka * S.XV.m[2][2].re);
        const cplx u = cadd(S.X0.m[0][1], LRI ? cadd(cscale(ka, S.XV.m[0][1]), cscale(two_e, S.XL.m[0][1])) : cscale(ka, S.XV.m[0][1]));
        const cplx v = cadd(S.X0.m[0][2], LRI ? cadd(cscale(ka, S.XV.m[0][2]), cscale(two_e, S.XL.m[0][2])) : cscale(ka, S.XV.m[0][2]));
        const cplx w = cadd(S.X0.m[1][2], LRI ? cadd(cscale(ka, S.XV.m[1][2]), cscale(two_e, S.XL.m[1][2])) : cscale(ka, S.XV.m[1][2]));
        const double uu = u.re * u.re + u.im * u.im, vv = v.re * v.re + v.im * v.im,
                     ww = w.re * w.re + w.im * w.im;
        const cplx vw = cmake(v.re * w.re + v.im * w.im, v.im * w.re - v.re * w.im);   // v conj(w)
        const cplx uw = cmul(u, w);
        const cplx uv = cmake(u.re * v.re + u.im * v.im, u.re * v.im - u.im * v.re);   // conj(u) v
        // Eigenvalues of X = 2E x those of the flavour-basis Hamiltonian: get_dms' trigonometric
        // solution (numba_osc_kernels.py:776-815) on the characteristic polynomial of X itself --
        // the same invariants (trace, second invariant, determinant) from the mass-basis entries,
        // which are already at hand for the projectors.  Root order as in the reference (the
        // coefficients scale by powers of 2E > 0, the angle is unchanged).
        double mu[3];
        {
            const double c2 = -(x0 + x1 + x2);
            const double c1 = x0 * (x1 + x2) + x1 * x2 - uu - ww - vv;
            const double c0 = x0 * ww + x1 * vv + x2 * uu - 2.0 * (uw.re * v.re + uw.im * v.im) - x0 * x1 * x2;
            double p = c2 * c2 - 3.0 * c1;
            p = fmax(0.0, p);
            const double q = -13.5 * c0 - c2 * (c2 * c2) + 4.5 * c1 * c2;
            double tmp = 27 * (0.25 * (c1 * c1) * (p - c1) + c0 * (q + 6.75 * c0));
            tmp = fmax(0.0, tmp);
            const double b = (2.0 / 3.0) * (FAST ? fast_sqrt(p) : sqrt(p));
            double sn, cs;
            if (FAST) {
                sincos_third_angle(fast_sqrt(tmp), q, &sn, &cs);
            } else {
                const double res = atan2(sqrt(tmp), q) * (1.0 / 3.0);
                sincos_phase(res, &sn, &cs);
            }
            const double ca = -0.5, sb = 0.86602540378443864676;  // cos, sin of 2pi/3
            const double shift = two_e * dm[0][0] - c2 * (1.0 / 3.0);
            mu[0] = b * (cs * ca - sn * sb) + shift;
            mu[1] = b * (cs * ca + sn * sb) + shift;
            mu[2] = b * cs + shift;
        }
        double Mr[3];
#pragma unroll
        for (int k = 0; k < 3; k++)  // vacuum ordering, resolved on the host (Prob3Consts::vac_order)
            Mr[k] = vac_order[k] == 0 ? mu[0] : (vac_order[k] == 1 ? mu[1] : mu[2]);
        // A unit phase common to a layer's amplitude drops out of every probability (the chain
        // product only collects a global phase), so the layer matrix may be taken as
        //     A' = exp(+i Mbar t) A = exp(-i (H - tr H / 3) t),  Mbar = (M_0+M_1+M_2)/3,
        // which is in SU(3): its third row is the conjugate cross product of the first two and is
        // neither computed nor stored.  With the projectors summing to the identity,
        //     A' = e_0 + (e_1 - e_0) Q_1 + (e_2 - e_0) Q_2,  e_k = exp(-i G_k t),  e_0 = conj(e_1 e_2),
        // G_k = M_k - Mbar.  The projectors Q_k = (X - M_a)(X - M_b) / den_k are Hermitian like X:
        // their rows 0 and 1 are the real entries Q00, Q11 and the complex entries Q01, Q02, Q12
        // (Q10 = conj Q01).  Fields: G_1, G_2, then for Q_1 and Q_2
        // (Q00, Q11, Re Q01, Im Q01, Re Q02, Im Q02, Re Q12, Im Q12): PROB3_NF_REDUCED = 18.
        const double d1 = Mr[1] - Mr[0], d2 = Mr[2] - Mr[0];
        store(0, (2.0 * d1 - d2) * (1.0 / 3.0));
        store(1, (2.0 * d2 - d1) * (1.0 / 3.0));
#pragma unroll
        for (int k = 1; k < 3; k++) {
            // Q_1 = (X - M_2)(X - M_0) / den_1,  Q_2 = (X - M_0)(X - M_1) / den_2
            const double sa_ = k == 1 ? Mr[2] : Mr[0], sb_ = k == 1 ? Mr[0] : Mr[1];
            const double den = k == 1 ? (Mr[1] - Mr[2]) * (Mr[1] - Mr[0]) : (Mr[2] - Mr[0]) * (Mr[2] - Mr[1]);
            const double inv = FAST ? fast_rcp(den) : 1.0 / den;   // one reciprocal per eigenvalue
            const double a0 = x0 - sa_, a1 = x1 - sa_, b0 = x0 - sb_, b1 = x1 - sb_, b2 = x2 - sb_;
            const int base = 2 + 8 * (k - 1);
            store(base + 0, (a0 * b0 + uu + vv) * inv);
            store(base + 1, (uu + a1 * b1 + ww) * inv);
            const double s01 = a0 + b1, s02 = a0 + b2, s12 = a1 + b2;
            store(base + 2, (u.re * s01 + vw.re) * inv);
            store(base + 3, (u.im * s01 + vw.im) * inv);
            store(base + 4, (v.re * s02 + uw.re) * inv);
            store(base + 5, (v.im * s02 + uw.im) * inv);
            store(base + 6, (w.re * s12 + uv.re) * inv);
            store(base + 7, (w.im * s12 + uv.im) * inv);
        }
        return;
    }
    // ---- general (decay) form
    double one_over_two_e = 0.5 / energy;
    mat3 Hf;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            cplx hm = cscale(sa, S.V.m[i][j]);
            hm.re = hm.re + S.lri[i][j];
            Hf.m[i][j] = cadd(cscale(one_over_two_e, S.Hvd.m[i][j]), hm);
        }
    mat3 X;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            X.m[i][j] = cadd(S.X0.m[i][j], cadd(cscale(ka, S.XV.m[i][j]), cscale(two_e, S.XL.m[i][j])));

    cplx M[3], den[3];
    {
        cplx lam[3];
        eigvals3_general(Hf, lam);
#pragma unroll
        for (int k = 0; k < 3; k++) M[k] = cscale(two_e, lam[k]);
        den[0] = cmul(csub(M[0], M[1]), csub(M[0], M[2]));
        den[1] = cmul(csub(M[1], M[2]), csub(M[1], M[0]));
        den[2] = cmul(csub(M[2], M[0]), csub(M[2], M[1]));
    }
    // Record layout: M[3] (re, im) then Q_k[i][j] (re, im), k fastest: 60 fields.
#pragma unroll
    for (int k = 0; k < 3; k++) {
        store(2 * k, M[k].re);
        store(2 * k + 1, M[k].im);
    }
    cplx Xd[3][3];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int i = 0; i < 3; i++) Xd[k][i] = csub(X.m[i][i], M[k]);
#define HMM(i_, j_, k_) (((i_) == (j_)) ? Xd[k_][i_] : X.m[i_][j_])
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            cplx p0 = cmul(HMM(i, 0, 1), HMM(0, j, 2));
            p0 = cadd(p0, cmul(HMM(i, 1, 1), HMM(1, j, 2)));
            p0 = cadd(p0, cmul(HMM(i, 2, 1), HMM(2, j, 2)));
            cplx p1 = cmul(HMM(i, 0, 2), HMM(0, j, 0));
            p1 = cadd(p1, cmul(HMM(i, 1, 2), HMM(1, j, 0)));
            p1 = cadd(p1, cmul(HMM(i, 2, 2), HMM(2, j, 0)));
            cplx p2 = cmul(HMM(i, 0, 0), HMM(0, j, 1));
            p2 = cadd(p2, cmul(HMM(i, 1, 0), HMM(1, j, 1)));
            p2 = cadd(p2, cmul(HMM(i, 2, 0), HMM(2, j, 1)));
            p0 = cdiv(p0, den[0]);
            p1 = cdiv(p1, den[1]);
            p2 = cdiv(p2, den[2]);
            const int base = 6 + 6 * (3 * i + j);
            store(base + 0, p0.re); store(base + 1, p0.im);
            store(base + 2, p1.re); store(base + 3, p1.im);
            store(base + 4, p2.re); store(base + 5, p2.im);
        }
#undef HMM
}

// A = sum_k exp(-i M_k L/E 2.534) Q_k from a stage-A record (load(f) reads field f); in the
// reduced (no-decay) form rows 0 and 1 of the SU(3) matrix A' (see eigen_terms), row 2 of A
// is left untouched
template <bool DECAY, class LoadFn>
__device__ __forceinline__ void amplitude_from_terms(const LoadFn &load, double L_over_E, mat3 &A) {
    const double hbar_c_factor = 2.534;
    if (!DECAY) {
        cplx e[3];
#pragma unroll
        for (int k = 1; k < 3; k++) {
            double arg = (-load(k - 1)) * L_over_E * hbar_c_factor;
            double sn, cs;
            sincos_phase(arg, &sn, &cs);
            e[k] = cmake(cs, sn);
        }
        e[0] = cmul(e[1], e[2]);
        e[0].im = -e[0].im;
        const cplx f1 = csub(e[1], e[0]), f2 = csub(e[2], e[0]);
        // diagonal entries: real projector entries
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const double q1 = load(2 + i), q2 = load(10 + i);
            A.m[i][i] = cmake(e[0].re + f1.re * q1 + f2.re * q2, e[0].im + f1.im * q1 + f2.im * q2);
        }
        // off-diagonal entries (0,1), (0,2), (1,2); (1,0) from the conjugate of Q01
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const cplx q1 = cmake(load(4 + 2 * t), load(5 + 2 * t));
            const cplx q2 = cmake(load(12 + 2 * t), load(13 + 2 * t));
            const cplx acc = cadd(cmul(f1, q1), cmul(f2, q2));
            if (t == 0) {
                A.m[0][1] = acc;
                A.m[1][0] = cadd(cmul(f1, cmake(q1.re, -q1.im)), cmul(f2, cmake(q2.re, -q2.im)));
            } else if (t == 1) {
                A.m[0][2] = acc;
            } else {
                A.m[1][2] = acc;
            }
        }
        return;
    }
    cplx ph[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        cplx Mk = cmake(load(2 * k), load(2 * k + 1));
        cplx arg = cscale(hbar_c_factor, cscale(L_over_E, cscale(-1.0, Mk)));
        double l = exp(-arg.im);
        double s, c;
        sincos(arg.re, &s, &c);
        ph[k] = cmake(l * c, l * s);
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int base = 6 + 6 * (3 * i + j);
            cplx acc = cmul(ph[0], cmake(load(base + 0), load(base + 1)));
            acc = cadd(acc, cmul(ph[1], cmake(load(base + 2), load(base + 3))));
            acc = cadd(acc, cmul(ph[2], cmake(load(base + 4), load(base + 5))));
            A.m[i][j] = acc;
        }
}

// third row of an SU(3) matrix from its first two: conj(row0 x row1)
__device__ __forceinline__ void su3_complete(mat3 &A) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int k = (j + 1) % 3, l = (j + 2) % 3;
        cplx c = csub(cmul(A.m[0][k], A.m[1][l]), cmul(A.m[0][l], A.m[1][k]));
        A.m[2][j] = cmake(c.re, -c.im);
    }
}

// Layer-matrix cache of the reference (numba_osc_kernels.py:230-249): layer i
// re-uses the matrix of the LAST earlier layer j with |drho|<1e-5 and
// |ddist|<1e-5.  Instead of storing up to 120 matrices per thread, the chain of
// matches is followed to the layer whose matrix was actually computed and that
// layer's (rho, dist) are returned: A(i) == A(src(i)) bit for bit.
template <class LayerFn>
__device__ __forceinline__ void resolve_layer(const LayerFn &layer, int i, double &rho,
                                              double &dist) {
    int cur = i;
    layer(cur, rho, dist);
    while (true) {
        int found = -1;
        for (int j = 0; j < cur; j++) {
            double rj, dj;
            layer(j, rj, dj);
            if (dj > 0.0 && fabs(rj - rho) < 1e-5 && fabs(dj - dist) < 1e-5) found = j;
        }
        if (found < 0) break;
        cur = found;
        layer(cur, rho, dist);
    }
}

// osc_probs_layers_kernel (numba_osc_kernels.py:121-345) for one element.
// `layer(i, rho, dist)` yields the i-th layer of this element's path.
// P receives P[init][final] (9 doubles).
template <bool DECAY, class LayerFn>
__device__ __forceinline__ void propagate_element(const Prob3Side &S, const double (&dm)[3][3],
                                                  double energy, int n_layers,
                                                  const LayerFn &layer, double (&P)[9]) {
    mat3 T;
    bool first = true;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) T.m[i][j] = cmake(0.0, 0.0);
    for (int l = 0; l < n_layers; l++) {
        double rho, dist;
        layer(l, rho, dist);
        if (dist > 0.0) {
            resolve_layer(layer, l, rho, dist);
            mat3 A;
            layer_amplitude<DECAY>(S, dm, energy, rho, dist, A);
            if (first) {
                T = A;
                first = false;
            } else {
                mat3 t2;
                mat_mul(A, T, t2);
                T = t2;
            }
        }
    }
    // flavour basis (:326-328) and probabilities (:331-345)
    mat3 t2, Tf;
    mat_mul(T, S.Ud, t2);
    mat_mul(S.U, t2, Tf);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            P[3 * i + j] = Tf.m[j][i].re * Tf.m[j][i].re + Tf.m[j][i].im * Tf.m[j][i].im;
}

// Index form of resolve_layer: the layer whose matrix the reference's cache would hand out
// for layer i (i itself if there is no earlier match).
template <class LayerFn>
__device__ __forceinline__ int resolve_layer_index(const LayerFn &layer, int i) {
    int cur = i;
    double rho, dist;
    layer(cur, rho, dist);
    while (true) {
        int found = -1;
        for (int j = 0; j < cur; j++) {
            double rj, dj;
            layer(j, rj, dj);
            if (dj > 0.0 && fabs(rj - rho) < 1e-5 && fabs(dj - dist) < 1e-5) found = j;
        }
        if (found < 0) break;
        cur = found;
        layer(cur, rho, dist);
    }
    return cur;
}

// Event-mode form of propagate_element for a path through the Earth that goes IN through
// layers 0..mid-1, crosses the innermost layer `mid` and comes OUT through mid+1..n-1, the
// out-going layer mid+s being the mirror of the in-going layer mid-s (layers.py:118-158).
// The product is built from the middle outwards,  T <- A_out(s) . T . A_in(s),  so that a
// mirrored pair needs ONE amplitude evaluation whenever the reference's cache would hand
// the in-going layer's matrix to the out-going one (`resolve_layer_index`); otherwise the
// out-going matrix is computed from its own resolved (rho, length), as propagate_element
// does.  Same matrices as the sequential form, associated differently: equal to rounding.
// mid < 0: no mirror structure (down-going paths), plain sequential product.
// Without decay the layer matrices are formed in the reduced SU(3) form of the planned grid
// path (eigen_terms / amplitude_from_terms: host-prepared mass-basis matrices and vacuum
// ordering, two projectors, two sincos, third row completed) -- about half the instructions
// of layer_amplitude; a layer matrix then differs from the reference's by a unit phase, which
// no probability sees.
// `src(l)` = resolve_layer_index(layer, l), tabulated by the caller once per path (each
// resolution is a scan over the earlier layers; three per step added up to a third of the
// event kernel).
template <bool DECAY, class LayerFn, class SrcFn>
__device__ __forceinline__ void propagate_path_nested(const Prob3Side &S, const double (&dm)[3][3],
                                                      const int32_t (&vac_order)[3],
                                                      double energy, int n_layers, int mid,
                                                      const LayerFn &layer, const SrcFn &src,
                                                      double (&P)[9]) {

    mat3 T;
    bool have = false;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) T.m[i][j] = cmake(0.0, 0.0);
    auto amplitude = [&](int l, mat3 &A) {  // matrix the reference uses for layer l
        double rho, dist;
        layer(src(l), rho, dist);
        if (DECAY) {
            layer_amplitude_decay_poly(S, energy, rho, dist, A);
        } else {
            double rec[PROB3_NF_REDUCED];
            auto store = [&](int f, double v) { rec[f] = v; };
            eigen_terms<false>(S, dm, vac_order, energy, rho, store);
            auto load = [&](int f) { return rec[f]; };
            amplitude_from_terms<false>(load, dist / energy, A);
            su3_complete(A);
        }
    };
    auto left = [&](const mat3 &A) {   // T <- A . T  (A later on the path)
        if (have) { mat3 t2; mat_mul(A, T, t2); T = t2; } else { T = A; have = true; }
    };
    auto right = [&](const mat3 &A) {  // T <- T . A  (A earlier on the path)
        if (have) { mat3 t2; mat_mul(T, A, t2); T = t2; } else { T = A; have = true; }
    };
    if (mid < 0) {
        for (int l = 0; l < n_layers; l++) {
            double rho, dist;
            layer(l, rho, dist);
            if (dist > 0.0) { mat3 A; amplitude(l, A); left(A); }
        }
    } else {
        {
            double rho, dist;
            layer(mid, rho, dist);
            if (dist > 0.0) { mat3 A; amplitude(mid, A); left(A); }
        }
        for (int s = 1; mid - s >= 0 || mid + s < n_layers; s++) {
            const int li = mid - s, lo = mid + s;
            double rho_i = 0.0, d_i = 0.0, rho_o = 0.0, d_o = 0.0;
            if (li >= 0) layer(li, rho_i, d_i);
            if (lo < n_layers) layer(lo, rho_o, d_o);
            mat3 A;
            const bool in_ok = li >= 0 && d_i > 0.0, out_ok = lo < n_layers && d_o > 0.0;
            if (in_ok) {
                amplitude(li, A);
                right(A);
            }
            if (out_ok) {
                // the reference re-uses the in-going layer's matrix iff the cache chain of
                // the out-going layer ends where the in-going layer's does
                const bool same = in_ok && src(lo) == src(li);
                if (!same) amplitude(lo, A);
                left(A);
            }
        }
    }
    mat3 t2, Tf;
    mat_mul(T, S.Ud, t2);
    mat_mul(S.U, t2, Tf);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            P[3 * i + j] = Tf.m[j][i].re * Tf.m[j][i].re + Tf.m[j][i].im * Tf.m[j][i].im;
}

// Event-mode variant with the running product T in LDS ([18][lanes], one column of doubles per lane):
// T is not needed while a layer matrix is formed, and a product  A.T  acts on T's columns (T.A on its
// rows) independently, so it is done in place one column (row) at a time -- the registers hold one layer
// matrix and six numbers of T instead of three matrices.
struct TLds {
    double *base;   // &lds[lane]
    int stride;     // lanes
    __device__ __forceinline__ cplx get(int i, int j) const {
        return cmake(base[(6 * i + 2 * j) * stride], base[(6 * i + 2 * j + 1) * stride]);
    }
    __device__ __forceinline__ void put(int i, int j, cplx v) {
        base[(6 * i + 2 * j) * stride] = v.re;
        base[(6 * i + 2 * j + 1) * stride] = v.im;
    }
};

template <bool DECAY, class LayerFn, class SrcFn>
__device__ __forceinline__ void propagate_path_nested_lds(const Prob3Side &S, const double (&dm)[3][3],
                                                          const int32_t (&vac_order)[3],
                                                          double energy, int n_layers, int mid,
                                                          const LayerFn &layer, const SrcFn &src,
                                                          TLds T, double (&P)[9]) {

    bool have = false;
    auto amplitude = [&](int l, mat3 &A) {
        double rho, dist;
        layer(src(l), rho, dist);
        if (DECAY) {
            layer_amplitude_decay_poly(S, energy, rho, dist, A);
        } else {
            double rec[PROB3_NF_REDUCED];
            auto store = [&](int f, double v) { rec[f] = v; };
            eigen_terms<false>(S, dm, vac_order, energy, rho, store);
            auto load = [&](int f) { return rec[f]; };
            amplitude_from_terms<false>(load, dist / energy, A);
            su3_complete(A);
        }
    };
    auto set = [&](const mat3 &A) {
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) T.put(i, j, A.m[i][j]);
        have = true;
    };
    auto left = [&](const mat3 &A) {   // T <- A . T, column by column
        if (!have) { set(A); return; }
#pragma unroll 1
        for (int j = 0; j < 3; j++) {
            const cplx t0 = T.get(0, j), t1 = T.get(1, j), t2 = T.get(2, j);
#pragma unroll
            for (int i = 0; i < 3; i++) {
                cplx acc = cmul(A.m[i][0], t0);
                acc = cadd(acc, cmul(A.m[i][1], t1));
                acc = cadd(acc, cmul(A.m[i][2], t2));
                T.put(i, j, acc);
            }
        }
    };
    auto right = [&](const mat3 &A) {  // T <- T . A, row by row
        if (!have) { set(A); return; }
#pragma unroll 1
        for (int i = 0; i < 3; i++) {
            const cplx t0 = T.get(i, 0), t1 = T.get(i, 1), t2 = T.get(i, 2);
#pragma unroll
            for (int j = 0; j < 3; j++) {
                cplx acc = cmul(t0, A.m[0][j]);
                acc = cadd(acc, cmul(t1, A.m[1][j]));
                acc = cadd(acc, cmul(t2, A.m[2][j]));
                T.put(i, j, acc);
            }
        }
    };
    if (mid < 0) {
        for (int l = 0; l < n_layers; l++) {
            double rho, dist;
            layer(l, rho, dist);
            if (dist > 0.0) { mat3 A; amplitude(l, A); left(A); }
        }
    } else {
        {
            double rho, dist;
            layer(mid, rho, dist);
            if (dist > 0.0) { mat3 A; amplitude(mid, A); left(A); }
        }
        for (int s = 1; mid - s >= 0 || mid + s < n_layers; s++) {
            const int li = mid - s, lo = mid + s;
            double rho_i = 0.0, d_i = 0.0, rho_o = 0.0, d_o = 0.0;
            if (li >= 0) layer(li, rho_i, d_i);
            if (lo < n_layers) layer(lo, rho_o, d_o);
            mat3 A;
            const bool in_ok = li >= 0 && d_i > 0.0, out_ok = lo < n_layers && d_o > 0.0;
            if (in_ok) {
                amplitude(li, A);
                right(A);
            }
            if (out_ok) {
                const bool same = in_ok && src(lo) == src(li);
                if (!same) amplitude(lo, A);
                left(A);
            }
        }
    }
    mat3 Tm, t2, Tf;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Tm.m[i][j] = have ? T.get(i, j) : cmake(0.0, 0.0);
    mat_mul(Tm, S.Ud, t2);
    mat_mul(S.U, t2, Tf);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            P[3 * i + j] = Tf.m[j][i].re * Tf.m[j][i].re + Tf.m[j][i].im * Tf.m[j][i].im;
}

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


}  // namespace pisa
