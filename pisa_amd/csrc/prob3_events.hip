// prob3_events.hip -- event-by-event oscillation probabilities (configs C2 / C5): every event's
// path through the Earth is rebuilt from its coszen with the PREM shell table in LDS
// (layers.py:86-159 evaluated lazily; no densities/distances[N][L] arrays in HBM), the
// reference's layer-matrix cache (numba_osc_kernels.py:230-249) is resolved once per path, and
// mirrored layers share one amplitude.
//
// This translation unit is compiled with -ffp-contract=fast (the Makefile's FLAGS_prob3_events):
// the kernel is bound by the issue rate of dependent fp64 instructions at 2 waves / SIMD, and
// fusing multiply-add pairs removes a third of them (0.289 -> 0.255 ms per 1e6 events).  The
// results move by rounding only (LLH of the 1e6-event workload: 4e-13 relative) and stay inside
// the prob3 tolerance of the oracle comparison (tests/test_gpu_kernels.py).  The array / grid
// kernels and calc_layers_kernel (prob3.hip) keep the reference's unfused operation order.
#include <string.h>

#include "common.hpp"
#include "prob3_device.hpp"
#include "prob3_paths.hpp"

namespace pisa {

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

// SIDE (0 nu / 1 nubar) is a template parameter: indexing the by-value constants with a run-time
// side made the compiler copy them to scratch (2.3 KB per lane) and read them back into VGPRs
//
// STAGED = false (every pair of crossable shell densities differs by >= 1e-5, checked on the host):
// the reference's cache can then only ever hand an out-going layer the matrix of its own mirror
// image, so nothing is tabulated -- a segment's length is recomputed from the shell radii where it
// is needed (two square roots) and the per-lane path tables (10 bytes per segment and lane of LDS)
// disappear: 0.220 -> 0.209 ms per 1e6 events.  STAGED = true is the general form (path, shell and
// cache source of every segment staged in LDS).
// WAVES = wavefronts per SIMD the registers are allocated for.  With everything in registers two is
// where the allocation has no spills (201-221 VGPRs; builds for three and four -- 168 / 128 VGPRs, 52-61 /
// 141-158 spilled registers -- ran at 0.236 and 0.430 ms).  WAVES = 3 (direct form, default) keeps the
// running product T in LDS, [18][64] doubles per wavefront, and multiplies in place -- A.T acts on the
// columns of T, T.A on its rows, one at a time (propagate_path_nested_lds): one layer matrix and six
// numbers of T in registers instead of three matrices, 8 spilled registers at 168, 0.211 -> 0.202 ms;
// four wavefronts still spill 97 (0.277 ms; with the direct walk of propagate_path_direct_lds 36, 0.200 against 0.158 ms).
// Direct form of the nested walk (no decay, pairwise distinct shell densities, running product T in
// LDS): the path geometry is unrolled into the loop instead of being asked for segment by segment.
// Walking outwards from the innermost chord, shell k is crossed once on the way in (segment
// l_k - l_{k+1}) and once on the way out (s_{k+1} - s_k), and both lengths are differences of the
// SAME two root terms sqrt(base + r^2) -- the inner one is carried over from the previous step, so a
// step costs one square root where path_segment() through the layer / cache-source callbacks cost
// up to twelve (each 18 instructions).  Lengths, densities, the cache rule (the out-going segment
// takes the in-going one's matrix iff their lengths agree to 1e-5, numba_osc_kernels.py:236-249) and
// the order of the products are those of propagate_path_nested_lds: results are bit-identical.
template <bool LRI, class E>
__device__ __forceinline__ void propagate_path_direct_lds(const Prob3Side &S, const double (&dm)[3][3],
                                                          const int32_t (&vac_order)[3], double energy,
                                                          const E &e, const PathGeom &g, bool ok,
                                                          TLds T, const double *s_uud, double (&P)[9], int32_t *status) {
    bool have = false;
    const double inv_energy = fast_rcp(energy);
    // Every layer matrix A' is in SU(3) (eigen_terms), so the running product T is as well: its third
    // row is the conjugate cross product of the first two.  Only rows 0 and 1 of T are kept ([12][64]
    // doubles of LDS per wavefront).  T.A needs those two rows and all of A (su3_complete); A.T needs
    // rows 0 and 1 of A and all of T (one cross product): 48 complex products per shell pair
    // instead of 60, and two thirds of the LDS traffic.
    auto amplitude = [&](double rho, double dist, mat3 &A) {   // rows 0 and 1
        double rec[PROB3_NF_REDUCED];
        auto store = [&](int f, double v) { rec[f] = v; };
        eigen_terms<false, true, LRI>(S, dm, vac_order, energy, rho, store);
        auto load = [&](int f) { return rec[f]; };
        amplitude_from_terms<false>(load, dist * inv_energy, A);
    };
    auto third_row = [](const cplx (&r0)[3], const cplx (&r1)[3], cplx (&r2)[3]) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int k = (j + 1) % 3, l = (j + 2) % 3;
            const cplx c = csub(cmul(r0[k], r1[l]), cmul(r0[l], r1[k]));
            r2[j] = cmake(c.re, -c.im);
        }
    };
    auto set = [&](const mat3 &A) {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) T.put(i, j, A.m[i][j]);
        have = true;
    };
    auto left = [&](const mat3 &A) {   // T <- A . T  (rows 0, 1 of A)
        if (!have) { set(A); return; }
        cplx t[3][3];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) t[i][j] = T.get(i, j);
        third_row(t[0], t[1], t[2]);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                cplx acc = cmul(A.m[i][0], t[0][j]);
                acc = cadd(acc, cmul(A.m[i][1], t[1][j]));
                acc = cadd(acc, cmul(A.m[i][2], t[2][j]));
                T.put(i, j, acc);
            }
    };
    auto right = [&](mat3 &A) {  // T <- T . A, row by row (completes A)
        su3_complete(A);
        if (!have) { set(A); return; }
#pragma unroll 1
        for (int i = 0; i < 2; i++) {
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
    auto checked = [&](double len) {   // a NaN length (the reference's construction broke down): flag, skip
        if (!(len == len)) { len = 0.0; if (status) atomicOr(status, 1); }
        return len;
    };
    // One loop for both geometries, two amplitude sites.  Up-going (case B of layers.py:94): step t
    // takes shell k = m-1-t, the innermost chord first (in-going "segment" = the whole chord, nothing
    // out-going), then the pairs outwards.  Down-going (case A): the shells above the detector in path
    // order, each a lone out-going segment (layers.py:95-101).  r_c carries the root term of the
    // previous step.
    auto root = [&](int k) { return fast_sqrt(g.base + e.radii[k] * e.radii[k]); };   // root_term()
    const bool tf = g.tangent_free;
    const int m = g.m;
    const int n_steps = ok ? (tf ? e.idx : m) : 0;
    double r_c = tf ? root(0) : 0.0;
    for (int t = 0; t < n_steps; t++) {
        const int k = tf ? t : m - 1 - t;
        const bool last_tf = t == e.idx - 1;
        const double r_n = (tf && last_tf) ? 0.0 : root(tf ? t + 1 : k);
        double d_i = 0.0, d_o = 0.0;
        if (tf) {
            d_o = checked((g.neg_rd_cz + r_c) - (last_tf ? 0.0 : (g.neg_rd_cz + r_n)));
        } else if (t == 0) {
            d_i = checked((g.neg_rd_cz + r_n) - (g.neg_rd_cz - r_n));
        } else {
            d_i = checked((g.neg_rd_cz + r_n) - (g.neg_rd_cz + r_c));
            if (k >= 1) d_o = checked((g.neg_rd_cz - r_c) - (k >= e.idx ? (g.neg_rd_cz - r_n) : 0.0));
        }
        const double rho = e.rhos[k];
        mat3 A;
        const bool in_ok = d_i > 0.0, out_ok = d_o > 0.0;
        if (in_ok) {
            amplitude(rho, d_i, A);
            right(A);
        }
        if (out_ok) {
            const bool same = in_ok && fabs(d_o - d_i) < 1e-5;
            if (!same) amplitude(rho, d_o, A);
            left(A);
        }
        r_c = r_n;
    }
    mat3 Tm, t2, Tf;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Tm.m[i][j] = have ? T.get(i, j) : cmake(0.0, 0.0);
    third_row(Tm.m[0], Tm.m[1], Tm.m[2]);
    // U and U^dagger of the closing flavour-basis transform come back from LDS (staged by the kernel before the walk):
    // as by-value kernel arguments they sat in 72 scalar registers across the whole layer loop, pushed the loop's own
    // constants (X0, XV: 36 more) beyond the scalar register file, and every use inside the loop then began with a
    // v_readlane from a spill lane -- 78 of the loop's 1 287 vector instructions (round 5, ISA count)
    mat3 Um, Udm;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            Um.m[i][j] = cmake(s_uud[2 * (3 * i + j)], s_uud[2 * (3 * i + j) + 1]);
            Udm.m[i][j] = cmake(s_uud[18 + 2 * (3 * i + j)], s_uud[18 + 2 * (3 * i + j) + 1]);
        }
    mat_mul(Tm, Udm, t2);
    mat_mul(Um, t2, Tf);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            P[3 * i + j] = Tf.m[j][i].re * Tf.m[j][i].re + Tf.m[j][i].im * Tf.m[j][i].im;
}

template <bool DECAY, int SIDE, bool STAGED, int WAVES = 2, bool LRI = true>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
prob3_events_kernel(const Prob3Consts c, const EarthDev earth, const EvArgs ev, int max_seg,
                    int32_t *__restrict__ status) {
    // Workgroups are dealt to the containers round-robin (workgroup b = chunk b / n_cont of
    // container b % n_cont).  Every container's events are sorted by coszen, longest paths
    // first, so the long paths of ALL containers run first and the short ones fill the tail
    // (container-major order left each later container's long paths for the end).
    const int ci = (int)(blockIdx.x % (unsigned)ev.n_cont);  // workgroup-uniform
    const int chunk = (int)(blockIdx.x / (unsigned)ev.n_cont);
    const EvCont &C = ev.cont[ci];
    constexpr int side = SIDE;
    const double *__restrict__ energy = C.energy;
    const double *__restrict__ coszen = C.coszen;
    const int64_t n = C.n;
    double *__restrict__ prob = C.prob;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // shell table in LDS (radii, rhos, coszen_limit)
    double *s_radii = reinterpret_cast<double *>(smem);
    double *s_rhos = s_radii + PISA_HIP_MAX_SHELLS;
    double *s_lim = s_rhos + PISA_HIP_MAX_SHELLS;
    double *s_len = s_lim + PISA_HIP_MAX_SHELLS;                       // [max_seg][blockDim] (STAGED)
    unsigned char *s_shell = reinterpret_cast<unsigned char *>(s_len + (size_t)(STAGED ? max_seg : 0) * blockDim.x);
    unsigned char *s_src = s_shell + (size_t)(STAGED ? max_seg : 0) * blockDim.x;
    for (int k = threadIdx.x; k < earth.n_shell; k += blockDim.x) {
        s_radii[k] = earth.radii[k];
        s_rhos[k] = earth.rhos[k];
        s_lim[k] = earth.coszen_limit[k];
    }
    __shared__ double s_uud[36];   // U, U^dagger of this side (re, im interleaved, row major): see propagate_path_direct_lds
    if (WAVES > 2 && !DECAY && !STAGED && threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                s_uud[2 * (3 * i + j)] = c.side[SIDE].U.m[i][j].re;
                s_uud[2 * (3 * i + j) + 1] = c.side[SIDE].U.m[i][j].im;
                s_uud[18 + 2 * (3 * i + j)] = c.side[SIDE].Ud.m[i][j].re;
                s_uud[18 + 2 * (3 * i + j) + 1] = c.side[SIDE].Ud.m[i][j].im;
            }
    }
    __syncthreads();
    struct LdsEarth {
        int32_t n_shell, idx;
        double r_detector;
        const double *radii, *rhos, *coszen_limit;
    } e{earth.n_shell, earth.idx, earth.r_detector, s_radii, s_rhos, s_lim};

    int64_t i = (int64_t)chunk * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int lane = threadIdx.x;
    const int bd = blockDim.x;
    PathGeom g = make_path(e, coszen[i]);
    bool ok = path_valid(e, g);
    int nseg = ok ? g.nseg : 0;
    if (!ok && status) atomicOr(status, 1);
    const int mid = (ok && !g.tangent_free) ? g.m - 1 : -1;
    double P[9];
    const int32_t vac_order[3] = {c.vac_order[0], c.vac_order[1], c.vac_order[2]};
    if (STAGED) {
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
        // the reference's layer-matrix cache, resolved once per path: src[l] = the layer whose
        // matrix layer l uses (numba_osc_kernels.py:236-249: the LAST earlier layer within 1e-5 in
        // density and length, followed through its own matches)
        for (int l = 0; l < nseg; l++) {
            double rho_l, d_l;
            layer(l, rho_l, d_l);
            int sl = l;
            if (d_l > 0.0) {
                int found = -1;
                for (int j = 0; j < l; j++) {
                    double rj, dj;
                    layer(j, rj, dj);
                    if (dj > 0.0 && fabs(rj - rho_l) < 1e-5 && fabs(dj - d_l) < 1e-5) found = j;
                }
                if (found >= 0) sl = s_src[(size_t)found * bd + lane];
            }
            s_src[(size_t)l * bd + lane] = (unsigned char)sl;
        }
        auto src = [&](int l) { return (int)s_src[(size_t)l * bd + lane]; };
        propagate_path_nested<DECAY>(c.side[side], c.dm, vac_order, energy[i], nseg, mid, layer, src, P);
    } else {
        auto layer = [&](int l, double &rho, double &dist) {
            path_segment(e, g, l, rho, dist);
            if (!(dist == dist)) { dist = 0.0; rho = 0.0; if (status) atomicOr(status, 1); }
        };
        // distinct shell densities: only the mirror image (same shell) can match
        auto src = [&](int l) {
            const int mir = 2 * mid - l;
            if (mid < 0 || l <= mid || mir < 0) return l;
            double r1, d1, r2, d2;
            layer(l, r1, d1);
            layer(mir, r2, d2);
            return (d2 > 0.0 && fabs(r2 - r1) < 1e-5 && fabs(d2 - d1) < 1e-5) ? mir : l;
        };
        if (WAVES > 2 && !DECAY) {
            TLds T{s_len + lane, bd};
            propagate_path_direct_lds<LRI>(c.side[side], c.dm, vac_order, energy[i], e, g, ok, T, s_uud, P, status);
        } else if (WAVES > 2 || DECAY) {
            // decay: the reference-order layer matrices (layer_amplitude: complex eigenvalues, three full
            // projectors) need every register there is; the running product waits in LDS meanwhile
            TLds T{s_len + lane, bd};
            propagate_path_nested_lds<DECAY>(c.side[side], c.dm, vac_order, energy[i], nseg, mid, layer, src, T, P);
        } else {
            propagate_path_nested<DECAY>(c.side[side], c.dm, vac_order, energy[i], nseg, mid, layer, src, P);
        }
    }
    if (prob) {
#pragma unroll
        for (int k = 0; k < 9; k++) prob[9 * i + k] = P[k];
    }
    if (C.pepmu) C.pepmu[i] = make_double2(P[C.flav], P[3 + C.flav]);
}

}  // namespace pisa

using namespace pisa;

// wavefronts per SIMD of the decay instantiation: two (248 VGPRs, no scratch) -- three (168 VGPRs, 73 spilled
// dwords) ran 0.496 against 0.408 ms per 1e6 events (EXPERIMENTS R4-7)
#ifndef DECAY_WAVES
#define DECAY_WAVES 2
#endif
static int launch_events(const Prob3Consts &c, const EarthDev &e, const EvCont *conts, int n_cont,
                         int32_t *d_status, hipStream_t s) {
    int max_seg = 2 * e.n_shell;
    if (max_seg > PISA_HIP_MAX_LAYERS + 8) return PISA_HIP_ERR_LAYERS;
    const int threads = 64;
    // The direct form needs pairwise distinct densities among the shells a path can cross at all
    // (coszen_limit > -1: make_path counts the shells with coszen_limit > coszen; the r = 0 entry of
    // the PREM tables repeats the inner core's density)
    bool staged = false;
    for (int a = 0; a < e.n_shell; a++)
        for (int b = a + 1; b < e.n_shell; b++)
            if (e.coszen_limit[a] > -1.0 && e.coszen_limit[b] > -1.0 && fabs(e.rhos[a] - e.rhos[b]) < 1e-5)
                staged = true;
    static const int force_staged = PISA_DEV_INT("EVENTS_STAGED", 0);
    if (force_staged) staged = true;   // development / test switch: the general form
    // direct form: 3 wavefronts per SIMD with the running product in LDS (see the kernel); 2: product in registers
    static const int waves_cfg = PISA_DEV_INT("EVENTS_WAVES", 3);
    // a long-range potential at all?  (XL = U^dagger . lri . U of either side: all zeros without one)
    bool has_lri = false;
    for (int sd = 0; sd < 2; sd++)
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                if (c.side[sd].XL.m[i][j].re != 0.0 || c.side[sd].XL.m[i][j].im != 0.0) has_lri = true;
    size_t lds = 3 * PISA_HIP_MAX_SHELLS * sizeof(double) +
                 (staged ? (size_t)max_seg * threads * 10
                         : (c.decay ? (size_t)18 * threads * 8 : (waves_cfg > 2 ? (size_t)12 * threads * 8 : 0))) + 16;
    // one launch per sign (see the kernel) and per EV_MAX_CONT containers
    for (int side = 0; side < 2; side++) {
        EvArgs a;
        a.n_cont = 0;
        a.blk_start[0] = 0;
        auto flush = [&]() -> int {
            if (a.n_cont == 0 || a.blk_start[a.n_cont] == 0) { a.n_cont = 0; return PISA_HIP_OK; }
            int max_blocks = 0;
            for (int k = 0; k < a.n_cont; k++) {
                const int nb = a.blk_start[k + 1] - a.blk_start[k];
                max_blocks = nb > max_blocks ? nb : max_blocks;
            }
            dim3 block(threads), grid((unsigned)max_blocks * (unsigned)a.n_cont);
#define LAUNCH_EV(D, S_, ST) hipLaunchKernelGGL((prob3_events_kernel<D, S_, ST>), grid, block, lds, s, c, e, a, max_seg, d_status)
#define LAUNCH_SIDE(D, ST) do { if (side == 0) LAUNCH_EV(D, 0, ST); else LAUNCH_EV(D, 1, ST); } while (0)
            if (c.decay) {
                if (staged) LAUNCH_SIDE(true, true);
                else if (side == 0) hipLaunchKernelGGL((prob3_events_kernel<true, 0, false, DECAY_WAVES>), grid, block, lds, s, c, e, a, max_seg, d_status);
                else hipLaunchKernelGGL((prob3_events_kernel<true, 1, false, DECAY_WAVES>), grid, block, lds, s, c, e, a, max_seg, d_status);
            }
            else if (staged) LAUNCH_SIDE(false, true);
            else if (waves_cfg == 4 && !has_lri) { if (side == 0) hipLaunchKernelGGL((prob3_events_kernel<false, 0, false, 4, false>), grid, block, lds, s, c, e, a, max_seg, d_status); else hipLaunchKernelGGL((prob3_events_kernel<false, 1, false, 4, false>), grid, block, lds, s, c, e, a, max_seg, d_status); }
            else if (waves_cfg >= 3 && !has_lri) { if (side == 0) hipLaunchKernelGGL((prob3_events_kernel<false, 0, false, 3, false>), grid, block, lds, s, c, e, a, max_seg, d_status); else hipLaunchKernelGGL((prob3_events_kernel<false, 1, false, 3, false>), grid, block, lds, s, c, e, a, max_seg, d_status); }
            else if (waves_cfg >= 3) { if (side == 0) hipLaunchKernelGGL((prob3_events_kernel<false, 0, false, 3>), grid, block, lds, s, c, e, a, max_seg, d_status); else hipLaunchKernelGGL((prob3_events_kernel<false, 1, false, 3>), grid, block, lds, s, c, e, a, max_seg, d_status); }
            else LAUNCH_SIDE(false, false);
#undef LAUNCH_SIDE
#undef LAUNCH_EV
            PISA_CHECK_LAUNCH("prob3_events_kernel");
            a.n_cont = 0;
            return PISA_HIP_OK;
        };
        for (int k = 0; k < n_cont; k++) {
            if (conts[k].side != side) continue;
            a.cont[a.n_cont] = conts[k];
            a.blk_start[a.n_cont + 1] = a.blk_start[a.n_cont] + (int)((conts[k].n + threads - 1) / threads);
            a.n_cont++;
            if (a.n_cont == EV_MAX_CONT) {
                int rc = flush();
                if (rc) return rc;
            }
        }
        int rc = flush();
        if (rc) return rc;
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

