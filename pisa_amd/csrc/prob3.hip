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

// ------------------------------------------------------------------- layers
struct EarthDev {
    int32_t n_shell;
    int32_t idx;  // first shell with radius < r_detector (layers.py:90)
    double r_detector;
    double radii[PISA_HIP_MAX_SHELLS];
    double rhos[PISA_HIP_MAX_SHELLS];
    double coszen_limit[PISA_HIP_MAX_SHELLS];
};

// Geometry of one path (layers.py:86-159), evaluated lazily:
// segment(i) returns the i-th (rho, length) in path order, production -> detector.
struct PathGeom {
    double coszen, neg_rd_cz, base;  // base = rd^2 cz^2 - rd^2
    int m;                            // shells crossed (coszen_limit > coszen)
    int nseg;
    bool tangent_free;                // case A of layers.py:94
};

template <class E>
__device__ __forceinline__ double root_term(const E &e, const PathGeom &g, int k) {
    return sqrt(g.base + e.radii[k] * e.radii[k]);
}

template <class E>
__device__ __forceinline__ PathGeom make_path(const E &e, double coszen) {
    PathGeom g;
    g.coszen = coszen;
    double rd = e.r_detector;
    double rd2 = rd * rd;
    g.neg_rd_cz = -rd * coszen;
    g.base = rd2 * (coszen * coszen) - rd2;
    g.tangent_free = coszen >= e.coszen_limit[e.idx];
    int m = 0;
    for (int k = 0; k < e.n_shell; k++) m += (e.coszen_limit[k] > coszen) ? 1 : 0;
    g.m = m;
    g.nseg = g.tangent_free ? e.idx : (2 * m - 2);
    return g;
}

// returns false if the reference's own construction breaks down for this path
template <class E>
__device__ __forceinline__ bool path_valid(const E &e, const PathGeom &g) {
    if (g.tangent_free) return true;
    // densities list has 2m-2 entries, segments 2m-idx (layers.py:148-158)
    return e.idx == 2 && g.m >= 3;
}

template <class E>
__device__ __forceinline__ void path_segment(const E &e, const PathGeom &g, int i, double &rho,
                                             double &len) {
    if (g.tangent_free) {
        // cumulative distance to shell k's outer radius, k < idx (layers.py:95-101)
        double ck = g.neg_rd_cz + root_term(e, g, i);
        double prev = (i == e.idx - 1) ? 0.0 : (g.neg_rd_cz + root_term(e, g, i + 1));
        len = ck - prev;
        rho = e.rhos[i] * (len > 0. ? 1.0 : 0.0);
        return;
    }
    const int m = g.m;
    int shell;
    if (i < m - 1) {  // far side, going in: l_i - l_{i+1}
        len = (g.neg_rd_cz + root_term(e, g, i)) - (g.neg_rd_cz + root_term(e, g, i + 1));
        shell = i;
    } else if (i == m - 1) {  // innermost chord: l_{m-1} - s_{m-1}
        double t = root_term(e, g, m - 1);
        len = (g.neg_rd_cz + t) - (g.neg_rd_cz - t);
        shell = m - 1;
    } else {  // near side, coming out: s_{sh+1} - s_sh  (s_1 := 0 at the detector)
        shell = 2 * m - 2 - i;
        double hi = g.neg_rd_cz - root_term(e, g, shell + 1);
        double lo = (shell >= e.idx) ? (g.neg_rd_cz - root_term(e, g, shell)) : 0.0;
        len = hi - lo;
    }
    rho = e.rhos[shell] * (len > 0. ? 1.0 : 0.0);
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
template <bool DECAY, int SIDE>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2)))
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
    double *s_len = s_lim + PISA_HIP_MAX_SHELLS;                       // [max_seg][blockDim]
    unsigned char *s_shell = reinterpret_cast<unsigned char *>(s_len + (size_t)max_seg * blockDim.x);
    unsigned char *s_src = s_shell + (size_t)max_seg * blockDim.x;
    for (int k = threadIdx.x; k < earth.n_shell; k += blockDim.x) {
        s_radii[k] = earth.radii[k];
        s_rhos[k] = earth.rhos[k];
        s_lim[k] = earth.coszen_limit[k];
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
    double P[9];
    // through-going paths: in 0..m-2, innermost m-1, out m..2m-3 (path_segment)
    const int32_t vac_order[3] = {c.vac_order[0], c.vac_order[1], c.vac_order[2]};
    propagate_path_nested<DECAY>(c.side[side], c.dm, vac_order, energy[i], nseg,
                                 (ok && !g.tangent_free) ? g.m - 1 : -1, layer, src, P);
    if (prob) {
#pragma unroll
        for (int k = 0; k < 9; k++) prob[9 * i + k] = P[k];
    }
    if (C.pepmu) C.pepmu[i] = make_double2(P[C.flav], P[3 + C.flav]);
}

__global__ void fill_probs_kernel(const double *__restrict__ prob, int init_flav, int flav,
                                  int64_t n, double *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = prob[9 * i + 3 * init_flav + flav];
}

static int make_earth_dev(const pisa_hip_earth *h, EarthDev &e) {
    if (!h || h->n_shell < 2 || h->n_shell > PISA_HIP_MAX_SHELLS) return PISA_HIP_ERR_INVALID;
    e.n_shell = h->n_shell;
    e.r_detector = h->r_detector;
    e.idx = -1;
    for (int k = 0; k < PISA_HIP_MAX_SHELLS; k++) {
        e.radii[k] = k < h->n_shell ? h->radii[k] : 0.0;
        e.rhos[k] = k < h->n_shell ? h->rhos[k] : 0.0;
        e.coszen_limit[k] = k < h->n_shell ? h->coszen_limit[k] : -2.0;
    }
    for (int k = 0; k < h->n_shell; k++)
        if (h->radii[k] < h->r_detector) { e.idx = k; break; }
    if (e.idx < 1) return PISA_HIP_ERR_GEOMETRY;
    return PISA_HIP_OK;
}

static int make_consts(const pisa_hip_prob3_params *p, Prob3Consts &c) {
    if (!p) return PISA_HIP_ERR_INVALID;
    prob3_make_consts(p->dm, p->mix, p->mat_pot, p->mat_decay, p->lri_pot, p->decay_flag, c);
    return PISA_HIP_OK;
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

static int launch_events(const Prob3Consts &c, const EarthDev &e, const EvCont *conts, int n_cont,
                         int32_t *d_status, hipStream_t s) {
    int max_seg = 2 * e.n_shell;
    if (max_seg > PISA_HIP_MAX_LAYERS + 8) return PISA_HIP_ERR_LAYERS;
    const int threads = 64;
    size_t lds = 3 * PISA_HIP_MAX_SHELLS * sizeof(double) + (size_t)max_seg * threads * 10 + 16;
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
#define LAUNCH_EV(D, S_) hipLaunchKernelGGL((prob3_events_kernel<D, S_>), grid, block, lds, s, c, e, a, max_seg, d_status)
            if (c.decay) { if (side == 0) LAUNCH_EV(true, 0); else LAUNCH_EV(true, 1); }
            else { if (side == 0) LAUNCH_EV(false, 0); else LAUNCH_EV(false, 1); }
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

