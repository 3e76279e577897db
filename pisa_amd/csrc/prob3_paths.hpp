// prob3_paths.hpp -- Earth shell table and lazily evaluated path geometry (layers.py:38-169),
// shared by calc_layers_kernel (prob3.hip, reference operation order, no contraction) and the
// event-mode kernel (prob3_events.hip).
#pragma once
#include "common.hpp"
#include "prob3_device.hpp"

namespace pisa {

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


inline int make_earth_dev(const pisa_hip_earth *h, EarthDev &e) {
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

inline int make_consts(const pisa_hip_prob3_params *p, Prob3Consts &c) {
    if (!p) return PISA_HIP_ERR_INVALID;
    prob3_make_consts(p->dm, p->mix, p->mat_pot, p->mat_decay, p->lri_pot, p->decay_flag, c);
    return PISA_HIP_OK;
}

}  // namespace pisa
