// evaluator.hip -- one template evaluation per C-ABI call (include/pisa_hip.h, "one template evaluation
// in ONE call"): the launches of pisa_hip_prob3_grid_planned, pisa_hip_reweight_hist[_acc] and
// pisa_hip_finalize_metric_split enqueued from here, with everything that does not change between parameter
// points held by the evaluator.  A fit loop's critical path between the metric of point k arriving in pinned
// memory and the first kernel of point k+1 is then one FFI crossing and `make_consts`.
#include <math.h>
#include <string.h>

#include <chrono>
#include <thread>
#include <vector>

#include "common.hpp"

struct pisa_hip_evaluator {
    std::vector<pisa_hip_container> cont;
    pisa_hip_binning calc_grid, out_binning;
    pisa_hip_evaluator_desc d;
    int64_t n_bins, limb_count;
};

using namespace pisa;

PISA_API int pisa_hip_evaluator_create(const pisa_hip_evaluator_desc *desc, pisa_hip_evaluator **out) {
    if (!desc || !out || !desc->h_containers || desc->n_containers < 1 || desc->n_containers > 1024 ||
        !desc->h_calc_grid || !desc->h_out_binning || !desc->plan || !desc->d_energy || desc->n_e < 1 ||
        !desc->d_pepmu || !desc->d_limbs || !desc->d_hist || !desc->d_sumw2 || !desc->partial ||
        !desc->d_status || !desc->d_metric_status || ((desc->allreduce == nullptr) != (desc->comm == nullptr)))
        return PISA_HIP_ERR_INVALID;
    DevBinning b;
    int64_t n_bins = 0, n_nodes = 0;
    int rc = make_dev_binning(desc->h_out_binning, b, n_bins);
    if (rc) return rc;
    if ((rc = make_dev_binning(desc->h_calc_grid, b, n_nodes))) return rc;
    if ((int64_t)desc->n_containers * n_bins > PISA_HIP_FINALIZE_METRIC_MAX) return PISA_HIP_ERR_INVALID;
    pisa_hip_evaluator *ev = new pisa_hip_evaluator();
    ev->cont.assign(desc->h_containers, desc->h_containers + desc->n_containers);
    ev->calc_grid = *desc->h_calc_grid;
    ev->out_binning = *desc->h_out_binning;
    ev->d = *desc;
    ev->d.h_containers = ev->cont.data();
    ev->d.h_calc_grid = &ev->calc_grid;
    ev->d.h_out_binning = &ev->out_binning;
    ev->n_bins = n_bins;
    ev->limb_count = (int64_t)desc->n_containers * n_bins * 2 * PISA_HIP_ACC_LIMBS;
    *out = ev;
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_evaluator_destroy(pisa_hip_evaluator *ev) {
    delete ev;
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_evaluator_set_scale(pisa_hip_evaluator *ev, int32_t container, double scale) {
    if (!ev || container < 0 || container >= (int32_t)ev->cont.size()) return PISA_HIP_ERR_INVALID;
    ev->cont[container].scale = scale;
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_evaluator_eval(pisa_hip_evaluator *ev, const pisa_hip_prob3_params *h_params, int32_t kind,
                                     const double *d_actual, int32_t limbs_zero, int64_t wait_us, double *value,
                                     void *stream) {
    if (!ev || !h_params || !d_actual || (wait_us > 0 && !value)) return PISA_HIP_ERR_INVALID;
    const pisa_hip_evaluator_desc &d = ev->d;
    const bool split = kind != PISA_HIP_METRIC_CHI2;
    volatile double *part = d.partial;
    constexpr int PARTS = 16;
    const int n_part = split ? PARTS : 1;
    if (wait_us > 0)
        for (int k = 0; k < n_part; k++) part[k] = NAN;   // "not yet": a partial sum is never NaN unless an input was negative
    int rc = pisa_hip_prob3_grid_planned(h_params, d.plan, d.d_energy, d.n_e, d.e_major, nullptr, nullptr, d.d_pepmu,
                                         stream);
    if (rc) return rc;
    rc = (limbs_zero ? pisa_hip_reweight_hist_acc : pisa_hip_reweight_hist)(
        ev->cont.data(), (int32_t)ev->cont.size(), &ev->calc_grid, nullptr, nullptr, d.d_pepmu, &ev->out_binning,
        d.d_limbs, d.d_status, stream);
    if (rc) return rc;
    if (d.allreduce) {
        const int nrc = d.allreduce(d.d_limbs, d.d_limbs, (size_t)ev->limb_count, 4 /* ncclInt64 */, 0 /* ncclSum */,
                                    d.comm, stream);
        if (nrc != 0) {
            set_last_hip_error(hipErrorUnknown, "all-reduce of the limbs (the binding's allreduce entry) failed");
            return PISA_HIP_ERR_HIP;
        }
    }
    if (split)
        rc = pisa_hip_finalize_metric_parts(d.d_limbs, 1, (int32_t)ev->cont.size(), ev->n_bins, d.d_hist, d.d_sumw2, kind,
                                            d_actual, nullptr, 0, nullptr, d.partial, PARTS, d.d_status,
                                            d.d_metric_status, 1, stream);
    else
        rc = pisa_hip_finalize_metric((int64_t *)d.d_limbs, (int32_t)ev->cont.size(), ev->n_bins, d.d_hist, d.d_sumw2, kind,
                                      d_actual, d.partial, d.d_status, d.d_metric_status, 1, stream);
    if (rc || wait_us <= 0) return rc;
    // The tail kernel's stores into pinned host memory are visible a few microseconds before the stream's
    // completion signal has travelled through the runtime: poll them.
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(wait_us);
    bool have = false;
    double p[PARTS] = {0};
    for (unsigned it = 0;; it++) {
        have = true;
        for (int k = 0; k < n_part; k++) {
            p[k] = part[k];
            if (p[k] != p[k]) have = false;
        }
        if (have) break;
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
        if ((it & 255u) == 255u && std::chrono::steady_clock::now() > t_end) break;
    }
    if (!have) {   // slow evaluation or a genuine NaN (negative input: the status word says so)
        PISA_TRY_HIP(hipStreamSynchronize(as_stream(stream)));
        for (int k = 0; k < n_part; k++) p[k] = part[k];
    }
    if (split)      // the kernel's reduction tree, its last four levels
        for (int w = PARTS / 2; w >= 1; w >>= 1)
            for (int i = 0; i < w; i++) p[i] += p[i + w];
    *value = p[0];
    return PISA_HIP_OK;
}
