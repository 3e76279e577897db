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
#ifdef PISA_DEV_PROBES
    // round-6 experiment (EXPERIMENTS R6-3): the accumulate kernel on a second stream, resident and polling while the chain
    // kernel runs (PISA_HIP_EVAL_OVERLAP=1)
    hipStream_t side = nullptr;
    hipEvent_t ev_before = nullptr, ev_after = nullptr;
    pisa::HandOver hand = {nullptr, 0, 0};
    bool overlap_ok = false;
#endif
};

using namespace pisa;

#ifdef PISA_DEV_PROBES
namespace pisa {
thread_local HandOver g_chain_signal = {nullptr, 0, 0}, g_hist_wait = {nullptr, 0, 0};
}
#endif

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
#ifdef PISA_DEV_PROBES
    if (PISA_DEV_INT("EVAL_OVERLAP", 0) && !desc->allreduce && desc->n_containers <= 16) {
        // only where the accumulate launch leaves every CU room for the chain kernel's workgroups: one workgroup per CU
        std::vector<int64_t> nev(desc->n_containers);
        std::vector<int32_t> wgs(desc->n_containers);
        for (int c = 0; c < desc->n_containers; c++) nev[c] = desc->h_containers[c].n_events;
        int total = 0, cus = 0, dev = 0;
        if (pisa_hip_hist_workgroups(nev.data(), desc->n_containers, wgs.data()) == PISA_HIP_OK &&
            hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) {
            for (int w : wgs) total += w;
            if (total <= cus && hipStreamCreateWithFlags(&ev->side, hipStreamNonBlocking) == hipSuccess &&
                hipEventCreateWithFlags(&ev->ev_before, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&ev->ev_after, hipEventDisableTiming) == hipSuccess &&
                hipMalloc((void **)&ev->hand.flags, HANDOVER_SLOTS * sizeof(unsigned long long)) == hipSuccess &&
                hipMemset(ev->hand.flags, 0, HANDOVER_SLOTS * sizeof(unsigned long long)) == hipSuccess)
                ev->overlap_ok = true;
        }
    }
#endif
    *out = ev;
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_evaluator_destroy(pisa_hip_evaluator *ev) {
#ifdef PISA_DEV_PROBES
    if (ev && ev->side) {
        (void)hipStreamSynchronize(ev->side);
        (void)hipStreamDestroy(ev->side);
        if (ev->ev_before) (void)hipEventDestroy(ev->ev_before);
        if (ev->ev_after) (void)hipEventDestroy(ev->ev_after);
        if (ev->hand.flags) (void)hipFree(ev->hand.flags);
    }
#endif
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
    int rc;
#ifdef PISA_DEV_PROBES
    if (ev->overlap_ok) {
        // the accumulate launch goes to the side stream behind everything queued on `stream` so far, the oscillation
        // launches to `stream`; the tail joins them again
        hipStream_t s = as_stream(stream);
        PISA_TRY_HIP(hipEventRecord(ev->ev_before, s));
        PISA_TRY_HIP(hipStreamWaitEvent(ev->side, ev->ev_before, 0));
        const int order = PISA_DEV_INT("EVAL_OVERLAP", 0);    // 1: oscillation launches first, 2: accumulate launch first
        g_chain_signal = ev->hand;
        g_chain_signal.epoch = ev->hand.epoch;
        auto osc = [&]() {
            int r = pisa_hip_prob3_grid_planned(h_params, d.plan, d.d_energy, d.n_e, d.e_major, nullptr, nullptr, d.d_pepmu, stream);
            return r;
        };
        auto acc = [&]() {
            return (limbs_zero ? pisa_hip_reweight_hist_acc : pisa_hip_reweight_hist)(
                ev->cont.data(), (int32_t)ev->cont.size(), &ev->calc_grid, nullptr, nullptr, d.d_pepmu, &ev->out_binning,
                d.d_limbs, d.d_status, (void *)ev->side);
        };
        if (order == 2 && ev->hand.n_wg > 0) {
            // (the workgroup count of the chain launch is known from the previous evaluation)
            g_hist_wait = {ev->hand.flags, ev->hand.epoch + 1, ev->hand.n_wg};
            rc = acc();
            g_hist_wait = {nullptr, 0, 0};
            if (rc) { g_chain_signal = {nullptr, 0, 0}; return rc; }
            rc = osc();
            const bool signalled = g_chain_signal.epoch == ev->hand.epoch + 1;
            ev->hand.epoch = g_chain_signal.epoch;
            ev->hand.n_wg = g_chain_signal.n_wg;
            g_chain_signal = {nullptr, 0, 0};
            if (rc) return rc;
            if (!signalled) { set_last_hip_error(hipErrorUnknown, "overlap: the chain launch did not signal"); return PISA_HIP_ERR_HIP; }
        } else {
            rc = osc();
            const bool signalled = g_chain_signal.epoch == ev->hand.epoch + 1;
            ev->hand.epoch = g_chain_signal.epoch;
            ev->hand.n_wg = g_chain_signal.n_wg;
            g_chain_signal = {nullptr, 0, 0};
            if (rc) return rc;
            if (signalled) g_hist_wait = {ev->hand.flags, ev->hand.epoch, ev->hand.n_wg};
            else PISA_TRY_HIP(hipStreamWaitEvent(ev->side, ev->ev_before, 0));   // (not the packed one-point form: no hand-over)
            if (!signalled) {
                // no signal: order the accumulate launch behind the oscillation launches the ordinary way
                PISA_TRY_HIP(hipEventRecord(ev->ev_before, s));
                PISA_TRY_HIP(hipStreamWaitEvent(ev->side, ev->ev_before, 0));
            }
            rc = acc();
            g_hist_wait = {nullptr, 0, 0};
            if (rc) return rc;
        }
        PISA_TRY_HIP(hipEventRecord(ev->ev_after, ev->side));
        PISA_TRY_HIP(hipStreamWaitEvent(s, ev->ev_after, 0));
    } else
#endif
    {
    rc = pisa_hip_prob3_grid_planned(h_params, d.plan, d.d_energy, d.n_e, d.e_major, nullptr, nullptr, d.d_pepmu,
                                         stream);
    if (rc) return rc;
    rc = (limbs_zero ? pisa_hip_reweight_hist_acc : pisa_hip_reweight_hist)(
        ev->cont.data(), (int32_t)ev->cont.size(), &ev->calc_grid, nullptr, nullptr, d.d_pepmu, &ev->out_binning,
        d.d_limbs, d.d_status, stream);
    if (rc) return rc;
    }
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
