// kde_batch.hip -- the estimators of one KDE-stage evaluation as ONE call
//
// One evaluation of the KDE stage (pisa/stages/utils/kde.py:154-293) builds an independent estimator per
// container and pid channel (kde_hist.py:303-372: 24 at the C3 size) and evaluates each on the same
// lattice of oversampled bin centres.  Every estimator is a chain of ~50 launches with a handful of host
// round trips (moments, cell heads, bandwidth range), so the chains are run side by side: a pool of host
// threads inside the library, each with its own stream and its own grow-only workspace, takes the jobs
// largest first.  The results are those of pisa_hip_kde_create + pisa_hip_kde_evaluate_lattice job by job
// (each estimator is deterministic by itself); what the pool removes is the interpreter between the calls.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "common.hpp"

namespace pisa {

// w[k] = weights[index[k]] (index may be null), NaN -> 0, +-inf -> +-DBL_MAX  (torch.nan_to_num,
// kde_hist.py:104-108 `weights = np.nan_to_num(weights)` semantics of the reference's wrapper)
__global__ void __launch_bounds__(256)
kde_job_weights_kernel(const double *__restrict__ weights, const int64_t *__restrict__ index, int64_t n,
                       double *__restrict__ out) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    double v = weights[index ? index[k] : k];
    if (v != v) v = 0.0;
    else if (v > 1.7976931348623157e308) v = 1.7976931348623157e308;
    else if (v < -1.7976931348623157e308) v = -1.7976931348623157e308;
    out[k] = v;
}

struct KdeWorkerState {   // per pool thread
    int device = -1;
    hipStream_t stream = nullptr;
    void *work = nullptr, *lwork = nullptr;
    size_t work_bytes = 0, lwork_bytes = 0;
};

static int grow(void **buf, size_t *have, size_t need, hipStream_t s) {
    if (need <= *have) return PISA_HIP_OK;
    if (*buf) {
        PISA_TRY_HIP(hipStreamSynchronize(s));
        (void)hipFree(*buf);
        *buf = nullptr;
        *have = 0;
    }
    const size_t bytes = need + need / 4;
    PISA_TRY_HIP(hipMalloc(buf, bytes));
    *have = bytes;
    return PISA_HIP_OK;
}

class KdePool {
  public:
    ~KdePool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_job_.notify_all();
        for (auto &t : threads_) t.join();
    }
    // runs fn(i, state) for i in order[], on up to n_threads pool threads; returns when all are done
    void run(const std::vector<int> &order, int n_threads, const std::function<void(int, KdeWorkerState &)> &fn) {
        std::unique_lock<std::mutex> lk(m_);
        while ((int)threads_.size() < n_threads) threads_.emplace_back([this] { loop(); });
        // one batch at a time (a second caller waits here)
        cv_done_.wait(lk, [this] { return pending_ == 0 && queue_.empty(); });
        fn_ = &fn;
        limit_ = n_threads;
        for (int i : order) queue_.push_back(i);
        pending_ = (int)order.size();
        cv_job_.notify_all();
        cv_done_.wait(lk, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }

  private:
    void loop() {
        KdeWorkerState st;
        int my_id;
        {
            std::lock_guard<std::mutex> lk(m_);
            my_id = n_started_++;
        }
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_job_.wait(lk, [&] { return stop_ || (!queue_.empty() && my_id < limit_); });
            if (stop_) return;
            const int i = queue_.front();
            queue_.pop_front();
            const auto *fn = fn_;
            lk.unlock();
            (*fn)(i, st);
            lk.lock();
            if (--pending_ == 0) cv_done_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_job_, cv_done_;
    std::vector<std::thread> threads_;
    std::deque<int> queue_;
    const std::function<void(int, KdeWorkerState &)> *fn_ = nullptr;
    int pending_ = 0, limit_ = 0, n_started_ = 0;
    bool stop_ = false;
};

static KdePool &pool() {
    static KdePool *p = new KdePool();   // never destroyed: its threads only ever wait on the queue at exit
    return *p;
}

}  // namespace pisa

using namespace pisa;

PISA_API int pisa_hip_kde_lattice_batch(pisa_hip_kde_job *jobs, int32_t n_jobs, int32_t dim, int32_t bw_method,
                                        int32_t adaptive, double alpha, double tol, const double *h_origin,
                                        const double *h_step, const int64_t *h_count, int32_t n_threads,
                                        void *stream) {
    if (n_jobs < 0 || (n_jobs > 0 && !jobs) || !h_origin || !h_step || !h_count || dim < 1 || dim > 3)
        return PISA_HIP_ERR_INVALID;
    if (n_jobs == 0) return PISA_HIP_OK;
    for (int i = 0; i < n_jobs; i++) {
        if (!jobs[i].d_x || !jobs[i].d_out || jobs[i].n < 2 || (jobs[i].d_index && !jobs[i].d_weights))
            return PISA_HIP_ERR_INVALID;
        jobs[i].status = PISA_HIP_ERR_INVALID;
        jobs[i].sum_w = 0.0;
    }
    int device = 0;
    PISA_TRY_HIP(hipGetDevice(&device));
    // the inputs were produced on the caller's stream
    hipEvent_t ready;
    PISA_TRY_HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    int rc0 = check_hip(hipEventRecord(ready, as_stream(stream)), "hipEventRecord");
    if (rc0 != PISA_HIP_OK) { (void)hipEventDestroy(ready); return rc0; }
    std::vector<int> order(n_jobs);
    for (int i = 0; i < n_jobs; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return jobs[a].n > jobs[b].n; });   // longest first
    const int nt = std::max(1, std::min<int>(n_threads > 0 ? n_threads : 8, n_jobs));
    auto run = [&](int i, KdeWorkerState &st) {
        pisa_hip_kde_job &job = jobs[i];
        auto body = [&]() -> int {
            if (st.device != device) {
                PISA_TRY_HIP(hipSetDevice(device));
                if (st.stream) { (void)hipStreamDestroy(st.stream); st.stream = nullptr; }
                if (st.work) { (void)hipFree(st.work); st.work = nullptr; st.work_bytes = 0; }
                if (st.lwork) { (void)hipFree(st.lwork); st.lwork = nullptr; st.lwork_bytes = 0; }
                PISA_TRY_HIP(hipStreamCreateWithFlags(&st.stream, hipStreamNonBlocking));
                st.device = device;
            }
            hipStream_t s = st.stream;
            PISA_TRY_HIP(hipStreamWaitEvent(s, ready, 0));
            const int64_t need = pisa_hip_kde_workspace_bytes(dim, job.n);
            if (need < 0) return PISA_HIP_ERR_INVALID;
            const size_t wbytes = (((size_t)job.n * 8) + 255) & ~(size_t)255;
            int rc = grow(&st.work, &st.work_bytes, (size_t)need + wbytes, s);
            if (rc != PISA_HIP_OK) return rc;
            const double *d_w = nullptr;
            if (job.d_weights) {
                double *w = (double *)st.work;
                hipLaunchKernelGGL(kde_job_weights_kernel, dim3((unsigned)((job.n + 255) / 256)), dim3(256), 0, s,
                                   job.d_weights, job.d_index, job.n, w);
                PISA_CHECK_LAUNCH("kde_job_weights_kernel");
                d_w = w;
            }
            pisa_hip_kde *k = nullptr;
            rc = pisa_hip_kde_create(dim, job.d_x, d_w, job.n, bw_method, adaptive, alpha, tol, (char *)st.work + wbytes,
                                     need, &k, s);
            if (rc != PISA_HIP_OK) return rc;
            const int64_t lneed = pisa_hip_kde_lattice_workspace_bytes(k, h_step, h_count);
            rc = lneed < 0 ? PISA_HIP_ERR_INVALID : grow(&st.lwork, &st.lwork_bytes, (size_t)lneed, s);
            if (rc == PISA_HIP_OK)
                rc = pisa_hip_kde_evaluate_lattice(k, h_origin, h_step, h_count, st.lwork, lneed, job.d_out, s);
            pisa_hip_kde_info_t info;
            if (rc == PISA_HIP_OK) rc = pisa_hip_kde_info(k, &info);
            if (rc == PISA_HIP_OK) {
                job.sum_w = info.sum_w;
                job.pairs_pilot = info.pairs_pilot;
                job.pairs_eval = info.pairs_eval;
            }
            (void)pisa_hip_kde_destroy(k);
            if (rc != PISA_HIP_OK) return rc;
            PISA_TRY_HIP(hipStreamSynchronize(s));
            return PISA_HIP_OK;
        };
        job.status = body();
    };
    pool().run(order, nt, run);
    (void)hipEventDestroy(ready);
    for (int i = 0; i < n_jobs; i++)
        if (jobs[i].status != PISA_HIP_OK) return jobs[i].status;
    return PISA_HIP_OK;
}
