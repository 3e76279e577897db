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
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
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

static void free_state(KdeWorkerState &st) {
    if (st.device >= 0) (void)hipSetDevice(st.device);
    if (st.stream) (void)hipStreamSynchronize(st.stream);
    (void)pisa_hip_kde_release_scratch();      // the estimator's per-thread scratch (kde.hip)
    if (st.work) (void)hipFree(st.work);
    if (st.lwork) (void)hipFree(st.lwork);
    if (st.stream) (void)hipStreamDestroy(st.stream);
    st = KdeWorkerState();
}

class KdePool {
  public:
    typedef std::function<void(KdeWorkerState &)> Task;
    ~KdePool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_job_.notify_all();
        for (auto &t : threads_) t.join();
    }
    // queues the tasks (in the given order) for up to n_threads pool threads and returns
    // (`tags`: what wait_for() is asked about -- the job a task works on)
    void submit(std::vector<Task> &&tasks, const std::vector<const void *> &tags, int n_threads) {
        std::lock_guard<std::mutex> lk(m_);
        // a new thread starts from the release epoch of its creation (read here, under the lock): a release()
        // that arrives before the thread has run a single instruction is still seen and answered by it
        while ((int)threads_.size() < n_threads) {
            const int epoch = release_epoch_;
            threads_.emplace_back([this, epoch] { loop(epoch); });
        }
        limit_ = std::max(limit_, n_threads);
        pending_ += (int)tasks.size();
        for (size_t i = 0; i < tasks.size(); i++) {
            ++open_[tags[i]];      // (counted: the same job struct may be queued again while its first run is still open)
            queue_.emplace_back(std::move(tasks[i]), tags[i]);
        }
        cv_job_.notify_all();
    }
    // returns when none of the n tagged tasks is queued or running any more
    void wait_for(const void *const *tags, int n) {
        std::unique_lock<std::mutex> lk(m_);
        cv_done_.wait(lk, [&] {
            for (int i = 0; i < n; i++)
                if (open_.count(tags[i])) return false;
            return true;
        });
    }
    // returns when everything submitted so far is done
    void wait() {
        std::unique_lock<std::mutex> lk(m_);
        cv_done_.wait(lk, [this] { return pending_ == 0; });
        limit_ = 0;
    }
    // every pool thread frees its stream, workspaces and library scratch (they are grow-only otherwise); returns when
    // all of them have
    void release() {
        std::unique_lock<std::mutex> lk(m_);
        cv_done_.wait(lk, [this] { return pending_ == 0; });
        release_epoch_++;
        released_ = 0;
        cv_job_.notify_all();
        cv_done_.wait(lk, [this] { return released_ == (int)threads_.size(); });
    }

  private:
    void loop(int seen_epoch) {
        KdeWorkerState st;
        std::unique_lock<std::mutex> lk(m_);
        const int my_id = n_started_++;
        for (;;) {
            cv_job_.wait(lk, [&] { return stop_ || seen_epoch != release_epoch_ || (!queue_.empty() && my_id < limit_); });
            if (stop_) return;
            if (seen_epoch != release_epoch_) {
                seen_epoch = release_epoch_;
                lk.unlock();
                free_state(st);
                lk.lock();
                if (++released_ == (int)threads_.size()) cv_done_.notify_all();
                continue;
            }
            Task task = std::move(queue_.front().first);
            const void *tag = queue_.front().second;
            queue_.pop_front();
            lk.unlock();
            task(st);
            lk.lock();
            auto it = open_.find(tag);
            if (it != open_.end() && --it->second <= 0) open_.erase(it);
            --pending_;
            cv_done_.notify_all();   // (wait() looks at the count, wait_for() at the tags)
        }
    }
    std::mutex m_;
    std::condition_variable cv_job_, cv_done_;
    std::vector<std::thread> threads_;
    std::deque<std::pair<Task, const void *>> queue_;
    std::unordered_map<const void *, int> open_;   // job -> its runs queued or running
    int pending_ = 0, limit_ = 0, n_started_ = 0, release_epoch_ = 0, released_ = 0;
    bool stop_ = false;
};

static KdePool &pool() {
    static KdePool *p = new KdePool();   // never destroyed: its threads only ever wait on the queue at exit
    return *p;
}

}  // namespace pisa

using namespace pisa;

namespace pisa {

struct KdeBatchParams {
    int32_t dim, bw_method, adaptive;
    double alpha, tol;
    double origin[3], step[3];
    int64_t count[3];
    int64_t n_max;   // largest job of the submission: the workspaces are sized for it at once
    int device;
};

static int run_job(pisa_hip_kde_job &job, const KdeBatchParams &P, hipEvent_t ready, KdeWorkerState &st) {
    if (st.device != P.device) {
        // everything this thread holds belongs to the device it worked on last -- stream, workspaces AND the
        // estimator's thread-local scratch (kde.hip), which is tied to that stream: released there, in that order
        free_state(st);
        PISA_TRY_HIP(hipSetDevice(P.device));
        PISA_TRY_HIP(hipStreamCreateWithFlags(&st.stream, hipStreamNonBlocking));
        st.device = P.device;
    }
    hipStream_t s = st.stream;
    PISA_TRY_HIP(hipStreamWaitEvent(s, ready, 0));
    const int64_t need = pisa_hip_kde_workspace_bytes(P.dim, job.n);
    const int64_t need_max = pisa_hip_kde_workspace_bytes(P.dim, std::max(job.n, P.n_max));
    if (need < 0 || need_max < 0) return PISA_HIP_ERR_INVALID;
    const size_t wbytes = (((size_t)job.n * 8) + 255) & ~(size_t)255;
    int rc = grow(&st.work, &st.work_bytes, (size_t)need_max + ((((size_t)std::max(job.n, P.n_max) * 8) + 255) & ~(size_t)255), s);
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
    rc = pisa_hip_kde_create(P.dim, job.d_x, d_w, job.n, P.bw_method, P.adaptive, P.alpha, P.tol,
                             (char *)st.work + wbytes, need, &k, s);
    if (rc != PISA_HIP_OK) return rc;
    const int64_t lneed = pisa_hip_kde_lattice_workspace_bytes(k, P.step, P.count);
    // (the lattice workspace grows with the number of sources: scaled to the largest job of the submission)
    const size_t lneed_max = lneed < 0 ? 0 : (size_t)((double)lneed * (double)std::max(job.n, P.n_max) / (double)job.n);
    rc = lneed < 0 ? PISA_HIP_ERR_INVALID : grow(&st.lwork, &st.lwork_bytes, std::max((size_t)lneed, lneed_max), s);
    if (rc == PISA_HIP_OK)
        rc = pisa_hip_kde_evaluate_lattice(k, P.origin, P.step, P.count, st.lwork, lneed, job.d_out, s);
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
}

}  // namespace pisa

PISA_API int pisa_hip_kde_lattice_submit(pisa_hip_kde_job *jobs, int32_t n_jobs, int32_t dim, int32_t bw_method,
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
    KdeBatchParams P;
    memset(&P, 0, sizeof(P));
    P.dim = dim; P.bw_method = bw_method; P.adaptive = adaptive; P.alpha = alpha; P.tol = tol;
    for (int d = 0; d < dim; d++) { P.origin[d] = h_origin[d]; P.step[d] = h_step[d]; P.count[d] = h_count[d]; }
    for (int i = 0; i < n_jobs; i++) P.n_max = std::max(P.n_max, jobs[i].n);
    PISA_TRY_HIP(hipGetDevice(&P.device));
    // the inputs were produced on the caller's stream
    hipEvent_t ev;
    PISA_TRY_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    int rc0 = check_hip(hipEventRecord(ev, as_stream(stream)), "hipEventRecord");
    if (rc0 != PISA_HIP_OK) { (void)hipEventDestroy(ev); return rc0; }
    std::shared_ptr<void> ready(ev, [](void *e) { (void)hipEventDestroy((hipEvent_t)e); });
    std::vector<int> order(n_jobs);
    for (int i = 0; i < n_jobs; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return jobs[a].n > jobs[b].n; });   // longest first
    std::vector<KdePool::Task> tasks;
    std::vector<const void *> tags;
    for (int i : order) {
        pisa_hip_kde_job *job = jobs + i;
        tags.push_back(job);
        tasks.push_back([job, P, ready](KdeWorkerState &st) { job->status = run_job(*job, P, (hipEvent_t)ready.get(), st); });
    }
    pool().submit(std::move(tasks), tags, n_threads > 0 ? n_threads : 8);
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_kde_lattice_wait(void) {
    pool().wait();
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_kde_lattice_wait_jobs(const pisa_hip_kde_job *jobs, int32_t n_jobs) {
    if (n_jobs < 0 || (n_jobs > 0 && !jobs)) return PISA_HIP_ERR_INVALID;
    std::vector<const void *> tags((size_t)n_jobs);
    for (int i = 0; i < n_jobs; i++) tags[i] = jobs + i;
    pool().wait_for(tags.data(), n_jobs);
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_kde_pool_release(void) {
    pool().release();
    return PISA_HIP_OK;
}

PISA_API int pisa_hip_kde_lattice_batch(pisa_hip_kde_job *jobs, int32_t n_jobs, int32_t dim, int32_t bw_method,
                                        int32_t adaptive, double alpha, double tol, const double *h_origin,
                                        const double *h_step, const int64_t *h_count, int32_t n_threads,
                                        void *stream) {
    int rc = pisa_hip_kde_lattice_submit(jobs, n_jobs, dim, bw_method, adaptive, alpha, tol, h_origin, h_step, h_count,
                                         n_threads, stream);
    if (rc != PISA_HIP_OK) return rc;
    pool().wait();
    for (int i = 0; i < n_jobs; i++)
        if (jobs[i].status != PISA_HIP_OK) return jobs[i].status;
    return PISA_HIP_OK;
}
