// common.hpp -- shared host-side helpers of libpisa_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pisa_hip.h"

#define PISA_API extern "C" __attribute__((visibility("default")))

namespace pisa {

void set_last_hip_error(hipError_t e, const char *what);

inline int check_hip(hipError_t e, const char *what) {
    if (e != hipSuccess) {
        set_last_hip_error(e, what);
        return PISA_HIP_ERR_HIP;
    }
    return PISA_HIP_OK;
}

#define PISA_TRY_HIP(expr)                                       \
    do {                                                         \
        int _rc = ::pisa::check_hip((expr), #expr);              \
        if (_rc != PISA_HIP_OK) return _rc;                      \
    } while (0)

#define PISA_CHECK_LAUNCH(name)                                  \
    do {                                                         \
        int _rc = ::pisa::check_hip(hipGetLastError(), name);    \
        if (_rc != PISA_HIP_OK) return _rc;                      \
    } while (0)

// Development probes.  The product library reads NO environment variable and has no switch that makes a kernel do
// less work: every run-time selection of a kernel form used while developing (scripts/dev/*) exists only in builds
// made with `make EXTRA=-DPISA_DEV_PROBES`.  In the default build PISA_DEV_INT(name, dflt) is the constant `dflt`
// (the name never reaches the binary: tests/test_abi.py checks `strings libpisa_hip.so`).
#ifdef PISA_DEV_PROBES
#include <stdlib.h>
inline long long dev_env_ll(const char *name, long long dflt) {
    const char *v = getenv(name);
    return v ? atoll(v) : dflt;
}
inline const char *dev_env_str(const char *name) { return getenv(name); }
#define PISA_DEV_INT(name, dflt) ((int)::pisa::dev_env_ll("PISA_HIP_" name, (dflt)))
#define PISA_DEV_LL(name, dflt) (::pisa::dev_env_ll("PISA_HIP_" name, (dflt)))
#define PISA_DEV_STR(name) (::pisa::dev_env_str("PISA_HIP_" name))
#else
#define PISA_DEV_INT(name, dflt) ((int)(dflt))
#define PISA_DEV_LL(name, dflt) ((long long)(dflt))
#define PISA_DEV_STR(name) ((const char *)nullptr)
#endif

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Cross-kernel hand-over of the oscillation tables (development builds only, evaluator.hip: the round-6 experiment
// "accumulate kernel resident and polling while the chain kernel runs", EXPERIMENTS R6-3).  The evaluator sets
// `g_chain_signal` before the one-point chain launch -- one lane of each of its workgroups then adds 1 to counter
// (linear workgroup index mod HANDOVER_SLOTS) behind an agent-scope release, the launch code leaves its workgroup count in
// `n_wg` -- and `g_hist_wait` before the accumulate launch, whose workgroups poll all counters for epoch x (workgroups of
// that slot) between their first column loads and their first table gathers.
constexpr int HANDOVER_SLOTS = 64;
struct HandOver {
    unsigned long long *flags;   // HANDOVER_SLOTS counters, never reset
    unsigned long long epoch;    // chain launches that have signalled so far (this one included)
    int n_wg;                    // workgroups of one chain launch
};
#ifdef PISA_DEV_PROBES
extern thread_local HandOver g_chain_signal, g_hist_wait;
#endif

// Regular (linear, equal-width) binning as the kernels see it
// (fast_histogram rule / translation.py:417-456): bin = (int)((x - min) * norm)
struct DevBinning {
    int32_t ndim;
    int32_t nb[3];
    double mins[3];
    double maxs[3];
    double norm[3];
};

inline int make_dev_binning(const pisa_hip_binning *b, DevBinning &d, int64_t &total) {
    if (!b || b->ndim < 1 || b->ndim > PISA_HIP_MAX_DIMS) return PISA_HIP_ERR_INVALID;
    d.ndim = b->ndim;
    total = 1;
    for (int k = 0; k < 3; k++) {
        d.nb[k] = 1; d.mins[k] = 0; d.maxs[k] = 1; d.norm[k] = 1;
    }
    for (int k = 0; k < b->ndim; k++) {
        if (b->nbins[k] < 1 || b->nbins[k] > (1 << 24) || !(b->maxs[k] > b->mins[k]))
            return PISA_HIP_ERR_INVALID;
        d.nb[k] = (int32_t)b->nbins[k];
        d.mins[k] = b->mins[k];
        d.maxs[k] = b->maxs[k];
        // normx = nx / (xmax - xmin)   (translation.py:419; fast_histogram)
        d.norm[k] = (double)b->nbins[k] / (b->maxs[k] - b->mins[k]);
        total *= b->nbins[k];
    }
    return PISA_HIP_OK;
}

}  // namespace pisa
