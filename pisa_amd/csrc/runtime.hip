// runtime.hip -- status strings, device-memory helpers of the C ABI.
#include <string.h>

#include "common.hpp"

namespace pisa {
static thread_local char g_last_hip_error[512] = "";
void set_last_hip_error(hipError_t e, const char *what) {
    snprintf(g_last_hip_error, sizeof(g_last_hip_error), "%s: %s (%d)", what, hipGetErrorString(e),
             (int)e);
}
}  // namespace pisa

using namespace pisa;

PISA_API const char *pisa_hip_strerror(int status) {
    switch (status) {
    case PISA_HIP_OK: return "ok";
    case PISA_HIP_ERR_INVALID: return "invalid argument";
    case PISA_HIP_ERR_LAYERS: return "more than 120 layers (numba_osc_kernels.py:227)";
    case PISA_HIP_ERR_HIP: return "HIP runtime error";
    case PISA_HIP_ERR_GEOMETRY: return "unsupported Earth-model / detector geometry";
    case PISA_HIP_ERR_NEGATIVE: return "`actual_values`/`expected_values` must all be >= 0";
    case PISA_HIP_ERR_OVERFLOW: return "weight not finite or outside accumulator range (|w| >= 2^76)";
    case PISA_HIP_ERR_NOMEM: return "out of device memory";
    default: return "unknown status";
    }
}
PISA_API const char *pisa_hip_last_hip_error(void) { return g_last_hip_error; }
PISA_API int pisa_hip_version(void) { return 100; }
PISA_API int pisa_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
PISA_API int pisa_hip_set_device(int device) {
    PISA_TRY_HIP(hipSetDevice(device));
    return PISA_HIP_OK;
}
PISA_API int pisa_hip_malloc(void **d_ptr, int64_t bytes) {
    if (!d_ptr || bytes < 0) return PISA_HIP_ERR_INVALID;
    hipError_t e = hipMalloc(d_ptr, (size_t)(bytes ? bytes : 8));
    if (e == hipErrorOutOfMemory) { set_last_hip_error(e, "hipMalloc"); return PISA_HIP_ERR_NOMEM; }
    return check_hip(e, "hipMalloc");
}
PISA_API int pisa_hip_free(void *d_ptr) {
    if (!d_ptr) return PISA_HIP_OK;
    PISA_TRY_HIP(hipFree(d_ptr));
    return PISA_HIP_OK;
}
PISA_API int pisa_hip_memcpy_h2d(void *d_dst, const void *h_src, int64_t bytes, void *stream) {
    if (bytes < 0) return PISA_HIP_ERR_INVALID;
    if (bytes == 0) return PISA_HIP_OK;
    PISA_TRY_HIP(hipMemcpyAsync(d_dst, h_src, (size_t)bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return PISA_HIP_OK;
}
PISA_API int pisa_hip_memcpy_d2h(void *h_dst, const void *d_src, int64_t bytes, void *stream) {
    if (bytes < 0) return PISA_HIP_ERR_INVALID;
    if (bytes == 0) return PISA_HIP_OK;
    PISA_TRY_HIP(hipMemcpyAsync(h_dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    PISA_TRY_HIP(hipStreamSynchronize(as_stream(stream)));
    return PISA_HIP_OK;
}
PISA_API int pisa_hip_memset(void *d_dst, int value, int64_t bytes, void *stream) {
    if (bytes < 0) return PISA_HIP_ERR_INVALID;
    if (bytes == 0) return PISA_HIP_OK;
    PISA_TRY_HIP(hipMemsetAsync(d_dst, value, (size_t)bytes, as_stream(stream)));
    return PISA_HIP_OK;
}
PISA_API int pisa_hip_stream_synchronize(void *stream) {
    PISA_TRY_HIP(hipStreamSynchronize(as_stream(stream)));
    return PISA_HIP_OK;
}
