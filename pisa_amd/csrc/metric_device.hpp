// metric_device.hpp -- per-bin metric term shared by metric_kernel
// (metric_flux.hip) and the fused finalize+metric kernel (hist.hip).
#pragma once
#include "common.hpp"

namespace pisa {

constexpr double SMALL_POS = 1e-10;  // stats.py:40
constexpr double FTYPE_PREC = 2.220446049250313e-16;

__device__ __forceinline__ double metric_bin(int kind, double k, double lam, double s2) {
    // expected clipped to >= SMALL_POS (stats.py:154-155, 246-247, 319-320, 686-687)
    if (lam < SMALL_POS) lam = SMALL_POS;
    double v;
    switch (kind) {
    case PISA_HIP_METRIC_LLH:
        v = k * log(lam) - lam;
        v -= k * log(k) - k;  // k == 0 -> NaN, dropped by nansum
        break;
    case PISA_HIP_METRIC_POISSON_LLH:
        v = k * log(lam) - lam;
        v -= lgamma(k + 1);
        break;
    case PISA_HIP_METRIC_CHI2: {
        double d = k - lam;
        v = (d * d) / lam;
        break;
    }
    default: {
        double d = k - lam;
        v = (d * d) / (s2 + lam);
    }
    }
    return v;
}

}  // namespace pisa
