// prior.h -- device evaluation of the independent prior densities (shared by the model-specific MH kernels)
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/smc_hip.h"

namespace smc {

__device__ __forceinline__ double prior_quiet_nan() { return __longlong_as_double(0x7ff8000000000000LL); }

// scipy.stats pdf of one independent prior at x (Micmem_SMC_main.py:71-85).
//   uniform: support mask on the standardised value y = (x-loc)/scale, closed interval [0,1];
//   normal : exp(-y^2/2)/sqrt(2*pi)/scale.
__device__ __forceinline__ double prior_pdf(int kind, double a, double b, double x) {
    if (kind == SMC_PRIOR_FLAT) return 1.0;
    if (kind == SMC_PRIOR_UNIFORM) {
        const double scale = b - a;
        const double y = (x - a) / scale;
        if (x != x) return x;
        return (y >= 0.0 && y <= 1.0 && scale > 0.0) ? 1.0 / scale : 0.0;
    } else {
        const double y = (x - a) / b;
        if (!(b > 0.0)) return prior_quiet_nan();
        return exp(-(y * y) / 2.0) / 2.5066282746310002 / b;
    }
}


}  // namespace smc
