// pow_fifth_exact.h -- correctly rounded pow(x, -0.2) and pow(x, 0.2) from an approximation that is good to a few ulp.
//
// Why: SciPy's step controller evaluates error_norm ** -0.2 (rk.py:155,169) and select_initial_step evaluates
// (0.01 / max(d1, d2)) ** (1 / 5) (common.py:130) with libm's pow, whose result is the correctly rounded one except in
// about one of 10^4 arguments (glibc >= 2.28: < 0.52 ulp).  The device's fast inverse fifth root (mm_rk45.h:
// pow_minus_fifth_core, <= 1.5 ulp) differs from it in the last bit in a good share of the arguments; on RK45's stability
// limit (thousands of attempts per solve) one such bit can flip one accept / reject decision of the controller, after which
// the device and the CPU checker follow different - equally valid - step sequences and logL agrees only to 1e-8 ... 1e-6
// (DESIGN.md "Parity in the stiff band").  In PARITY mode (host RNG; smc_set_exact_pow) the kernels therefore finish the
// fast value with the correction below; the device-RNG default keeps the fast form (the lone-chain latency matters there).
//
// WHICH function: Python evaluates `error_norm ** -0.2` and `x ** (1 / 5)` with the DOUBLES -0.2 and 0.2, and
//   double(0.2) = 0.2 + kDelta,   kDelta = 1.1102230246251565e-17,
// so what libm returns is x^-(1/5 + kDelta) = x^(-1/5) * (1 - kDelta ln x + ...), which is more than an ulp away from the
// fifth root for |ln x| > 10 (error norms of 1e-5 are ordinary).  The target here is that function, not the fifth root.
//
// Method: y0 ~ x^(-1/5) within a few ulp.  The residual r = x * y0^5 - 1 is formed in double-double arithmetic (error-free
// products through fma), so it is known to ~2^-100 while it is itself ~5 * (error of y0) <= 2^-49; one Newton step
//   y = y0 - y0 * r / 5                     [the neglected second-order term is 0.12 y0 r^2 < 2^-98 y0]
// is then accurate to ~2^-90 relative BEFORE the final addition, whose single rounding delivers the correctly rounded
// root unless the exact value lies within 2^-37 ulp of a midpoint between two doubles (probability ~2^-36 per call).
// The exponent's kDelta enters as the first-order factor (1 -+ kDelta ln x); ln x is needed to 2^-17 relative only (the
// term is < 2^-45), so the f32 logarithm that produced the seed serves.
// pow(x, 0.2) likewise from y0 ~ x^(1/5):  y = y0 - (y0^5 - x) / (5 y0^4) + y0 kDelta ln x.
// Host/device portable (the only #ifdef is the function qualifier): tests/hostcheck/pow_fifth_hostcheck.cpp compiles this
// file with g++ and compares 10^7 arguments with a 113-bit reference on the CPU, no GPU needed.
#pragma once
#include <math.h>

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define SMC_PF __host__ __device__ __forceinline__
#else
#define SMC_PF inline
#endif

namespace smc {

struct dd_t { double hi, lo; };   // value hi + lo, |lo| <= ulp(hi) / 2

SMC_PF dd_t dd_mul_d_d(double a, double b) {          // exact product of two doubles
    dd_t r;
    r.hi = a * b;
    r.lo = fma(a, b, -r.hi);
    return r;
}
SMC_PF dd_t dd_mul_dd_d(dd_t a, double b) {           // (hi + lo) * b, relative error ~2^-104
    dd_t r;
    r.hi = a.hi * b;
    r.lo = fma(a.lo, b, fma(a.hi, b, -r.hi));
    return r;
}
SMC_PF dd_t dd_sqr(dd_t a) {                          // (hi + lo)^2, the lo^2 term (2^-106) dropped
    dd_t r;
    r.hi = a.hi * a.hi;
    r.lo = fma(2.0 * a.hi, a.lo, fma(a.hi, a.hi, -r.hi));
    return r;
}
SMC_PF dd_t dd_pow5(double y) {
    const dd_t y2 = dd_mul_d_d(y, y);
    const dd_t y4 = dd_sqr(y2);
    return dd_mul_dd_d(y4, y);
}

constexpr double kFifthDelta = 1.1102230246251565404e-17;   // double(0.2) - 1/5

// correctly rounded pow(x, -0.2) [the double -0.2] for finite x > 0 with x, y0 and x*y0^5 in the normal range; y0 within a
// few ulp of x^(-1/5); ln_x: the natural logarithm of x to ~2^-17 relative
SMC_PF double pow_minus_fifth_finish(double x, double y0, double ln_x) {
    const dd_t q = dd_mul_dd_d(dd_pow5(y0), x);       // x * y0^5 = 1 + r
    const double r = (q.hi - 1.0) + q.lo;             // q.hi - 1 is exact (q.hi within 2^-48 of 1)
    return fma(fma(-0.2, r, -kFifthDelta * ln_x), y0, y0);
}
// correctly rounded pow(x, 0.2) [the double 0.2], same conditions with y0 ~ x^(1/5)
SMC_PF double pow_plus_fifth_finish(double x, double y0, double ln_x) {
    const dd_t y2 = dd_mul_d_d(y0, y0);
    const dd_t y4 = dd_sqr(y2);
    const dd_t y5 = dd_mul_dd_d(y4, y0);
    const double d = (y5.hi - x) + y5.lo;             // y5.hi - x is exact (Sterbenz: within a factor 2 of each other)
    return y0 + fma(y0, kFifthDelta * ln_x, -d / (5.0 * y4.hi));
}

}  // namespace smc
