// rk45_math.h -- device arithmetic shared by the two RK45 likelihood kernels: the built-in Michaelis-Menten kernel
// (mm_rk45.h) and the run-time compiled user-model kernel (user_model.hip hands this file to hiprtc as an in-memory header):
// Python's min / max, min_step, the six-operation division and the fast inverse fifth root of the step controller.
// Device only; no #include that hiprtc could not resolve.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

namespace smc {

#define RK_MAX_ATTEMPTS (1 << 20)  // hard bound so that every wave drains; SciPy has none

// a / b with a shorter dependent chain than the compiler's IEEE sequence (v_div_scale, v_rcp, two
// Newton steps, v_div_fmas, v_div_fixup: ~30 ns on the chain).  v_rcp_f64 is accurate to 2^-24.4
// (measured, tools/div_probe.hip), so ONE Newton step gives 2^-48.8 and the Markstein correction
// q + (a - b*q)*r is within ~2^-97 relative of a/b before its single rounding: the rounded result is the correctly rounded
// quotient unless a/b lies within that distance of a rounding boundary - about one operand pair in 2^44, i.e. a
// percent-level chance per 10^6-particle run of ONE last-bit deviation from IEEE division somewhere, far inside every stated
// tolerance; parity mode (EXACT) divides with the IEEE sequence.  Measured: bit-identical to a/b on 4M wide-exponent operand
// pairs and 4M right-hand-side-shaped ones, both forms:
//   lean_div6  rcp -> e -> r -> q = a*r -> rem -> result: six dependent operations, six instructions.  The product path.
//   lean_div5  the Newton step applied to the QUOTIENT, q1 = q0 + q0*e with q0 = a*r0, beside the one on the reciprocal:
//              rcp -> e -> q1 -> rem -> result, five dependent operations but seven instructions.  Tried this round to
//              shorten the serial chain of a stiff solve, and NOT adopted: no consistent gain.  A 12 562-attempt solve
//              alone on the GPU (tools/attempt_probe.hip, both variants alternating in one process) came out at
//              0.419 / 0.460 us per attempt (six / five) on two boxes and at 0.462 / 0.448 on two others - the time of
//              a lone wave moves by +-10 % with where it lands and what else the chip is doing (same probe: 0.41 ... 0.52
//              us with 0 ... 250 busy blocks beside it, not monotonic) - and the bulk loop was 2 % slower with it
//              (profiles/r02_attempt_probe.log, r02_ab_lean_div.log).  The attempt is not a pure latency chain: ~200
//              vector instructions at >= 4 issue cycles each are ~2/3 of its ~1000 cycles, so an extra instruction per
//              division costs about what the dependent operation it removes saves.  Kept as a template option for the probe.
// No scaling and no special-case fix-up, so they are only used inside rk_attempt_core<DIV != 0>, whose caller
// re-runs the whole attempt with IEEE division whenever the result is not finite; quotients in the
// denormal range may differ from IEEE in the last bits (they sit > 280 orders below atol).
__device__ __forceinline__ double lean_div6(double a, double b) {
    double r = __builtin_amdgcn_rcp(b);
    const double e = fma(-b, r, 1.0);
    r = fma(r, e, r);
    const double q = a * r;
    const double rem = fma(-b, q, a);
    return fma(rem, r, q);
}
__device__ __forceinline__ double lean_div5(double a, double b) {
    const double r0 = __builtin_amdgcn_rcp(b);
    const double e = fma(-b, r0, 1.0);
    const double q0 = a * r0;            // beside e
    const double r = fma(r0, e, r0);     // beside q1
    const double q1 = fma(q0, e, q0);
    const double rem = fma(-b, q1, a);
    return fma(rem, r, q1);
}
// a / b for code that cannot re-run itself (user models, user_model.hip: smc_div): the six-operation division, and the
// compiler's IEEE sequence whenever that did not come out finite (a divisor that is subnormal or whose reciprocal is, an
// overflowing a * (1 / b), 0 / 0, ...).  Equal to a / b (up to lean_div6's one pair in 2^44) unless |b| >= 2^1021 (the reciprocal is subnormal and
// loses bits) or the quotient itself is subnormal (last bits).  On wave-uniform operands the test is one v_cmp_class and one
// scalar branch; the IEEE sequence costs ten more instructions and twice the latency on the serial chain of a stiff solve.
__device__ __forceinline__ double checked_lean_div(double a, double b) {
    const double q = lean_div6(a, b);
    if (__builtin_expect(!__builtin_isfinite(q), 0)) return a / b;
    return q;
}
constexpr int kDivIeee = 0, kDivLean6 = 1, kDivLean5 = 2;
// Python's min(a,b)/max(a,b): keep a unless b is strictly better (NaN never is)
__device__ __forceinline__ double py_min(double a, double b) { return (b < a) ? b : a; }
__device__ __forceinline__ double py_max(double a, double b) { return (b > a) ? b : a; }
// 10*|nextafter(t,inf)-t| for t >= 0 (rk.py:120)
__device__ __forceinline__ double min_step_of(double t) {
    const double up = __longlong_as_double(__double_as_longlong(t) + 1);
    return 10.0 * fabs(up - t);
}
__device__ __forceinline__ double quiet_nan() { return __longlong_as_double(0x7ff8000000000000LL); }

// x ** -0.2 for x >= 0 (the step-size controller's error_norm ** error_exponent, rk.py:104,155,169).
// The generic pow() costs 352 ns on the dependent chain of an attempt (measured, tools/div_probe.hip) -
// nearly half of it - and that chain is the serial critical path of a stiff solve (up to 10^5 dependent
// attempts).  Dedicated inverse fifth root: seed y0 = exp2(-0.2*log2(x)) from the hardware f32
// transcendentals (relative error <= 2^-20 for 2^-64 < x < 2^64), then ONE third-order correction
//   rho = 1 - x*y^5,   y <- y + y*rho*(1/5 + (3/25)*rho)      [ (1-rho)^(-1/5) = 1 + rho/5 + 3rho^2/25 + ... ]
// whose truncation error 0.09*rho^3 <= 2^-58.  Result within ~1.5 ulp - the same class as the
// libm-vs-device pow difference it replaces (the CPU checker under tests keeps libm pow).  Arguments
// outside the window are reduced first: x = m*2^e, e = 5q + r, m*2^r in [0.5, 16).
// EXACT (parity mode, smc_set_exact_pow): the fast value is finished to the correctly rounded pow(x, -0.2) - the function
// libm evaluates for SciPy, with the DOUBLE exponent -0.2 = -(1/5 + 1.1e-17) - by pow_fifth_exact.h: ~15 more operations
// on the chain, so only the host-RNG (parity) mode pays for it.
template <bool EXACT>
__device__ __forceinline__ double pow_minus_fifth_core(double x, double ln_scale /* ln of x's scaling removed by the caller */) {
    const float lf = __builtin_amdgcn_logf((float)x);                 // v_log_f32: log2
    const double y = (double)__builtin_amdgcn_exp2f(-0.2f * lf);      // v_exp_f32
    const double y2 = y * y;
    const double y5 = (y2 * y2) * y;
    const double rho = fma(-x, y5, 1.0);
    const double y1 = fma(y * rho, fma(0.12, rho, 0.2), y);
    if (!EXACT) return y1;
#ifdef SMC_HAVE_POW_FIFTH_EXACT      // pow_fifth_exact.h was included first (mm_rk45.h: parity mode)
    return pow_minus_fifth_finish(x, y1, fma((double)lf, 0.6931471805599453, ln_scale));
#else
    (void)ln_scale;
    return y1;
#endif
}
template <bool EXACT = false>
__device__ __forceinline__ double pow_minus_fifth(double x) {
    const unsigned e = ((unsigned)__double2hiint(x) >> 20) & 0xfffu;  // sign + biased exponent
    if (__builtin_expect((e - 959u) < 128u, 1)) return pow_minus_fifth_core<EXACT>(x, 0.0);   // 2^-64 <= x < 2^64
    if (!(x > 0.0)) return (x == 0.0) ? __longlong_as_double(0x7ff0000000000000LL) : quiet_nan();  // 0 -> inf
    if (x == __longlong_as_double(0x7ff0000000000000LL)) return 0.0;
    const double m = __builtin_amdgcn_frexp_mant(x);   // [0.5, 1)
    const int ex = __builtin_amdgcn_frexp_exp(x);
    const int q = (ex + 1075) / 5 - 215;               // floor(ex / 5), ex in [-1073, 1024]
    const int r = ex - 5 * q;                          // 0..4
    // x = (m 2^r) 2^(5q): the fifth root scales exactly; the 1.1e-17 excess of the double exponent sees the whole ln x
    return ldexp(pow_minus_fifth_core<EXACT>(ldexp(m, r), (5 * q) * 0.6931471805599453), -q);
}
// x ** (1 / 5) of select_initial_step (common.py:130): the reciprocal of the fast inverse root, or (EXACT) the correctly
// rounded pow(x, 0.2) with the double exponent
template <bool EXACT>
__device__ __forceinline__ double pow_plus_fifth(double x) {
    const double y0 = 1.0 / pow_minus_fifth<false>(x);
    if (!EXACT) return y0;
    const unsigned e = ((unsigned)__double2hiint(x) >> 20) & 0xfffu;
#ifdef SMC_HAVE_POW_FIFTH_EXACT
    if ((e - 959u) < 128u) return pow_plus_fifth_finish(x, y0, (double)__builtin_amdgcn_logf((float)x) * 0.6931471805599453);
#endif
    return pow(x, 0.2);   // outside 2^-64 .. 2^64: the library routine (never on the path of a sane model)
}

}  // namespace smc
