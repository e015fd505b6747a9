// TEST TOOLING (tests/ only): compiles the product's host/device-portable correction csrc/pow_fifth_exact.h with g++ and
// checks it on the CPU against a 113-bit reference (__float128 / libquadmath): for seeds perturbed by up to +-3 ulp the
// finished value must be THE correctly rounded pow(x, -0.2) resp. pow(x, 0.2) with the DOUBLE exponents Python uses.  Also reports how often glibc's pow() differs from
// the correctly rounded value on the same arguments (the CPU checker oracle/smc_oracle.c uses pow, like SciPy).
// Not part of libsmc_hip.so; no product code path can reach it.
#include <quadmath.h>
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../../python-based-sequential-monte-carlo-method-with-likelihood-tempering_amd/csrc/pow_fifth_exact.h"

static inline double next_by(double v, int k) {
    int64_t b;
    std::memcpy(&b, &v, 8);
    b += k;
    std::memcpy(&v, &b, 8);
    return v;
}
static inline uint64_t xorshift(uint64_t &s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

extern "C" {
// n arguments log-uniform over [2^lo_exp, 2^hi_exp); out[0] = wrong results of pow_minus_fifth_finish, out[1] = of
// pow_plus_fifth_finish, out[2] = arguments where glibc pow(x, -0.2) is not the correctly rounded value, out[3] = same for
// pow(x, 0.2), out[4] = seeds that were already correct (sanity: the perturbation is real)
void pf_check(long n, int lo_exp, int hi_exp, uint64_t seed, long *out) {
    uint64_t s = seed ? seed : 88172645463325252ull;
    for (int k = 0; k < 5; ++k) out[k] = 0;
    for (long i = 0; i < n; ++i) {
        const double u = (double)(xorshift(s) >> 11) * 0x1p-53, v = (double)(xorshift(s) >> 11) * 0x1p-53;
        const double x = ldexp(1.0 + u, lo_exp + (int)(v * (hi_exp - lo_exp)));
        const __float128 xm = powq((__float128)x, (__float128)(-0.2)), xp = powq((__float128)x, (__float128)(0.2));   // double exponents
        const double rm = (double)xm, rp = (double)xp;          // conversion rounds to nearest: the correctly rounded values
        const int k = (int)(xorshift(s) % 7) - 3;               // seed error -3 .. +3 ulp about the FIFTH ROOT (what the device seed approximates)
        const double fm = (double)powq((__float128)x, (__float128)-0.2Q), fp = (double)powq((__float128)x, (__float128)0.2Q);
        const double ln_x = (double)logf((float)x);             // f32 accuracy, like the device's v_log_f32
        const double ym = smc::pow_minus_fifth_finish(x, next_by(fm, k), ln_x);
        const double yp = smc::pow_plus_fifth_finish(x, next_by(fp, k), ln_x);
        out[0] += (ym != rm);
        out[1] += (yp != rp);
        out[2] += (pow(x, -0.2) != rm);
        out[3] += (pow(x, 0.2) != rp);
        out[4] += (k == 0);
    }
}
double pf_minus(double x, double y0, double ln_x) { return smc::pow_minus_fifth_finish(x, y0, ln_x); }
double pf_plus(double x, double y0, double ln_x) { return smc::pow_plus_fifth_finish(x, y0, ln_x); }
}
