// mm_kernels.hip -- the dominant kernels of the path: the Michaelis-Menten likelihood sweep (K1) and
// the random-walk Metropolis iteration fused with it (K5).  gfx950 (MI355X) only.
//
// Mapping (wave64): one workgroup = 64 particles x n_ex experiments = n_ex waves; wave e solves
// experiment e for 64 CONSECUTIVE particles, so that
//   - the particle parameters are read as three coalesced 512-byte rows (SoA, d x N in HBM),
//   - the experiment's data (t, P_obs: 2 x n_t doubles) are wave-uniform LDS reads (broadcast when
//     the lanes are at the same output index, which neighbouring particles mostly are),
//   - lanes of a wave differ only in theta - after resampling neighbours are copies or close
//     relatives (ancestor order), which keeps their adaptive step counts aligned.
// The n_ex per-experiment sums meet in LDS; wave 0 adds them in experiment order (the order of the
// reference's `logL_total += logL_i`, Micmem_likelihood.py:73) and finishes the particle: plain
// likelihood (K1) or the Metropolis accept/select (K5, Micmem_SMC_main.py:220-241).
//
// Roofline: FP64 vector ALU (no contraction over a dimension > 7, hence no MFMA); HBM traffic is
// ~100 B per particle against ~2e4 flop (DESIGN.md "Kernels").
#include <hip/hip_runtime.h>

#include "mm_rk45.h"
#include "philox.h"
#include "smc_internal.h"

namespace smc {

// scipy.stats pdf of one independent prior at x (Micmem_SMC_main.py:71-85).
//   uniform: support mask on the standardised value y = (x-loc)/scale, closed interval [0,1];
//   normal : exp(-y^2/2)/sqrt(2*pi)/scale.
__device__ __forceinline__ double prior_pdf(int kind, double a, double b, double x) {
    if (kind == SMC_PRIOR_UNIFORM) {
        const double scale = b - a;
        const double y = (x - a) / scale;
        if (x != x) return x;
        return (y >= 0.0 && y <= 1.0 && scale > 0.0) ? 1.0 / scale : 0.0;
    } else {
        const double y = (x - a) / b;
        if (!(b > 0.0)) return __longlong_as_double(0x7ff8000000000000LL);
        return exp(-(y * y) / 2.0) / 2.5066282746310002 / b;
    }
}

template <int MODE /*0 = likelihood only, 1 = fused MH*/, bool WRITE_PRED>
__global__ void __launch_bounds__(1024)
mm_sweep_kernel(MMModel mm, Prior prior, const double *theta_in /* may alias theta_filt */, int64_t stride, int64_t n,
                double *lk_io, double *__restrict__ pred, MHParams mh, double *theta_filt,
                uint8_t *__restrict__ r_ac, SweepCounters *__restrict__ counters, double *__restrict__ dbg_prop,
                double *__restrict__ dbg_lk2, uint8_t *__restrict__ dbg_p0, uint8_t *__restrict__ dbg_r) {
    extern __shared__ double smem[];
    const int n_ex = mm.n_ex, n_t = mm.n_t;
    double *s_t = smem;                      // n_ex*n_t
    double *s_P = s_t + n_ex * n_t;          // n_ex*n_t
    double *s_part = s_P + n_ex * n_t;       // n_ex*64 partial sums of squares
    int *s_flag = (int *)(s_part + n_ex * kWave);  // n_ex*64: attempts | failed<<30

    for (int i = threadIdx.x; i < n_ex * n_t; i += blockDim.x) {
        s_t[i] = mm.t[i];
        s_P[i] = mm.P_obs[i];
    }
    __syncthreads();

    const int lane = threadIdx.x & (kWave - 1);
    const int e = threadIdx.x >> 6;  // wave index = experiment
    const int64_t p = (int64_t)blockIdx.x * kWave + lane;
    const bool valid = p < n;

    double th0 = 1.0, th1 = 1.0, th2 = 1.0;   // Vmax, Km, sigma of the point the likelihood is evaluated at
    double f0 = 0, f1 = 0, f2 = 0;            // current p_filt row (MODE 1)
    double p0 = 1.0;
    if (valid) {
        th0 = theta_in[p];
        th1 = theta_in[stride + p];
        th2 = theta_in[2 * stride + p];
        if (MODE == 1) {
            f0 = th0;
            f1 = th1;
            f2 = th2;
            // ---- proposal (Micmem_SMC_main.py:220): p_filt + noise * mhstep_ratio ----
            double z0, z1, z2;
            if (mh.device_rng) {
                const uint64_t g = (uint64_t)(mh.global_offset + p);
                const u32x4 ra = philox_block(mh.seed, g, mh.stream, 0);
                const u32x4 rb = philox_block(mh.seed, g, mh.stream, 1);
                // Box-Muller on (0,1] x [0,1)
                const double ua = 1.0 - u01_from(ra.x, ra.y), ub = u01_from(ra.z, ra.w);
                const double uc = 1.0 - u01_from(rb.x, rb.y), ud = u01_from(rb.z, rb.w);
                const double ra_ = sqrt(-2.0 * log(ua)), rc_ = sqrt(-2.0 * log(uc));
                double sa, ca, sc, cc;
                sincos(6.283185307179586 * ub, &sa, &ca);
                sincos(6.283185307179586 * ud, &sc, &cc);
                const double g0 = ra_ * ca, g1 = ra_ * sa, g2 = rc_ * cc;
                (void)sc;
                // x = z @ transform  (NumPy multivariate_normal: z @ (sqrt(s)[:,None]*v))
                const double *T = mh.transform;
                z0 = g0 * T[0] + g1 * T[3] + g2 * T[6];
                z1 = g0 * T[1] + g1 * T[4] + g2 * T[7];
                z2 = g0 * T[2] + g1 * T[5] + g2 * T[8];
            } else {
                z0 = mh.noise[p];
                z1 = mh.noise[n + p];
                z2 = mh.noise[2 * n + p];
            }
            // separately rounded multiply and add, as NumPy evaluates `p_filt + noise * mhstep_ratio`
            double c0 = __dadd_rn(f0, __dmul_rn(z0, mh.ratio)), c1 = __dadd_rn(f1, __dmul_rn(z1, mh.ratio)),
                   c2 = __dadd_rn(f2, __dmul_rn(z2, mh.ratio));
            // ---- support mask (cal_prior > 0, :225-226) and reset of out-of-support proposals (:228) ----
            double pdf = prior_pdf(prior.kind[0], prior.a[0], prior.b[0], c0);
            pdf = pdf * prior_pdf(prior.kind[1], prior.a[1], prior.b[1], c1);
            pdf = pdf * prior_pdf(prior.kind[2], prior.a[2], prior.b[2], c2);
            p0 = (pdf > 0.0) ? 1.0 : 0.0;
            const double q0 = 1.0 - p0;
            th0 = __dadd_rn(__dmul_rn(c0, p0), __dmul_rn(f0, q0));
            th1 = __dadd_rn(__dmul_rn(c1, p0), __dmul_rn(f1, q0));
            th2 = __dadd_rn(__dmul_rn(c2, p0), __dmul_rn(f2, q0));
        }
    }
    const double sigma = mm.est_sigma ? th2 : mm.sigma_fixed;
    const bool skip = !(sigma > 0.0) && !(sigma != sigma);  // sigma <= 0 -> -inf without solving (:53-54)

    if (valid && !skip) {
        double *pred_pe = WRITE_PRED ? pred + ((size_t)p * n_ex + e) * n_t : nullptr;
        MMSolveResult r = mm_solve_experiment<WRITE_PRED>(th0, th1, mm.S0[e], s_t + e * n_t, s_P + e * n_t, n_t,
                                                          mm.rtol, mm.atol, pred_pe);
        s_part[e * kWave + lane] = r.sum_r2;
        s_flag[e * kWave + lane] = r.attempts | (r.failed << 30);
    } else {
        s_part[e * kWave + lane] = 0.0;
        s_flag[e * kWave + lane] = 0;
        if (WRITE_PRED && valid) {
            double *pred_pe = pred + ((size_t)p * n_ex + e) * n_t;
            for (int i = 0; i < n_t; ++i) pred_pe[i] = __longlong_as_double(0x7ff8000000000000LL);
        }
    }
    __syncthreads();
    if (e != 0) return;

    // ---- wave 0: finish the particle ----
    unsigned long long attempts = 0, failed = 0, acc_now = 0, acc_ever = 0;
    if (valid) {
        double lk2;
        if (skip) {
            lk2 = -__longlong_as_double(0x7ff0000000000000LL);
        } else {
            const double s2 = sigma * sigma;
            const double c0 = (-0.5 * n_t) * log(2.0 * 3.141592653589793 * s2);  // :70
            lk2 = 0.0;
            for (int k = 0; k < n_ex; ++k) {
                lk2 += c0 - s_part[k * kWave + lane] / (2.0 * s2);               // :70-73
                const int fl = s_flag[k * kWave + lane];
                attempts += (unsigned)(fl & 0x3fffffff);
                failed |= (unsigned)(fl >> 30) & 1u;
            }
        }
        if (MODE == 0) {
            lk_io[p] = lk2;
        } else {
            // ---- accept / select (:231-241) ----
            const double lk1 = lk_io[p];
            double rr;
            if (mh.device_rng) {
                const u32x4 ru = philox_block(mh.seed, (uint64_t)(mh.global_offset + p), mh.stream,
                                              SMC_PHILOX_BLOCK_UNIFORM);
                rr = u01_from(ru.x, ru.y);
            } else {
                rr = mh.rr[p];
            }
            const double px = lk2 - lk1;
            const double pp = exp(px * mh.gamma) * p0;
            const double r = (pp >= rr) ? 1.0 : 0.0;
            const double nr = 1.0 - r;
            theta_filt[p] = __dadd_rn(__dmul_rn(th0, r), __dmul_rn(f0, nr));
            theta_filt[stride + p] = __dadd_rn(__dmul_rn(th1, r), __dmul_rn(f1, nr));
            theta_filt[2 * stride + p] = __dadd_rn(__dmul_rn(th2, r), __dmul_rn(f2, nr));
            lk_io[p] = __dadd_rn(__dmul_rn(lk2, r), __dmul_rn(lk1, nr));
            const uint8_t ever = (uint8_t)(r_ac[p] | (uint8_t)(r != 0.0));
            r_ac[p] = ever;
            acc_now = (r != 0.0);
            acc_ever = ever;
            if (dbg_prop) {
                dbg_prop[p] = th0;
                dbg_prop[n + p] = th1;
                dbg_prop[2 * n + p] = th2;
                dbg_lk2[p] = lk2;
                dbg_p0[p] = (uint8_t)(p0 != 0.0);
                dbg_r[p] = (uint8_t)(r != 0.0);
            }
        }
    }
    // wave-level integer reductions, then one atomic per block and counter
    for (int off = 32; off > 0; off >>= 1) {
        attempts += __shfl_down(attempts, off);
        failed += __shfl_down(failed, off);
        acc_now += __shfl_down(acc_now, off);
        acc_ever += __shfl_down(acc_ever, off);
    }
    if (lane == 0) {
        if (attempts) atomicAdd(&counters->rk_attempts, attempts);
        if (failed) atomicAdd(&counters->n_failed, failed);
        if (MODE == 1) {
            if (acc_now) atomicAdd(&counters->accepted_now, acc_now);
            if (acc_ever) atomicAdd(&counters->accepted_ever, acc_ever);
        }
    }
}

static size_t sweep_lds_bytes(const MMModel &mm) {
    return (size_t)(2 * mm.n_ex * mm.n_t + mm.n_ex * kWave) * sizeof(double) + (size_t)mm.n_ex * kWave * sizeof(int);
}

void launch_mm_loglik(smc_ctx *ctx, const double *theta, int64_t stride, int64_t n, double *lk, double *pred) {
    if (n <= 0) return;
    const MMModel &mm = ctx->mm;
    dim3 grid((unsigned)((n + kWave - 1) / kWave)), block(kWave * mm.n_ex);
    MHParams mh{};
    const size_t lds = sweep_lds_bytes(mm);
    if (pred)
        hipLaunchKernelGGL((mm_sweep_kernel<0, true>), grid, block, lds, ctx->stream, mm, ctx->prior, theta, stride, n,
                           lk, pred, mh, nullptr, nullptr, ctx->d_counters, nullptr, nullptr, nullptr, nullptr);
    else
        hipLaunchKernelGGL((mm_sweep_kernel<0, false>), grid, block, lds, ctx->stream, mm, ctx->prior, theta, stride, n,
                           lk, nullptr, mh, nullptr, nullptr, ctx->d_counters, nullptr, nullptr, nullptr, nullptr);
}

void launch_mm_mh(smc_ctx *ctx, int64_t n, const MHParams &mh) {
    if (n <= 0) return;
    const MMModel &mm = ctx->mm;
    dim3 grid((unsigned)((n + kWave - 1) / kWave)), block(kWave * mm.n_ex);
    const size_t lds = sweep_lds_bytes(mm);
    ParticleSet &F = ctx->set[SMC_SET_FILT];
    const bool dbg = ctx->debug_capture != 0;
    hipLaunchKernelGGL((mm_sweep_kernel<1, false>), grid, block, lds, ctx->stream, mm, ctx->prior, F.theta, F.stride, n,
                       F.lk, nullptr, mh, F.theta, ctx->r_ac, ctx->d_counters, dbg ? ctx->dbg_prop : nullptr,
                       dbg ? ctx->dbg_lk2 : nullptr, dbg ? ctx->dbg_p0 : nullptr, dbg ? ctx->dbg_r : nullptr);
}

}  // namespace smc
