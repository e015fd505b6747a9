// mm_kernels.hip -- the dominant stage of the path: the Michaelis-Menten likelihood sweep
// (sim_particle, Micmem_likelihood.py:79-92) and the random-walk Metropolis iteration built around
// it (Micmem_SMC_main.py:220-241).  gfx950 (MI355X) only.
//
// A sweep is three launches on the context's stream:
//
//   mm_propose_kernel   (MH only)  proposal = p_filt + noise*ratio, prior support mask, reset of
//                       out-of-support proposals (:220-228).  HBM-bound, ~60 B/particle.
//   mm_solve_kernel     one work item = one (particle, experiment) RK45 solve -> sum of squared
//                       residuals.  FP64-VALU-bound (vector ALUs 86 % busy in the posterior phase): >99 % of
//                       the sweep.  PERSISTENT waves with LANE-LEVEL dynamic scheduling: a wave starts items
//                       48 or more at a time into a pool in LDS (chunks of 128 items come from one global
//                       atomic counter) and lanes that finish an item take the next started one from the
//                       pool.  Adaptive step counts differ by 100x between particles over the prior;
//                       static mapping made the early tempering steps 20-40x slower than the late ones
//                       (profiles/r01_a).  In Metropolis sweeps a solve is cancelled as soon as its proposal
//                       is certain to be rejected (mm_certainly_rejected).
//   mm_finish_kernel    per particle: the n_ex sums -> logL in experiment order (:70-73), then
//                       either store it (likelihood sweep) or accept/select (:231-241).
//
// Item order: item = (e * n_blocks64 + block64) * 64 + l  <->  particle block64*64 + l, experiment e: consecutive
// items share the experiment (same LDS rows, broadcast reads) and read consecutive particles (coalesced SoA rows), and
// the experiments of ONE particle lie n_blocks64 groups apart.  Stiffness is a property of the particle (the long solves
// have Vmax/Km > ~250: ~3.7 Vmax/Km T attempts) that shows in several of its experiments; with the particle-major order
// of round 1 two of them used to land in one wave, whose tail then could not take the single-item path below.
//
// Roofline: FP64 vector ALU (no contraction over a dimension > 7, hence no MFMA); HBM traffic of a
// whole sweep is ~200 B per particle against ~2e4 flop (DESIGN.md "Kernels").
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "mm_rk45.h"
#include "solve_sched.h"
#include "philox.h"
#include "prior.h"
#include "smc_internal.h"

namespace smc {

// info word of an item: attempts | cancelled << 29 | failed << 30
constexpr int kInfoAttemptsMask = 0x1fffffff, kInfoCancelled = 1 << 29, kInfoFailed = 1 << 30;
// sum_r2 of an item as other waves see it DURING a sweep with early rejection: NaN = not finished yet (or failed),
// kSumCancelled = stopped because its proposal is certainly rejected, >= 0 = finished
constexpr double kSumCancelled = -1.0;

// ---------------------------------------------------------------------------------------------
// the stiff list: solves that are predictably long start first
// ---------------------------------------------------------------------------------------------
// A solve with Vmax/Km = r runs on RK45's stability limit for ~3.7 r attempts (every solve above
// 10^3 attempts of a 10^6-particle prior sample has r > 269): a serial chain of up to 8e4 attempts = 35 ms that no amount
// of parallelism shortens.  What CAN be chosen is when it starts.  Handed out in index order, the chain that bounds a
// sweep began wherever the queue happened to reach it - up to a whole bulk pass (1.3 ms of a 1.7 ms Metropolis sweep,
// 4 ms of the initial sweep) late.  Particles above a threshold of that ratio are therefore collected into a list while the proposal
// is formed (or by mm_stiff_scan_kernel before a plain likelihood sweep) and mm_solve_kernel hands that list out before
// the index-ordered items.  The order in which independent items are solved cannot change any result - the reference's
// own fan-out (one Ray task per particle, Micmem_likelihood.py:83-87) leaves it to the scheduler as well.
// Threshold: every solve above 10^3 attempts has Vmax/Km > 269, but the sweeps of the middle of a run are bounded by chains
// of a few hundred attempts; handing out everything above ~220 attempts first is longest-first scheduling for them too.
// Whole run at 10^6 particles, one box (profiles/r03_ab_stiff_ratio.log): 1000 -> 94.5 ms, 250 -> 92.4, 120 -> 90.7, 60 -> 89.8,
// 30 -> 91.2, 15 -> 94.0, 8 -> 97.1 (the list then holds a third of the prior population and its 16-entry chunks cost dequeues).
#ifndef SMC_STIFF_RATIO
#define SMC_STIFF_RATIO 60.0
#endif
constexpr double kStiffRatio = SMC_STIFF_RATIO;
// ... and the stiffest of them run SOLO (solve_sched.h): one wave per solve on wave-uniform operands from the first attempt
// on.  Sharing a wave with 63 other live lanes a chain advances at ~0.55 us per attempt, alone at 0.43; the sweeps of a run
// are bounded by exactly such chains (3 000 attempts in mid-run, 10^4 in the first Metropolis sweeps, 10^5 in the initial
// sweep), so the ~1.3 ms the rest of the population keeps the queue busy cost every one of them ~25 % of that stretch.
// A solo solve wastes 63 lanes, hence only above a second threshold of the same ratio (profiles/r03_ab_solo.log).
#ifndef SMC_SOLO_RATIO
#define SMC_SOLO_RATIO 1000.0
#endif
constexpr double kSoloRatio = SMC_SOLO_RATIO;
static_assert(kSoloRatio >= kStiffRatio, "the index-ordered pass skips what either list holds by the stiff predicate alone");
__device__ __forceinline__ bool mm_is_stiff(double Vmax, double Km) { return Km > 0.0 && Vmax > kStiffRatio * Km; }
__device__ __forceinline__ bool mm_is_solo(double Vmax, double Km) { return Km > 0.0 && Vmax > kSoloRatio * Km; }
// One atomic per stiff lane on purpose: no cross-lane read follows it, so it is correct whether or not the compiler's
// atomic optimiser folds the wave's atomics into one.  Every particle is appended at most once, to one of the two lists,
// which grow towards each other in one array of n entries: they cannot overflow.
// The solo list is capped at what the grid of the sweep can run at once - one solo solve per wave (sl.solo_cap particles x
// n_ex experiments <= waves; a population of nothing but very stiff particles must not turn every solve into a one-lane
// wave, and a small grid must not queue solo solves behind each other): a particle that finds it full goes onto the
// ordinary list; the solve kernel reads min(count[1], solo_cap).
__device__ __forceinline__ void stiff_list_append(const StiffList &sl, int64_t p, double Vmax, double Km) {
    if (mm_is_solo(Vmax, Km)) {
        const unsigned k = atomicAdd(sl.count + 1, 1u);
        if (k < sl.solo_cap) {
            sl.particles[sl.cap - 1 - (int64_t)k] = (int32_t)p;
            return;
        }
    }
    sl.particles[atomicAdd(sl.count, 1u)] = (int32_t)p;
}
// ---------------------------------------------------------------------------------------------
// Cost order of a Metropolis sweep.  What a wave loses in a heterogeneous population is lanes (SQ counters on the proposals of
// a mid-run sweep: vector ALUs 86 % busy as ever, 59 % of the lanes active against 77-82 % in the posterior phase): an item
// of ~8 attempts still owes its 40 outputs, the output loop of a wave runs as long as the lane that has just taken the longest
// step needs, and unlike neighbours are never in phase.  The number of attempts of a solve is very nearly a function of
// Vmax / Km, so the proposals are binned by it - four classes per octave, descending, out-of-support proposals last - with a
// counting sort on the device (propose kernel: class byte; histogram per block of the same 256 contiguous slices, one block
// turns the table into offsets, scatter), and the index-ordered pass of the solve kernel hands out position -> order[position]:
// the 64 items a wave starts together are alike and - with the in-phase patience, which a wave drops as soon as one of its
// lanes holds a straggler (solve_sched.h: long_running) - stay in phase.  Proposals of sweep 12 of a 10^6-particle run
// (tools/sort_probe.py): 1.340 ms in the run's order, 1.178 physically sorted, 1.193 through order[], 1.008 with patience 12.
// (Not quite monotone: below Km ~ 1.5e-3 SciPy's RK45 - and so this kernel - finishes a solve in 5-10 steps, tests/attempts_map.py;
// those few proposals sit in the "longest" classes and on the stiff / solo lists and simply finish at once.)
// The order of independent solves changes no result; the order inside a class depends on atomics and need not repeat.
// ---------------------------------------------------------------------------------------------
constexpr int kCostBuckets = 128, kCostBlocks = 256;
struct alignas(32) SortedProposal {
    double Vmax, Km, sigma;
    long long particle;
};
__device__ __forceinline__ unsigned mm_cost_bucket(double Vmax, double Km, bool in_support) {
    if (!in_support) return kCostBuckets - 1;     // the ONLY members of the last class: published by the propose kernel, never solved
    // In support but with no meaningful ratio (Km <= 0 or a NaN: reachable under a normal or flat prior on Km, and in
    // SMC_PRIOR_MODE_RATIO where every proposal counts as in support): a real class, so that the proposal gets a position in
    // sorted[] and is solved like any other (it used to share class 127 and nobody wrote its sums).
    if (!(Km > 0.0) || !(Vmax > 0.0)) return 0u;
    // exponent and two mantissa bits of the single-precision ratio: four classes per octave between 2^-12 and 2^19
    const int u = (int)(__float_as_uint((float)(Vmax / Km)) >> 21) - (127 - 12) * 4;
    const int k = u < 0 ? 0 : (u > 123 ? 123 : u);
    return (unsigned)(123 - k);         // long solves first
}
// Table of the counting sort (ctx->d_order_hist, unsigned words):  counts[kCostBlocks][kCostBuckets] - zero between sweeps: the
// offsets kernel clears what it has read - | offs[kCostBlocks][kCostBuckets] | totals[kCostBuckets] | n_ordered.
// A slice is a whole number of 256-particle blocks, so that every block of the propose kernel belongs to exactly one slice and
// can add its LDS histogram to that slice's row (round 4: the separate histogram launch is gone).
constexpr int kCostCounts = 0, kCostOffs = kCostBlocks * kCostBuckets, kCostTotals = 2 * kCostBlocks * kCostBuckets,
              kCostNOrdered = kCostTotals + kCostBuckets, kCostTableWords = kCostNOrdered + 64;
__host__ __device__ __forceinline__ int64_t cost_slice_blocks(int64_t n) { return (((n + 255) / 256) + kCostBlocks - 1) / kCostBlocks; }
__host__ __device__ __forceinline__ int64_t cost_slice_particles(int64_t n) { return cost_slice_blocks(n) * 256; }
size_t cost_table_bytes() { return (size_t)kCostTableWords * sizeof(unsigned); }
const unsigned *cost_n_ordered(const smc_ctx *ctx) { return ctx->d_order_hist + kCostNOrdered; }

// histogram of class bytes somebody else wrote (user_model.hip: the scan kernel of a model with a cost hint)
__global__ void __launch_bounds__(256) cost_hist_kernel(const uint8_t *__restrict__ bucket, int64_t n, unsigned *__restrict__ table) {
    __shared__ unsigned h[kCostBuckets];
    if (threadIdx.x < kCostBuckets) h[threadIdx.x] = 0u;
    __syncthreads();
    const int64_t per = cost_slice_particles(n), lo = per * blockIdx.x, hi = (lo + per < n) ? lo + per : n;
    for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) atomicAdd(&h[bucket[i]], 1u);
    __syncthreads();
    if (threadIdx.x < kCostBuckets) table[kCostCounts + (int64_t)blockIdx.x * kCostBuckets + threadIdx.x] = h[threadIdx.x];
}
// one block per class, one thread per slice: offs[b][k] <- members of class k in the slices below b; totals[k]; counts <- 0
// (round 3: one block of 128 threads walked the 256 rows one after the other, 8.5 us on the critical path of every sweep)
__global__ void __launch_bounds__(kCostBlocks) cost_offsets_kernel(unsigned *__restrict__ table, const MHControl *__restrict__ ctl) {
    __shared__ unsigned wsum[kCostBlocks / 64];
    if (ctl && ctl->stop) return;      // the Metropolis loop of this batch has ended (stage_kernels.hip: mh_control_kernel)
    const int k = blockIdx.x, b = threadIdx.x, w = b >> 6, l = b & 63;
    const unsigned v = table[kCostCounts + b * kCostBuckets + k];
    table[kCostCounts + b * kCostBuckets + k] = 0u;
    unsigned inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(inc, off);
        if (l >= off) inc += o;
    }
    if (l == 63) wsum[w] = inc;
    __syncthreads();
    unsigned base = 0, tot = 0;
    for (int q = 0; q < kCostBlocks / 64; ++q) {
        if (q < w) base += wsum[q];
        tot += wsum[q];
    }
    table[kCostOffs + b * kCostBuckets + k] = base + inc - v;
    if (b == 0) table[kCostTotals + k] = tot;
}
// first position of every class from the totals (each block of a scatter kernel for itself: 128 words), then this slice's cursors
__device__ __forceinline__ void cost_cursors(unsigned *__restrict__ table, unsigned *cur /* LDS, kCostBuckets */, unsigned *tot /* LDS */) {
    if (threadIdx.x < kCostBuckets) tot[threadIdx.x] = table[kCostTotals + threadIdx.x];
    __syncthreads();
    if (threadIdx.x < kCostBuckets) {
        unsigned base = 0;
        for (int q = 0; q < (int)threadIdx.x; ++q) base += tot[q];
        cur[threadIdx.x] = table[kCostOffs + (int64_t)blockIdx.x * kCostBuckets + threadIdx.x] + base;
        if (blockIdx.x == 0 && threadIdx.x == kCostBuckets - 1) table[kCostNOrdered] = base;   // positions the solve kernel hands out
    }
    __syncthreads();
}
// ... and the proposals themselves (rows [c * stride + i]) into cost order, as 32-byte records (Vmax, Km, sigma, particle): one
// scattered 32-byte write per proposal here instead of one scattered line read per value and experiment in the solve kernel
// (whose counter traffic went from 339 to 832 MB per launch with the gathers, profiles/r03_ab_cost_order.log)
__global__ void __launch_bounds__(256) cost_scatter_kernel(const uint8_t *__restrict__ bucket, int64_t n, unsigned *__restrict__ table,
                                                          const double *__restrict__ theta, int64_t stride,
                                                          SortedProposal *__restrict__ sorted, const MHControl *__restrict__ ctl) {
    __shared__ unsigned cur[kCostBuckets], tot[kCostBuckets];
    if (ctl && ctl->stop) return;
    cost_cursors(table, cur, tot);
    const int64_t per = cost_slice_particles(n), lo = per * blockIdx.x, hi = (lo + per < n) ? lo + per : n;
    for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        SortedProposal r;
        r.Vmax = theta[i];
        r.Km = theta[stride + i];
        r.sigma = theta[2 * stride + i];
        r.particle = i;
        sorted[atomicAdd(&cur[bucket[i]], 1u)] = r;
    }
}

// ... or, for a model the library knows nothing about (user_model.hip: the class bytes come from the model's cost hint), just the
// particle indices: position -> particle, gathered by the solve kernel
__global__ void __launch_bounds__(256) cost_scatter_order_kernel(const uint8_t *__restrict__ bucket, int64_t n,
                                                                unsigned *__restrict__ table, int32_t *__restrict__ order) {
    __shared__ unsigned cur[kCostBuckets], tot[kCostBuckets];
    cost_cursors(table, cur, tot);
    const int64_t per = cost_slice_particles(n), lo = per * blockIdx.x, hi = (lo + per < n) ? lo + per : n;
    for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) order[atomicAdd(&cur[bucket[i]], 1u)] = (int32_t)i;
}
// ctx->d_bucket (class bytes of n proposals, 127 = needs no solve) -> ctx->d_order; returns where the solve kernel finds the
// number of positions (the first position of the last class), or nullptr when the context has no buffers for it
const unsigned *launch_cost_sort_order(smc_ctx *ctx, int64_t n) {
    if (!ctx->d_order || !ctx->d_bucket || !ctx->d_order_hist) return nullptr;
    hipLaunchKernelGGL(cost_hist_kernel, dim3(kCostBlocks), dim3(256), 0, ctx->stream, ctx->d_bucket, n, ctx->d_order_hist);
    hipLaunchKernelGGL(cost_offsets_kernel, dim3(kCostBuckets), dim3(kCostBlocks), 0, ctx->stream, ctx->d_order_hist, (const MHControl *)nullptr);
    hipLaunchKernelGGL(cost_scatter_order_kernel, dim3(kCostBlocks), dim3(256), 0, ctx->stream, ctx->d_bucket, n, ctx->d_order_hist, ctx->d_order);
    return cost_n_ordered(ctx);
}

__global__ void __launch_bounds__(256)
mm_stiff_scan_kernel(const double *__restrict__ theta, int64_t stride, int64_t n, StiffList sl) {
    if (blockIdx.x == 0 && threadIdx.x < 2) sl.count_next[threadIdx.x] = 0u;
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n && mm_is_stiff(theta[p], theta[stride + p])) stiff_list_append(sl, p, theta[p], theta[stride + p]);
}

// ---------------------------------------------------------------------------------------------
// proposal (Micmem_SMC_main.py:220-228)
// ---------------------------------------------------------------------------------------------
// one proposal (Micmem_SMC_main.py:220-228); s_hist: the block's class histogram in LDS (cost-ordered sweeps)
__device__ __forceinline__ void mm_propose_one(const Prior &prior, const MHParams &mh, double mh_ratio, const double *__restrict__ filt,
                                               int64_t stride, int64_t n, double *__restrict__ prop, int64_t pstride,
                                               uint8_t *__restrict__ p0_out, int64_t p, unsigned *s_hist) {
    const double f0 = filt[p], f1 = filt[stride + p], f2 = filt[2 * stride + p];
    double z0, z1, z2;
    if (mh.device_rng) {
        const uint64_t g = (uint64_t)(mh.global_offset + p);
        const u32x4 ra = philox_block(mh.seed, g, mh.stream, 0);
        const u32x4 rb = philox_block(mh.seed, g, mh.stream, 1);
        // Box-Muller on (0,1] x [0,1)
        const double ua = 1.0 - u01_from(ra.x, ra.y), ub = u01_from(ra.z, ra.w);
        const double uc = 1.0 - u01_from(rb.x, rb.y), ud = u01_from(rb.z, rb.w);
        const double ra_ = sqrt(-2.0 * log(ua)), rc_ = sqrt(-2.0 * log(uc));
        double sa, ca;
        sincos(6.283185307179586 * ub, &sa, &ca);
        const double g0 = ra_ * ca, g1 = ra_ * sa, g2 = rc_ * cos(6.283185307179586 * ud);
        // x = z @ transform  (NumPy multivariate_normal: z @ (sqrt(s)[:,None]*v))
        const double *T = mh.transform_dev ? mh.transform_dev : mh.transform;
        z0 = g0 * T[0] + g1 * T[3] + g2 * T[6];
        z1 = g0 * T[1] + g1 * T[4] + g2 * T[7];
        z2 = g0 * T[2] + g1 * T[5] + g2 * T[8];
    } else {
        z0 = mh.noise[p];
        z1 = mh.noise[n + p];
        z2 = mh.noise[2 * n + p];
    }
    // separately rounded multiply and add, as NumPy evaluates `p_filt + noise * mhstep_ratio`
    const double c0 = __dadd_rn(f0, __dmul_rn(z0, mh_ratio)), c1 = __dadd_rn(f1, __dmul_rn(z1, mh_ratio)),
                 c2 = __dadd_rn(f2, __dmul_rn(z2, mh_ratio));
    // support mask (cal_prior > 0, :225-226) and reset of out-of-support proposals (:228)
    double pdf = prior_pdf(prior.kind[0], prior.a[0], prior.b[0], c0);
    pdf = pdf * prior_pdf(prior.kind[1], prior.a[1], prior.b[1], c1);
    pdf = pdf * prior_pdf(prior.kind[2], prior.a[2], prior.b[2], c2);
    if (mh.prior_mode != SMC_PRIOR_MODE_MASK) {   // p0_2 / p0_1 (SMC_methanation_main.py:323-324, 343)
        double cur = prior_pdf(prior.kind[0], prior.a[0], prior.b[0], f0);
        cur = cur * prior_pdf(prior.kind[1], prior.a[1], prior.b[1], f1);
        cur = cur * prior_pdf(prior.kind[2], prior.a[2], prior.b[2], f2);
        mh.pratio[p] = pdf / cur;
    }
    const double p0 = (pdf > 0.0 || mh.prior_mode == SMC_PRIOR_MODE_RATIO) ? 1.0 : 0.0, q0 = 1.0 - p0;
    const double w0 = __dadd_rn(__dmul_rn(c0, p0), __dmul_rn(f0, q0)), w1 = __dadd_rn(__dmul_rn(c1, p0), __dmul_rn(f1, q0));
    prop[p] = w0;
    prop[pstride + p] = w1;
    prop[2 * pstride + p] = __dadd_rn(__dmul_rn(c2, p0), __dmul_rn(f2, q0));
    p0_out[p] = (uint8_t)(p0 != 0.0);
    if (mh.cost_bucket) {
        const unsigned cls = mm_cost_bucket(w0, w1, p0 != 0.0);
        mh.cost_bucket[p] = (uint8_t)cls;
        atomicAdd(&s_hist[cls], 1u);
    }
    // early rejection reads the siblings' results: none of this sweep's items has finished yet (NaN).  In a cost-ordered sweep
    // the solve kernel never sees an out-of-support proposal: its items are published here (sum 0, no attempts)
    const bool done_here = mh.cost_bucket && p0 == 0.0;
    if (mh.pending_sums || done_here)
        for (int k = 0; k < mh.pending_n_ex; ++k) {
            if (done_here) {
                mh.done_sums[(int64_t)k * n + p] = 0.0;
                mh.done_info[(int64_t)k * n + p] = 0;
            } else {
                mh.pending_sums[(int64_t)k * n + p] = __longlong_as_double(0x7ff8000000000000LL);
            }
        }
    // a proposal inside the support whose solves will be long: onto the list of this sweep (the solve kernel applies the
    // same predicate to the same stored values when it skips the particle in its index-ordered pass)
    if (mh.stiff.particles && p0 != 0.0 && mm_is_stiff(w0, w1)) stiff_list_append(mh.stiff, p, w0, w1);
}
__global__ void __launch_bounds__(256)
mm_propose_kernel(Prior prior, MHParams mh, const double *__restrict__ filt, int64_t stride, int64_t n,
                  double *__restrict__ prop, int64_t pstride, uint8_t *__restrict__ p0_out) {
    // batch of iterations under device control: nothing of an iteration after the loop's `break` may happen - no counter is
    // cleared, no list touched, p_pred keeps the proposals of the last iteration that ran (Micmem_SMC_main.py:243-246)
    if (mh.ctl && mh.ctl->stop) return;
    const double mh_ratio = mh.ctl ? mh.ctl->ratio : mh.ratio;      // mhstep_ratio (:190,247-249)
    if (blockIdx.x == 0) {   // fused iteration: the counters of this sweep and the work queue of its solve start from zero
        if (mh.zero_counters && threadIdx.x < sizeof(SweepCounters) / 8)
            reinterpret_cast<unsigned long long *>(mh.zero_counters)[threadIdx.x] = 0ull;
        if (mh.zero_queue && threadIdx.x == 64) mh.zero_queue[0] = 0ull;
        if (mh.stiff.particles && (threadIdx.x == 65 || threadIdx.x == 67)) mh.stiff.count_next[threadIdx.x == 67] = 0u;
        if (mh.reject_out && threadIdx.x == 66) {   // what mm_certainly_rejected reads during the solve of this sweep
            RejectArgs r;
            r.lk1 = mh.reject_lk1;
            r.rr = mh.rr;
            r.pratio = mh.pratio;
            r.gamma = mh.gamma;
            r.seed = mh.seed;
            r.stream = mh.stream;
            r.global_offset = mh.global_offset;
            r.device_rng = mh.device_rng;
            r.prior_mode = mh.prior_mode;
            *mh.reject_out = r;
        }
    }
    // cost-ordered sweep: the class histogram of this block's 256 proposals, added to the row of the slice the block belongs to
    // (the counting sort's first pass; its table is zero on entry, see cost_offsets_kernel)
    __shared__ unsigned s_hist[kCostBuckets];
    if (mh.cost_bucket) {
        if (threadIdx.x < kCostBuckets) s_hist[threadIdx.x] = 0u;
        __syncthreads();
    }
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) mm_propose_one(prior, mh, mh_ratio, filt, stride, n, prop, pstride, p0_out, p, s_hist);
    if (mh.cost_bucket) {
        __syncthreads();
        if (threadIdx.x < kCostBuckets && s_hist[threadIdx.x])
            atomicAdd(&mh.cost_table[kCostCounts + ((int64_t)blockIdx.x / cost_slice_blocks(n)) * kCostBuckets + threadIdx.x], s_hist[threadIdx.x]);
    }
}

// ---------------------------------------------------------------------------------------------
// solve: persistent waves, lane-level dynamic scheduling (solve_sched.h) of Michaelis-Menten items
// ---------------------------------------------------------------------------------------------
constexpr int kSolveBlock = 256;   // 4 waves
// An item that needed more attempts than this is "long": the accept kernel counts them, and a Metropolis sweep whose
// predecessor had (almost) none runs its waves in phase (SolveArgs::patience, solve_sched.h).  A posterior-like solve takes
// 19 +- a few attempts, the stragglers of a prior-like population hundreds to thousands.
constexpr int kLongItemAttempts = 64;
constexpr int kPoolWords = 15;     // 8-byte words of a pooled item: 11 doubles, 2 packed int pairs, out_idx, prediction pointer

struct SolveArgs {              // everything the attempt loops do not touch stays behind a pointer (RejectArgs, StiffList)
    const double *theta;        // SoA rows: Vmax at [p], Km at [stride + p], sigma at [2*stride + p]
    int64_t stride, n;
    const uint8_t *p0;          // MH: support flags; items of particles with p0 == 0 are not solved
    double *sum_r2;             // [e*n + p]
    int *info;                  // [e*n + p]: attempts | cancelled << 29 | failed << 30
    double *pred;               // optional: P_model, [(p*n_ex + e)*n_t + i]
    unsigned long long *queue;  // global item counter (zeroed before the launch)
    const RejectArgs *rej;      // exact early rejection (Metropolis sweeps only; nullptr: off): see mm_certainly_rejected()
    const int32_t *stiff_list;  // particles whose solves are handed out first (nullptr: none), and how many: [0] in the
    const unsigned *stiff_count;    // list proper (from the front of the array), [1] solo (from its back, stiff_cap - 1 downwards)
    int64_t stiff_cap;
    unsigned solo_cap;
    int patience;               // solve_sched.h: attempts a wave waits for all its lanes before a hand-out (homogeneous sweeps)
    const int32_t *order;       // cost order of the sweep: position of the index-ordered pass -> particle (nullptr: identity)
    const unsigned *n_ordered;  // ... and how many positions it has: the in-support proposals (the others were published by the
                                // propose kernel and come last in `order`); nullptr: all n (a probe's uploaded order)
    const SortedProposal *sorted;   // with n_ordered: the proposals in cost order, one 32-byte record per position (written by the
                                    // counting sort's scatter): a wave's 64 starts read 2 KB in a row instead of 192 scattered lines
    const MHControl *ctl;           // batch of iterations under device control: the kernel leaves at once when ctl->stop is set
};

// A kernel argument as a scalar register of its OWN.  The kernel's arguments arrive in 4-, 8- and 16-dword loads, i.e. as
// register TUPLES; when the allocator runs out of scalar registers it spills and reloads a tuple as a whole - in round 4's
// kernel the bulk attempt loop reloaded all 16 dwords of one (16 v_readlane per attempt, vector-ALU slots in an issue-bound loop)
// to get at rtol and atol, and 4 + 4 more for the two pointers of publish_item (tools/bulk_loop_report.py).  The empty asm makes
// the value opaque: what the loop uses is a 1- or 2-dword register that can be kept or spilled on its own.
#ifdef SMC_NO_OWN_SGPR   // A/B builds: round 4's kernel (arguments used straight out of their tuples, no waves-per-SIMD request)
template <class T> __device__ __forceinline__ T own_sgpr(T v) { return v; }
#else
__device__ __forceinline__ double own_sgpr(double v) {
    double r;
    asm volatile("s_mov_b64 %0, %1" : "=s"(r) : "s"(v));      // (an explicit move: a tied "+s" operand is coalesced back into the tuple)
    return r;
}
template <class T>
__device__ __forceinline__ T *own_sgpr(T *p) {
    T *r;
    asm volatile("s_mov_b64 %0, %1" : "=s"(r) : "s"(p));
    return r;
}
__device__ __forceinline__ int own_sgpr(int v) {
    int r;
    asm volatile("s_mov_b32 %0, %1" : "=s"(r) : "s"(v));
    return r;
}
#endif

// one experiment's term of logL (Micmem_likelihood.py:70-73), shared by the accept kernel and by the rejection bound so that
// both evaluate the same floating-point expression
__device__ __forceinline__ double mm_loglik_term(double c0, double sum_r2, double s2) { return c0 - sum_r2 / (2.0 * s2); }

// The sum of an item must become visible to waves on other XCDs while the kernel runs (L2 is not coherent across XCDs):
// ONE relaxed agent-scope 8-byte store (write-through, no fence: a release per item tripled the time of a sweep).  The
// sum itself says whether the item is finished (see kSumCancelled), so no ordering with the info word is needed; the info
// word is only read by the accept kernel, after this kernel has ended.
__device__ __forceinline__ void publish_item(double *sums, int *infos, int64_t idx, double sum_r2, int info) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(sums) + idx, (unsigned long long)__double_as_longlong(sum_r2),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    infos[idx] = info;
}
__device__ __forceinline__ void publish_item(const SolveArgs &a, int64_t idx, double sum_r2, int info) {
    publish_item(a.sum_r2, a.info, idx, sum_r2, info);
}

// EXACT early rejection.  The accept test of the sweep (Micmem_SMC_main.py:231-236) is  exp((lk2 - lk1) * gamma) [* p0_2/p0_1]
// >= rr  with lk2 = sum over the experiments of c0 - sum_r2_e / (2 sigma^2), and both lk1 and rr are known before the solve
// (device RNG: the uniform is a pure function of the Philox key).  sum_r2_e only grows while a solve runs, so replacing the
// sums of the unfinished experiments by what they have accumulated so far (0 for a sibling still running elsewhere) gives
// an UPPER bound of lk2 - in floating point too, because every operation of the expression is monotone and the bound runs
// through the same expression in the same order.  If even the bound fails the test, the proposal is rejected whatever the
// rest of the solve would add: the solve can stop, and nothing observable changes (p_filt, lk1, the accept flags and counts
// are those of the full computation; only the attempt counters are smaller).  The long solves of a sweep are proposals with
// Vmax/Km in the thousands; their other experiments finish within microseconds and fit so badly that the bound decides
// 469 of 470 of them (CPU replay of a 1e6-particle run) before the long solve has produced a single output.
// Returns true only when rejection is certain; any NaN makes the comparison false.  `cancel_seen`: a sibling was already
// cancelled, i.e. the particle is known to be rejected.
__device__ __forceinline__ bool mm_certainly_rejected(const MMModel &mm, const SolveArgs &a, int64_t p, int e_self,
                                                      double partial_self) {
    const double sigma = mm.est_sigma ? a.theta[2 * a.stride + p] : mm.sigma_fixed;
    if (!(sigma > 0.0)) return false;
    const double s2 = sigma * sigma;
    const double c0 = (-0.5 * mm.n_t) * log(2.0 * 3.141592653589793 * s2);
    double lk2_bound = 0.0;
    for (int k = 0; k < mm.n_ex; ++k) {
        double S = 0.0;
        if (k == e_self) {
            S = partial_self;
        } else {
            const double v = __longlong_as_double((long long)__hip_atomic_load(
                reinterpret_cast<unsigned long long *>(a.sum_r2) + (int64_t)k * a.n + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (v < 0.0) return true;      // kSumCancelled: a sibling has already established the rejection
            if (v == v) S = v;             // finished; NaN = still running (or failed): counts as 0
        }
        lk2_bound += mm_loglik_term(c0, S, s2);
    }
    const RejectArgs &r = *a.rej;     // device memory, written by the propose kernel of this sweep
    double rr;
    if (r.device_rng) {
        const u32x4 ru = philox_block(r.seed, (uint64_t)(r.global_offset + p), r.stream, SMC_PHILOX_BLOCK_UNIFORM);
        rr = u01_from(ru.x, ru.y);
    } else {
        rr = r.rr[p];
    }
    double pp = exp((lk2_bound - r.lk1[p]) * r.gamma);
    if (r.prior_mode != SMC_PRIOR_MODE_MASK) pp = pp * r.pratio[p];
    return pp < rr * (1.0 - 1e-12);   // the margin covers a last-bit non-monotonicity of exp
}

// What solve_sched.h needs to know about a Michaelis-Menten item (see the list at the top of that file).
// EXACT: parity mode (smc_set_exact_pow) - the step controller's power is the correctly rounded pow(x, -0.2), mm_rk45.h.
// FAST: lone chains run the hand-written attempt loop (mm_rk45.h: mm_fast_uniform_attempts).  A template parameter, not a flag:
// selected at run time the block's registers cost the kernel its fourth wave per SIMD (140 instead of 128 VGPRs); as its own
// instantiation it fits into 128, but its bulk loop is still ~6 % slower than the plain one's (register allocation around the
// block), which a sweep over 10^6 particles does not notice and one over 10^7 does: launch_solve picks by the size of the sweep.
template <bool WRITE_PRED, bool EXACT, bool FAST>
struct MMOps {
    struct Item {
        MMItem s;
        int64_t out_idx;    // e*n + p
        double *pred;       // WRITE_PRED: where its predictions go
    };
    static constexpr int kPoolWords = smc::kPoolWords;
    const MMModel &mm;
    const SolveArgs &a;
    const double2 *s_tp;    // LDS: n_ex rows of n_t + 1 (time, P_obs) pairs (mm_rk45.h)
    const double *s_S0;     // LDS
    long long n;
    int n_ex, n_t;
    const int32_t *list;
    unsigned n_list;
    const int32_t *solo;        // n_solo particles at solo[0], solo[-1], ... (the back of the list array)
    unsigned n_solo;
    double rtol, atol;
    int patience;
    long long n_pos;            // positions of the index-ordered pass
    double *pub_sums;           // a.sum_r2 / a.info as registers of their own (own_sgpr): what finish() inside the attempt loop uses
    int *pub_info;

    __device__ __forceinline__ int start(long long p, int e, bool from_list, Item &nb) const {
        // masked proposal: lk2 == lk1, no solve (a cost-ordered sweep holds none: the propose kernel has published them)
        const bool masked = !a.n_ordered && a.p0 && a.p0[p] == 0;
        return start_values(p, e, from_list, masked, a.theta[p], a.theta[a.stride + p], mm.est_sigma ? a.theta[2 * a.stride + p] : mm.sigma_fixed, nb);
    }
    __device__ __forceinline__ int start_values(long long p, int e, bool from_list, bool masked, double Vmax, double Km, double sigma,
                                                Item &nb) const {
        nb.out_idx = (int64_t)e * a.n + p;
        nb.pred = nullptr;
        // index-ordered pass: a particle of the stiff list has been handed out already
        if (!from_list && list && !masked && mm_is_stiff(Vmax, Km)) return kStartSkipped;
        if (masked || sigma <= 0.0) {                    // sigma <= 0: -inf without solving (:53-54)
            publish_item(a, nb.out_idx, 0.0, 0);
            if (WRITE_PRED) {
                double *pp = a.pred + ((size_t)p * n_ex + e) * n_t;
                for (int i = 0; i < n_t; ++i) pp[i] = quiet_nan();
            }
            return kStartDone;
        }
        if (WRITE_PRED) nb.pred = a.pred + ((size_t)p * n_ex + e) * n_t;
        if (mm_item_begin<WRITE_PRED, EXACT>(nb.s, Vmax, Km, s_S0[e], s_tp, mm_table_row(e, n_t), n_t, rtol, atol, nb.pred))
            return kStartStarted;
        const bool ok = (nb.s.i_out == n_t);             // nothing to integrate: finished at once
        publish_item(a, nb.out_idx, ok ? nb.s.sum_r2 : quiet_nan(), ok ? 0 : kInfoFailed);
        return kStartDone;
    }
    // word-major slots so that lanes reading or writing consecutive slots hit consecutive banks
    __device__ __forceinline__ void pack(const Item &nb, double *slot) const {
        slot[0 * kWave] = nb.s.negVmax;
        slot[1 * kWave] = nb.s.Km;
        slot[2 * kWave] = nb.s.S0;
        slot[3 * kWave] = nb.s.t;
        slot[4 * kWave] = nb.s.y;
        slot[5 * kWave] = nb.s.f;
        slot[6 * kWave] = nb.s.h_abs;
        slot[7 * kWave] = nb.s.min_step;
        slot[8 * kWave] = nb.s.t_bound;
        slot[9 * kWave] = nb.s.t_next;
        slot[10 * kWave] = nb.s.sum_r2;
        slot[11 * kWave] = __hiloint2double(nb.s.i_out, nb.s.t_off);
        slot[12 * kWave] = __hiloint2double((int)nb.s.rejected, nb.s.attempts);
        slot[13 * kWave] = __longlong_as_double((long long)nb.out_idx);
        if (WRITE_PRED) slot[14 * kWave] = __longlong_as_double((long long)nb.pred);
    }
    __device__ __forceinline__ void unpack(Item &it, const double *slot) const {
        it.s.negVmax = slot[0 * kWave];
        it.s.Km = slot[1 * kWave];
        it.s.S0 = slot[2 * kWave];
        it.s.t = slot[3 * kWave];
        it.s.y = slot[4 * kWave];
        it.s.f = slot[5 * kWave];
        it.s.h_abs = slot[6 * kWave];
        it.s.min_step = slot[7 * kWave];
        it.s.t_bound = slot[8 * kWave];
        it.s.t_next = slot[9 * kWave];
        it.s.sum_r2 = slot[10 * kWave];
        const double w11 = slot[11 * kWave], w12 = slot[12 * kWave];
        it.s.t_off = __double2loint(w11);
        it.s.i_out = __double2hiint(w11);
        it.s.attempts = __double2loint(w12);
        it.s.rejected = __double2hiint(w12) != 0;
        it.out_idx = (int64_t)__double_as_longlong(slot[13 * kWave]);
        it.pred = WRITE_PRED ? (double *)__double_as_longlong(slot[14 * kWave]) : nullptr;
    }
    __device__ __forceinline__ int attempt(Item &it) const {
        return mm_item_attempt<WRITE_PRED, kDivLean6, EXACT>(it.s, s_tp, n_t, rtol, atol, it.pred);
    }
    __device__ __forceinline__ bool long_running(const Item &it) const { return it.s.attempts > kLongItemAttempts; }
    __device__ __forceinline__ long long positions() const { return n_pos; }
    __device__ __forceinline__ int start_at(long long pos, int e, Item &nb) const {
        if (!a.n_ordered) return start(a.order ? (long long)a.order[pos] : pos, e, false, nb);   // identity, or a probe's uploaded order
        const SortedProposal r = a.sorted[pos];
        return start_values(r.particle, e, false, false, r.Vmax, r.Km, mm.est_sigma ? r.sigma : mm.sigma_fixed, nb);
    }
    // The lone chain (solve_sched.h: solo phase, uniform tail): the hand-written loop of mm_rk45.h for the attempts of a stiff
    // solve that neither produce an output nor hit a special case, mm_item_attempt for the others.  The block is entered only
    // where it pays - the next data time at least four steps away (an attempt that turns out to need an output is computed
    // twice) - and never in parity mode (its arithmetic is the default mode's).
    __device__ __forceinline__ int uniform_attempts(Item &it, int budget) const {
        if (EXACT || !FAST) return uniform_attempts_plain(*this, it, budget);
        int st = 0;
        do {
            SMC_ISA_MARK("uniform_tail_attempt");
            MMItem &s = it.s;
            if (s.t_next - s.t > 4.0 * s.h_abs && s.attempts + budget < RK_MAX_ATTEMPTS) {
                const int more = mm_fast_uniform_attempts(s, rtol, atol, budget);
                // the block's results come back in vector registers: tell the compiler that they are wave-uniform
                s.t = lane_value(s.t, 0);
                s.y = lane_value(s.y, 0);
                s.f = lane_value(s.f, 0);
                s.h_abs = lane_value(s.h_abs, 0);
                s.min_step = lane_value(s.min_step, 0);
                if (!more) break;          // budget used up
            }
            st = attempt(it);
        } while (st == 0 && --budget > 0);
        return st;
    }
    __device__ __forceinline__ void finish(Item &it, int st) const {
        const bool ok = (st == 1) && (it.s.i_out == n_t);
        publish_item(pub_sums, pub_info, it.out_idx, ok ? it.s.sum_r2 : quiet_nan(), it.s.attempts | (ok ? 0 : kInfoFailed));
        if (WRITE_PRED && !ok)
            for (int i = it.s.i_out; i < n_t; ++i) it.pred[i] = quiet_nan();
    }
    __device__ __forceinline__ Item broadcast(const Item &it, int src) const {
        Item u;
        u.s.negVmax = lane_value(it.s.negVmax, src);
        u.s.Km = lane_value(it.s.Km, src);
        u.s.S0 = lane_value(it.s.S0, src);
        u.s.t = lane_value(it.s.t, src);
        u.s.y = lane_value(it.s.y, src);
        u.s.f = lane_value(it.s.f, src);
        u.s.h_abs = lane_value(it.s.h_abs, src);
        u.s.min_step = lane_value(it.s.min_step, src);
        u.s.t_bound = lane_value(it.s.t_bound, src);
        u.s.t_next = lane_value(it.s.t_next, src);
        u.s.sum_r2 = lane_value(it.s.sum_r2, src);
        u.s.t_off = __builtin_amdgcn_readlane(it.s.t_off, src);
        u.s.i_out = __builtin_amdgcn_readlane(it.s.i_out, src);
        u.s.attempts = __builtin_amdgcn_readlane(it.s.attempts, src);
        u.s.rejected = __builtin_amdgcn_readlane((int)it.s.rejected, src) != 0;
        u.out_idx = lane_value_ll(it.out_idx, src);
        u.pred = WRITE_PRED ? (double *)lane_value_ll((long long)it.pred, src) : nullptr;
        return u;
    }
    __device__ __forceinline__ bool reject_enabled() const { return a.rej != nullptr; }
    __device__ __forceinline__ bool certainly_rejected(const Item &it) const {
        const int e_self = (int)(it.out_idx / a.n);
        return mm_certainly_rejected(mm, a, it.out_idx - (int64_t)e_self * a.n, e_self, it.s.sum_r2);
    }
    __device__ __forceinline__ void cancel(Item &it) const {
        publish_item(a, it.out_idx, kSumCancelled, it.s.attempts | kInfoCancelled);
    }
};

template <bool WRITE_PRED, bool EXACT, bool FAST>
__device__ __forceinline__ void mm_solve_body(const MMModel &mm, const SolveArgs &a) {
    extern __shared__ double2 smem_tp[];
    if (a.ctl && __builtin_amdgcn_readfirstlane(a.ctl->stop)) return;   // the Metropolis loop has ended: nothing to solve (scalar branch)
    const int n_ex = mm.n_ex, n_t = mm.n_t;
    double2 *s_tp = smem_tp;                                        // n_ex rows of n_t + 1 (time, P_obs) pairs, mm_rk45.h
    double *s_S0 = reinterpret_cast<double *>(s_tp + n_ex * (n_t + 1));   // n_ex
    mm_table_fill(s_tp, mm.t, mm.P_obs, n_ex, n_t, threadIdx.x, blockDim.x);
    if (threadIdx.x < n_ex) s_S0[threadIdx.x] = mm.S0[threadIdx.x];
    __syncthreads();
    // the wave's pool of STARTED items (solve_sched.h): a ring of 64 slots of kPoolWords words
    double *s_pool = s_S0 + ((n_ex + 1) & ~1) + (size_t)(threadIdx.x >> 6) * (kPoolWords * kWave);
    const unsigned n_stiff = a.stiff_list ? (unsigned)__builtin_amdgcn_readfirstlane((int)a.stiff_count[0]) : 0u;
    unsigned n_solo = a.stiff_list ? (unsigned)__builtin_amdgcn_readfirstlane((int)a.stiff_count[1]) : 0u;
    if (n_solo > a.solo_cap) n_solo = a.solo_cap;   // the overflow went onto the ordinary list (stiff_list_append)
    MMOps<WRITE_PRED, EXACT, FAST> ops{mm, a, s_tp, s_S0, (long long)a.n, n_ex, own_sgpr(n_t), a.stiff_list, n_stiff,
                                 a.stiff_list ? a.stiff_list + (a.stiff_cap - 1) : nullptr, n_solo, own_sgpr(mm.rtol), own_sgpr(mm.atol),
                                 own_sgpr(a.patience),
                                 a.n_ordered ? (long long)__builtin_amdgcn_readfirstlane((int)a.n_ordered[0]) : (long long)a.n,
                                 own_sgpr(a.sum_r2), own_sgpr(a.info)};
    solve_persistent(ops, a.queue, s_pool);
}
// The kernel: the primary template serves the parity arithmetic (EXACT: 164 VGPRs, three waves per SIMD); the default-mode
// instantiations are explicit specialisations that ask for FOUR waves per SIMD (128 VGPRs) - with the kernel arguments of the
// attempt loop in registers of their own the allocator otherwise lands on 130.
template <bool WRITE_PRED, bool EXACT, bool FAST>
__global__ void __launch_bounds__(kSolveBlock) mm_solve_kernel(MMModel mm, SolveArgs a) {
    mm_solve_body<WRITE_PRED, EXACT, FAST>(mm, a);
}
#ifndef SMC_NO_OWN_SGPR
#define SMC_SOLVE_DEFAULT_MODE(WP, F)                                                                                              \
    template <>                                                                                                                    \
    __global__ void __launch_bounds__(kSolveBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) mm_solve_kernel<WP, false, F>(MMModel mm, SolveArgs a) { \
        mm_solve_body<WP, false, F>(mm, a);                                                                                        \
    }
SMC_SOLVE_DEFAULT_MODE(false, false)     // (the WRITE_PRED instantiations - predictions for the drop-in's plots - stay with the primary template)
SMC_SOLVE_DEFAULT_MODE(false, true)
#undef SMC_SOLVE_DEFAULT_MODE
#endif

// ---------------------------------------------------------------------------------------------
// finish: logL from the per-experiment sums, then store or accept/select
// ---------------------------------------------------------------------------------------------
template <int MODE /*0 = likelihood only, 1 = MH accept/select*/>
__global__ void __launch_bounds__(256)
mm_finish_kernel(MMModel mm, MHParams mh, const double *__restrict__ theta /* evaluated point (proposal) */,
                 int64_t stride, int64_t n, const double *__restrict__ sum_r2, const int *__restrict__ info,
                 const uint8_t *__restrict__ p0_in, double *lk_io, double *filt, int64_t fstride,
                 uint8_t *__restrict__ r_ac, SweepCounters *__restrict__ counters, double *__restrict__ dbg_lk2,
                 uint8_t *__restrict__ dbg_r) {
    __shared__ unsigned long long s_cnt[4][4];
    __shared__ double s_mom[4][9];
    if (MODE == 1 && mh.ctl && mh.ctl->stop) return;   // after the loop's `break`: p_filt, lk1, r_ac and the counters stay as they are
    unsigned long long attempts = 0, failed = 0, acc_now = 0, acc_ever = 0, long_items = 0;
    // (items whose solve ran to t_bound and produced its n_t dense outputs - bench.py's roofline numerator - are counted in a
    // 32-bit register and ride in the HIGH half of `failed` through the block reduction: the kernel's time is its per-block tail)
    unsigned solved = 0;
    // moments of the SELECTED particles about mh.moment_shift (MODE 1, fused iteration): sum y, sum y y^T (upper), y = x - shift
    double m0 = 0, m1 = 0, m2 = 0, c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
    const bool acc_mom = (MODE == 1) && mh.moment_rows != nullptr;
    double sh0 = 0, sh1 = 0, sh2 = 0;
    if (acc_mom) {
        sh0 = mh.moment_shift[0];
        sh1 = mh.moment_shift[1];
        sh2 = mh.moment_shift[2];
    }
    // grid-stride: a few hundred blocks, so that the counters cost one atomic per block, not per wave
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const int n_ex = mm.n_ex;
        const double sigma = mm.est_sigma ? theta[2 * stride + p] : mm.sigma_fixed;
        const bool masked = (MODE == 1) && (p0_in[p] == 0);
        bool cancelled = false;   // a solve of this proposal stopped because its rejection was certain (mm_certainly_rejected)
        double lk2;
        if (masked) {
            lk2 = lk_io[p];  // proposal was reset to the current point: the likelihood is the stored one
        } else if (sigma <= 0.0) {
            lk2 = -__longlong_as_double(0x7ff0000000000000LL);  // Micmem_likelihood.py:53-54
        } else {
            const double s2 = sigma * sigma;
            const double c0 = (-0.5 * mm.n_t) * log(2.0 * 3.141592653589793 * s2);  // :70
            lk2 = 0.0;
            unsigned pf = 0;
            for (int k = 0; k < n_ex; ++k) {
                lk2 += mm_loglik_term(c0, sum_r2[(int64_t)k * n + p], s2);          // :70-73
                const int fl = info[(int64_t)k * n + p];
                attempts += (unsigned)(fl & kInfoAttemptsMask);
                long_items += (unsigned)(fl & kInfoAttemptsMask) > (unsigned)kLongItemAttempts;
                pf |= (unsigned)(fl >> 30) & 1u;
                cancelled = cancelled || (fl & kInfoCancelled) != 0;
#ifndef SMC_NO_SOLVED_COUNT   // A/B builds
                solved += (fl & kInfoCancelled) == 0;
#endif
            }
            failed += pf;
        }
        if (MODE == 0) {
            lk_io[p] = lk2;
        } else {
            // ---- accept / select (:231-241) ----
            const double lk1 = lk_io[p];
            if (cancelled) lk2 = lk1;   // rejected for certain: its logL was never completed and is not needed (r = 0 below)
            const double p0 = masked ? 0.0 : 1.0;
            double rr;
            if (mh.device_rng) {
                const u32x4 ru = philox_block(mh.seed, (uint64_t)(mh.global_offset + p), mh.stream,
                                              SMC_PHILOX_BLOCK_UNIFORM);
                rr = u01_from(ru.x, ru.y);
            } else {
                rr = mh.rr[p];
            }
            const double px = lk2 - lk1;
            double pp = exp(px * mh.gamma);
            if (mh.prior_mode != SMC_PRIOR_MODE_MASK) pp = pp * mh.pratio[p];
            if (mh.prior_mode != SMC_PRIOR_MODE_RATIO) pp = pp * p0;
            const double r = (!cancelled && pp >= rr) ? 1.0 : 0.0;
            const double nr = 1.0 - r;
            double sel[3];
            for (int c = 0; c < 3; ++c) {
                const double th = theta[c * stride + p], f = filt[c * fstride + p];
                sel[c] = __dadd_rn(__dmul_rn(th, r), __dmul_rn(f, nr));
                filt[c * fstride + p] = sel[c];
            }
            if (acc_mom) {
                const double y0 = sel[0] - sh0, y1 = sel[1] - sh1, y2 = sel[2] - sh2;
                m0 += y0; m1 += y1; m2 += y2;
                c00 += y0 * y0; c01 += y0 * y1; c02 += y0 * y2; c11 += y1 * y1; c12 += y1 * y2; c22 += y2 * y2;
            }
            lk_io[p] = __dadd_rn(__dmul_rn(lk2, r), __dmul_rn(lk1, nr));
            const uint8_t ever = (uint8_t)(r_ac[p] | (uint8_t)(r != 0.0));
            r_ac[p] = ever;
            acc_now += (r != 0.0);
            acc_ever += ever;
            if (dbg_lk2) {
                dbg_lk2[p] = lk2;
                dbg_r[p] = (uint8_t)(r != 0.0);
            }
        }
    }
    failed += (unsigned long long)solved << 32;
    // integer reductions (order-independent): wave shuffles, LDS across the 4 waves, one atomic per block
    for (int off = 32; off > 0; off >>= 1) long_items += __shfl_down(long_items, off);
    if ((threadIdx.x & 63) == 0 && long_items) atomicAdd(&counters->long_items, long_items);
    for (int off = 32; off > 0; off >>= 1) {
        attempts += __shfl_down(attempts, off);
        failed += __shfl_down(failed, off);
        acc_now += __shfl_down(acc_now, off);
        acc_ever += __shfl_down(acc_ever, off);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_cnt[w][0] = attempts;
        s_cnt[w][1] = failed;
        s_cnt[w][2] = acc_now;
        s_cnt[w][3] = acc_ever;
    }
    if (acc_mom) {   // fixed-order block sums -> one row of 9 per block (deterministic; reduced by moments_reduce_kernel)
        double mv[9] = {m0, m1, m2, c00, c01, c02, c11, c12, c22};
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            double v = mv[q];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if ((threadIdx.x & 63) == 0) s_mom[w][q] = v;
        }
    }
    __syncthreads();
    if (acc_mom && threadIdx.x < 9)
        mh.moment_rows[(size_t)blockIdx.x * 9 + threadIdx.x] =
            ((s_mom[0][threadIdx.x] + s_mom[1][threadIdx.x]) + s_mom[2][threadIdx.x]) + s_mom[3][threadIdx.x];
    if (threadIdx.x < 4) {
        unsigned long long v = s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];
        if (threadIdx.x == 1) {      // low half: failed solves; high half: solves that produced their outputs
            if (v >> 32) atomicAdd(&counters->solved_items, v >> 32);
            v &= 0xffffffffULL;
        }
        unsigned long long *dst = threadIdx.x == 0 ? &counters->rk_attempts
                                  : threadIdx.x == 1 ? &counters->n_failed
                                  : threadIdx.x == 2 ? &counters->accepted_now : &counters->accepted_ever;
        if (v) atomicAdd(dst, v);
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
// Blocks of the accept kernel: grid-stride, so that counters and moment rows cost one atomic / one row per block.  More blocks
// do NOT help (A/B on one box, whole run of 10^6 particles, profiles/r04_ab_finish_grid.log: 1024 -> 74.6 ms, 2048 -> 75.1,
// 4096 -> 76.3): the kernel's time goes into the per-block tail - nine 6-step wave reductions of the moments and the counters -
// not into its 170 B per particle.
#ifndef SMC_FINISH_GRID_CAP
#define SMC_FINISH_GRID_CAP 1024
#endif
static unsigned finish_grid(int64_t n) {
    static const int64_t cap = [] {
        const char *e = getenv("SMC_FINISH_GRID_CAP");     // A/B knob
        const int64_t v = e ? atoll(e) : 0;
        return (v >= 1 && v <= 8192) ? v : (int64_t)SMC_FINISH_GRID_CAP;
    }();
    const int64_t g = (n + 255) / 256;
    return (unsigned)(g < cap ? (g < 1 ? 1 : g) : cap);
}

// dynamic LDS of the solve kernel: the (time, observation) table, S0, and the four waves' pools of started items
static size_t solve_lds_bytes(int n_ex, int n_t) {
    return (size_t)n_ex * (n_t + 1) * sizeof(double2) + (size_t)((n_ex + 1) & ~1) * sizeof(double) +
           (size_t)(kSolveBlock / kWave) * kPoolWords * kWave * sizeof(double);
}

// Persistent grid of a sweep over n particles: enough blocks to fill every CU at the kernel's occupancy; for a small
// population as many as its chunks need plus one wave per item, so that every solo solve finds a wave of its own (blocks
// beyond the work find the queue empty and leave at once).
// Up to this many particles per sweep the lone chains run the hand-written loop (the FAST instantiations).  Whole runs on one
// GPU: 10^6 particles 86.3 -> 76.7 ms with it, 10^7 particles 430 -> 440 ms, 10^8 3.48 -> 3.69 s (profiles/r03_ab_fast_tail.log):
// the chains do not grow with the population, the bulk that pays for the lost wave does.
#ifndef SMC_FAST_TAIL_MAX
#define SMC_FAST_TAIL_MAX 4000000
#endif
constexpr int64_t kFastTailMaxParticles = SMC_FAST_TAIL_MAX;
static bool use_fast_tail(const smc_ctx *ctx, int64_t n) { return ctx->fast_tail != 0 && ctx->exact_pow == 0 && n <= kFastTailMaxParticles; }

static int64_t solve_grid_blocks(const smc_ctx *ctx, int64_t n) {
    const int64_t waves_per_block = kSolveBlock / kWave;
    const int64_t items = ((n + kWave - 1) / kWave) * kWave * ctx->mm.n_ex;
    const int64_t chunks = (items + kChunk - 1) / kChunk;
    const int64_t need = (chunks + waves_per_block - 1) / waves_per_block + (n * ctx->mm.n_ex + waves_per_block - 1) / waves_per_block;
    int64_t blocks = (int64_t)ctx->cu_count * (use_fast_tail(ctx, n) ? ctx->solve_blocks_per_cu_fast : ctx->solve_blocks_per_cu);
    if (blocks > need) blocks = need;
    return blocks < 1 ? 1 : blocks;
}

// The lists of the next sweep: their counters, and the other pair for the kernel that builds the lists to clear.
static StiffList next_stiff_list(smc_ctx *ctx, int64_t n) {
    StiffList sl{};
    if (!ctx->stiff_first || !ctx->d_stiff_list) return sl;
    ctx->stiff_parity ^= 1;
    sl.particles = ctx->d_stiff_list;
    sl.count = ctx->d_stiff_count + 2 * ctx->stiff_parity;
    sl.count_next = ctx->d_stiff_count + 2 * (ctx->stiff_parity ^ 1);
    sl.cap = ctx->item_cap;
    sl.solo_cap = (unsigned)(solve_grid_blocks(ctx, n) * (kSolveBlock / kWave) / ctx->mm.n_ex);   // one solo solve per wave
    return sl;
}

static void launch_solve(smc_ctx *ctx, const double *theta, int64_t stride, int64_t n, const uint8_t *p0, double *pred,
                         const StiffList &sl, bool queue_cleared = false, bool reject = false, int patience = 0,
                         const int32_t *order = nullptr, bool cost_ordered = false, const MHControl *ctl = nullptr) {
    const MMModel &mm = ctx->mm;
    SolveArgs a{};
    a.theta = theta;
    a.stride = stride;
    a.n = n;
    a.p0 = p0;
    a.sum_r2 = ctx->d_sum_r2;
    a.info = ctx->d_info;
    a.pred = pred;
    a.queue = ctx->d_queue;
    a.rej = reject ? ctx->d_reject : nullptr;
    a.stiff_list = sl.particles;
    a.stiff_count = sl.count;
    a.stiff_cap = sl.cap;
    a.solo_cap = sl.solo_cap;
    a.patience = patience;
    a.order = order;
    // a cost-ordered Metropolis sweep (not a probe's uploaded order): the first position of the last class = the in-support proposals
    a.n_ordered = cost_ordered ? cost_n_ordered(ctx) : nullptr;
    a.sorted = static_cast<const SortedProposal *>(ctx->d_sorted);
    a.ctl = ctl;
#ifdef SMC_DEBUG_PATIENCE_ENV   // A/B builds only (tools/ab_build.sh): in-phase patience of every sweep from the environment
    if (const char *e = getenv("SMC_DEBUG_PATIENCE")) a.patience = atoi(e);
#endif
    if (!queue_cleared) (void)hipMemsetAsync(ctx->d_queue, 0, sizeof(unsigned long long), ctx->stream);
    const size_t lds = solve_lds_bytes(mm.n_ex, mm.n_t);
    const bool exact = ctx->exact_pow != 0;
    const bool fast = use_fast_tail(ctx, n);
    void (*kern)(MMModel, SolveArgs) =
        pred ? (exact ? mm_solve_kernel<true, true, false> : fast ? mm_solve_kernel<true, false, true> : mm_solve_kernel<true, false, false>)
             : (exact ? mm_solve_kernel<false, true, false> : fast ? mm_solve_kernel<false, false, true> : mm_solve_kernel<false, false, false>);
    if (lds > 48 * 1024 && !ctx->solve_lds_raised) {
        // the largest data set (16 x 256) needs 66 + 30 KB of the CU's 160 KB: above the default dynamic limit.  The
        // attribute belongs to the (function, device) pair, so the flag lives in the context, not in the process.
        hipError_t e = hipSuccess;
        for (const void *f : {reinterpret_cast<const void *>(&mm_solve_kernel<true, true, false>), reinterpret_cast<const void *>(&mm_solve_kernel<true, false, false>),
                              reinterpret_cast<const void *>(&mm_solve_kernel<true, false, true>), reinterpret_cast<const void *>(&mm_solve_kernel<false, true, false>),
                              reinterpret_cast<const void *>(&mm_solve_kernel<false, false, false>), reinterpret_cast<const void *>(&mm_solve_kernel<false, false, true>)})
            if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e != hipSuccess) {
            smc_fail(ctx, "mm_solve_kernel: raising the dynamic LDS limit failed (hipFuncSetAttribute)");
            ctx->launch_failed = true;
            return;
        }
        ctx->solve_lds_raised = true;
    }
    const int64_t blocks = solve_grid_blocks(ctx, n);   // the waves go on until the queue is empty
    ScopedTimer tm(ctx, SMC_T_SOLVE);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kSolveBlock), lds, ctx->stream, mm, a);
}

void launch_mm_loglik(smc_ctx *ctx, const double *theta, int64_t stride, int64_t n, double *lk, double *pred) {
    if (n <= 0) return;
    const StiffList sl = next_stiff_list(ctx, n);
    if (sl.particles)
        hipLaunchKernelGGL(mm_stiff_scan_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, theta, stride, n, sl);
    // a probe's uploaded order (smc_debug_set_order) has n_local entries: a host-batch sweep of another size runs in index order
    const bool dbg_order = ctx->order_debug && n == ctx->n_local;
    launch_solve(ctx, theta, stride, n, nullptr, pred, sl, false, false, dbg_order ? ctx->order_debug_patience : 0,
                 dbg_order ? ctx->d_order : nullptr);
    MHParams mh{};
    hipLaunchKernelGGL((mm_finish_kernel<0>), dim3(finish_grid(n)), dim3(256), 0, ctx->stream, ctx->mm, mh,
                       theta, stride, n, ctx->d_sum_r2, ctx->d_info, nullptr, lk, nullptr, 0, nullptr, ctx->d_counters,
                       nullptr, nullptr);
}

void launch_mm_mh(smc_ctx *ctx, int64_t n, const MHParams &mh_in) {
    if (n <= 0) return;
    ParticleSet &F = ctx->set[SMC_SET_FILT];
    ParticleSet &P = ctx->set[SMC_SET_PRED];  // receives the proposals, as the reference's p_pred does (:220,228)
    const bool dbg = ctx->debug_capture != 0;
    // exact early rejection: off while the proposals' likelihoods are captured for inspection (they would be incomplete)
    const bool reject = ctx->early_reject != 0 && !dbg && mh_in.gamma > 0.0 && ctx->d_reject;
    MHParams mh = mh_in;
    if (reject) {
        mh.pending_sums = ctx->d_sum_r2;
        mh.pending_n_ex = ctx->mm.n_ex;
        mh.reject_out = ctx->d_reject;
        mh.reject_lk1 = F.lk;
    }
    mh.stiff = next_stiff_list(ctx, n);
    // in phase (solve_sched.h) when the previous Metropolis sweep of this context had fewer than 1 long item in 20 000 ...
    const bool homogeneous = ctx->last_sweep_items > 0 && ctx->last_sweep_long_items * 20000 < ctx->last_sweep_items;
    // ... and otherwise, for a sweep large enough to have something to sort, in cost order (above), also in phase
    const bool cost_order = ctx->cost_order != 0 && ctx->in_phase != 0 && !homogeneous && !ctx->order_debug && n >= 16384 &&
                            ctx->dim == 3 && ctx->d_order && ctx->d_bucket && ctx->d_order_hist && ctx->d_sorted;
    mh.cost_bucket = cost_order ? ctx->d_bucket : nullptr;
    mh.cost_table = ctx->d_order_hist;
    mh.done_sums = ctx->d_sum_r2;
    mh.done_info = ctx->d_info;
    mh.pending_n_ex = ctx->mm.n_ex;
    hipLaunchKernelGGL(mm_propose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->prior, mh,
                       F.theta, F.stride, n, P.theta, P.stride, ctx->d_p0);
    if (cost_order) {
        hipLaunchKernelGGL(cost_offsets_kernel, dim3(kCostBuckets), dim3(kCostBlocks), 0, ctx->stream, ctx->d_order_hist, mh.ctl);
        hipLaunchKernelGGL(cost_scatter_kernel, dim3(kCostBlocks), dim3(256), 0, ctx->stream, ctx->d_bucket, n, ctx->d_order_hist, P.theta, P.stride,
                           static_cast<SortedProposal *>(ctx->d_sorted), mh.ctl);
    }
    const int patience = (ctx->in_phase != 0 && (homogeneous || cost_order)) ? kInPhasePatience : (ctx->order_debug ? ctx->order_debug_patience : 0);
    launch_solve(ctx, P.theta, P.stride, n, ctx->d_p0, nullptr, mh.stiff, mh.zero_queue != nullptr, reject, patience,
                 ctx->order_debug ? ctx->d_order : nullptr, cost_order, mh.ctl);
    ctx->pending_sweep_items = n * ctx->mm.n_ex;
    ctx->moment_rows_n = mh.moment_rows ? (int)finish_grid(n) : 0;
    hipLaunchKernelGGL((mm_finish_kernel<1>), dim3(finish_grid(n)), dim3(256), 0, ctx->stream, ctx->mm, mh,
                       P.theta, P.stride, n, ctx->d_sum_r2, ctx->d_info, ctx->d_p0, F.lk, F.theta, F.stride, ctx->r_ac,
                       ctx->d_counters, dbg ? ctx->dbg_lk2 : nullptr, dbg ? ctx->dbg_r : nullptr);
}

int query_solve_blocks_per_cu(bool fast) {
    if (const char *e = getenv("SMC_SOLVE_BLOCKS_PER_CU")) {   // experiments: persistent blocks (4 waves each) per CU
        const int v = atoi(e);
        if (v >= 1) return v < 16 ? v : 16;     // beyond the hardware's wave slots the extra blocks only queue up
    }
    int nb = 0;
    const hipError_t e = fast ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, mm_solve_kernel<false, false, true>, kSolveBlock, solve_lds_bytes(6, 40))
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, mm_solve_kernel<false, false, false>, kSolveBlock, solve_lds_bytes(6, 40));
    if (e != hipSuccess || nb < 1)
        nb = 2;
    return nb;
}

}  // namespace smc
