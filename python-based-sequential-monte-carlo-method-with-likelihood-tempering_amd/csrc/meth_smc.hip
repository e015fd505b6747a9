// meth_smc.hip -- the methanation model inside the SMC loop (configs 4-5): likelihood sweep and Metropolis
// iteration on the resident particle sets.  Same three-stage shape as the Michaelis-Menten sweep
// (mm_kernels.hip): propose -> solve -> accept, where "solve" is K8, the DAE time integration of every
// (particle, experiment) pair (meth_dae_wave.h; PARITY UNPINNED against the reference's IDA), followed by
// my_loglike (methanation_set_likelihood.py:280-300) per particle.
//   sim_particle / cal_parallel_new   methanation_functions.py:44-92
//   MH iteration                       SMC_methanation_main.py:295-391 (the taken branch: normal_pred False)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "meth_dae_wave.h"
#include "meth_dae_elem.h"
#include "meth_dae_split.h"
#include "philox.h"
#include "prior.h"
#include "smc_internal.h"

namespace smc {

using namespace meth;

constexpr int kStatusUnsolved = -1;    // what launch_solves fills the status array with before a sweep
constexpr int kStatusCancelled = -2;   // not solved: the proposal was already certain to be rejected (meth_certainly_rejected)

// my_loglike (methanation_set_likelihood.py:280-300) in two pieces that BOTH the likelihood kernel and the early-rejection
// bound go through, so that the bound is an upper bound of the value the likelihood kernel forms in floating point too:
//   acc_i  = sum over the experiments e, in order, of (flow[e][i] - obs[i][e])^2         (:289-294, one species i)
//   logL   = sum over the species i, in order, of  -0.5/sigma^2 * acc_i - n_data*log(sigma)   (:295-298)
// Every operation is monotone (round-to-nearest preserves order), so leaving an experiment out - a zero in place of its
// non-negative square - can only raise the result.
__device__ __forceinline__ double meth_acc_step(double acc, double d) { return acc + d * d; }
__device__ __forceinline__ double meth_loglike_step(double total, double c, double acc, double l) { return total + (c * acc - l); }

// Which experiments tell most about a proposal?  Per experiment, over the solved items of a sweep, the mean of
// log(1 + sum over the species of (flow - obs)^2) - robust against the 1e8 of a failed solve's sentinel flows.  The NEXT
// Metropolis sweep solves the experiments in descending order of this misfit (launch_solves): a proposal that is going to be
// rejected then accumulates its damning residuals first and the early-rejection bound cancels its remaining solves sooner.
// Any order gives the same results (the likelihood is always summed in index order); the order only decides how soon.
__global__ void __launch_bounds__(256)
meth_experiment_stats_kernel(const double *__restrict__ flows, const int *__restrict__ status, const uint8_t *__restrict__ p0mask,
                             const double *__restrict__ obs, int64_t n, int n_data, double *__restrict__ stat /* 2 x n_data */,
                             const MHControl *__restrict__ ctl) {
    extern __shared__ double s_stat[];
    if (ctl && ctl->stop) return;       // batch of iterations under device control: the loop has ended
    for (int i = threadIdx.x; i < 2 * n_data; i += blockDim.x) s_stat[i] = 0.0;
    __syncthreads();
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        if (p0mask && p0mask[p] == 0) continue;
        for (int e = 0; e < n_data; ++e) {
            if (status[p * n_data + e] < 0) continue;        // not solved (cancelled)
            double t = 0.0;
            for (int i = 0; i < 5; ++i) {
                const double d = flows[(p * n_data + e) * 5 + i] - obs[i * n_data + e];
                t += d * d;
            }
            atomicAdd(&s_stat[e], log1p(t));
            atomicAdd(&s_stat[n_data + e], 1.0);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * n_data; i += blockDim.x)
        if (s_stat[i] != 0.0) atomicAdd(&stat[i], s_stat[i]);
}

// The order in which the NEXT sweep solves the experiments, on the device (rounds 2-4 read the statistics back and sorted on the
// host: one synchronisation per sweep, which a batch of iterations under device control cannot afford): experiment e goes to
// position #{e' : mean[e'] > mean[e], or equal and e' < e} - the stable descending sort of the means.  One block; then the
// statistics are cleared for the sweep that is about to start.  Experiments without a solved item have mean 0.
__global__ void __launch_bounds__(kWave)
meth_order_kernel(double *__restrict__ stat /* 2 x n_data */, int n_data, int *__restrict__ order, const MHControl *__restrict__ ctl) {
    __shared__ double s_mean[kWave];
    if (ctl && ctl->stop) return;
    const int e = threadIdx.x;
    double mean = 0.0;
    if (e < n_data) {
        const double cnt = stat[n_data + e];
        mean = cnt > 0 ? stat[e] / cnt : 0.0;
        s_mean[e] = mean;
    }
    __syncthreads();
    if (e < n_data) {
        int pos = 0;
        for (int q = 0; q < n_data; ++q) pos += (s_mean[q] > mean || (s_mean[q] == mean && q < e)) ? 1 : 0;
        order[pos] = e;
        stat[e] = 0.0;
        stat[n_data + e] = 0.0;
    }
}

// the live proposals of an MH sweep, compacted (order irrelevant); queue[1] counts them (ctl: none once the loop has ended - the
// solve kernel then finds an empty list and leaves)
__global__ void meth_livelist_kernel(const uint8_t *__restrict__ p0mask, int64_t n, int64_t *__restrict__ list,
                                     unsigned long long *__restrict__ queue, const MHControl *__restrict__ ctl) {
    if (ctl && ctl->stop) return;
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n || p0mask[p] == 0) return;   // masked proposal: lk2 == lk1, nothing to solve
    list[atomicAdd(&queue[1], 1ULL)] = p;   // one atomic per live lane: no cross-lane read follows, correct with or without the atomic optimiser
}

// EXACT early rejection for the methanation sweeps (the Michaelis-Menten counterpart is
// mm_certainly_rejected in mm_kernels.hip).  The accept test of the sweep, exp((lk2 - lk1) * gamma) [* p0_2/p0_1] >= rr
// (SMC_methanation_main.py:376-383), has lk1 and rr fixed before the proposal is solved, and lk2 = my_loglike only falls
// with every experiment that is added to it: each contributes five non-negative squares times -0.5/sigma^2 (a failed
// solve's -10000 flows - :244-249 - are just very large squares).  So the likelihood formed from the experiments of the
// particle that HAVE finished is an upper bound of lk2, and if even that bound fails the test the proposal is rejected
// whatever its remaining 357-state DAE solves would give: they are not started.  Unlike the Michaelis-Menten path a solve
// contributes nothing before it ends (the flows are read off the final state), so the saving comes from the ORDER of the
// work: the sweep runs experiment-major - experiment 0 of every live proposal, then experiment 1, ... - and by the time
// the later experiments of a proposal come up its earlier ones have long finished.  Nothing observable changes (p_filt,
// lk1, accept flags and counts are those of the full computation); the solve counters are smaller.
// Called by the whole wave before a solve: lane e looks at experiment e of the particle; the sums run in the likelihood
// kernel's order on every lane (v_readlane broadcast), so the result is wave-uniform.  n_data <= 64.
// Not inlined on purpose: the caller is the 512-VGPR integrator kernel, and folding this code into it made its register
// allocation spill twice as much into scratch; as a function with scalar arguments it gets its own allocation and costs
// one call per solve.
__device__ __attribute__((noinline)) bool meth_certainly_rejected(const RejectArgs *rp, const double *obs, int n_data,
                                                                  double sigma, int64_t p, const double *flows,
                                                                  const int *status, int lane) {
    const RejectArgs &r = *rp;
    double dv[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    bool cancelled = false;
    if (lane < n_data) {
        const int64_t w = p * n_data + lane;
        // the solving wave publishes flows, then status with release semantics at agent scope (L2 is per XCD)
        const int st = __hip_atomic_load(status + w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        cancelled = (st == kStatusCancelled);
        if (st >= 0) {
#pragma unroll
            for (int f = 0; f < 5; ++f) {
                const double fl = __longlong_as_double((long long)__hip_atomic_load(
                    reinterpret_cast<const unsigned long long *>(flows) + (w * 5 + f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                dv[f] = fl - obs[f * n_data + lane];
            }
        }
    }
    if (__ballot(cancelled) != 0ull) return true;     // a sibling has already established the rejection
    const double c = -(0.5 / (sigma * sigma)), l = n_data * log(sigma);
    double total = 0.0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        double acc = 0.0;
        for (int e = 0; e < n_data; ++e) acc = meth_acc_step(acc, lane_bcast(dv[i], e));   // 0 for an unfinished experiment
        total = meth_loglike_step(total, c, acc, l);
    }
    double rr;
    if (r.device_rng) {
        const u32x4 ru = philox_block(r.seed, (uint64_t)(r.global_offset + p), r.stream, SMC_PHILOX_BLOCK_UNIFORM);
        rr = u01_from(ru.x, ru.y);
    } else {
        rr = r.rr[p];
    }
    double pp = exp((total - r.lk1[p]) * r.gamma);
    if (r.prior_mode != SMC_PRIOR_MODE_MASK) pp = pp * r.pratio[p];
    return pp < rr * (1.0 - 1e-12);   // the margin covers a last-bit non-monotonicity of exp; any NaN: false
}

// K8 over (particle, experiment) pairs of the resident set; one wave per solve (meth_dae_elem.h), persistent waves
// that take solves from an atomic counter (a failed solve costs ~9 ordinary ones).  list == nullptr: all n * n_data
// pairs; otherwise the queue[1] pairs of the work list.  Loop control is scalar (wave_dequeue, wave_uniform in
// meth_dae_wave.h): the position and the count live in SGPRs, so the whole wave breaks together.  The counter only
// grows, so every wave leaves after its first position >= count; the hard bound of count + 1 trips is a second exit.
// Bookkeeping for the host (smc_meth_sweep_check): expected = count, completed += 1 per finished solve, wave_split.
__global__ void __launch_bounds__(64)
meth_particles_dae_kernel(MethModel m, const double *__restrict__ theta, int64_t stride, int64_t n,
                          const int64_t *__restrict__ live, double *flows, int *status, const RejectArgs *__restrict__ rej,
                          const int *__restrict__ order /* experiment solved at rank k, or nullptr = k */,
                          SweepCounters *__restrict__ counters, unsigned long long *__restrict__ queue) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const DViewE D{lds + kLdsD, lane};
    // EXPERIMENT-MAJOR order: position q = k * n_live + j is the experiment of rank k of the j-th live particle, so that the experiments
    // of one proposal come up one after the other, n_live positions apart, and the later ones can be cancelled on the
    // strength of the earlier ones (meth_certainly_rejected).  Which experiment is solved first cannot change a result.
    const int64_t n_live = live ? (int64_t)queue[1] : n;
    const int64_t count = n_live * m.n_data;
    if (blockIdx.x == 0 && lane == 0) counters->expected_solves = (unsigned long long)count;
    unsigned split = 0;
    for (int64_t it = 0; it <= count; ++it) {
        const int64_t pos = wave_dequeue(&queue[0], lane, split);
        if (pos >= count) break;
        const int e_rank = (int)(pos / n_live);
        const int64_t j = pos - (int64_t)e_rank * n_live;
        const int e = order ? order[e_rank] : e_rank;           // most informative experiments first (meth_experiment_stats_kernel)
        const int64_t particle = live ? live[j] : j;
        const int64_t w = particle * m.n_data + e;
        if (rej) {
            // wave-uniform by construction (every lane runs the same sums on broadcast operands); the v_readfirstlane tells
            // the compiler so - the atomic loads inside are a divergence source for its analysis
            double sigma = m.sigma_fixed;
            if (m.est_sigma) {
                sigma = m.base[8];
                for (int kq = 0; kq < m.dim; ++kq)
                    if (m.est_pos[kq] == 8) sigma = theta[kq * stride + particle];
            }
            if (__builtin_amdgcn_readfirstlane((int)meth_certainly_rejected(rej, m.obs, m.n_data, sigma, particle, flows, status, lane))) {
                // no `if (lane == 0)` in front of the `continue`: the join of such a branch would be the loop's latch, and the
                // compiler's uniformity analysis then calls the whole dequeue loop - DPP, permlane and all - a cycle with a
                // divergent exit (tests/test_k8_uniform_control.py).  Every lane stores the same word; lane 0 counts.
                __hip_atomic_store(status + w, kStatusCancelled, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                atomicAdd(&counters->cancelled_solves, lane == 0 ? 1ULL : 0ULL);
                continue;
            }
        }
        double p[18];
        for (int q = 0; q < 10; ++q) p[q] = m.cond[e * 10 + q];
        for (int jj = 0; jj < 8; ++jj) {  // p_pred_bases[:, est_position] = particle (methanation_functions.py:80)
            double v = m.base[jj];
            for (int kq = 0; kq < m.dim; ++kq)
                if (m.est_pos[kq] == jj) v = theta[kq * stride + particle];
            p[10 + jj] = v;
        }
        if (lane < kNX)
            for (int f = 0; f < 7; ++f) {
                D(0, f) = m.guess[(int64_t)e * kNS + f * kNX + lane];
                for (int kk = 1; kk < 8; ++kk) D(kk, f) = 0.0;
            }
        DaeStats st;
        dae_elem_integrate(lds, lane, p, m.tf, m.rtol, m.atol, m.h0, kDaeMaxAttempts, st);
        if (lane == kNX - 1) {
            const double u = D(0, 6), T = D(0, 5);
            const double P_total = (p[0] + p[1] + p[2] + p[3] + p[4]) * k::R * p[5];
            for (int f = 0; f < 5; ++f) {
                const double cc = D(0, f);
                const double fl = (st.status == 0)
                                      ? cc * m.S * u * 60 * k::R * T / (P_total) * 1e6 * (P_total) / m.P_stp * 298 / T
                                      : -10000.0;   // methanation_set_likelihood.py:244-249
                // visible to the waves of other XCDs while the kernel runs: agent-scope stores, the status last (release)
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(flows) + (w * 5 + f),
                                   (unsigned long long)__double_as_longlong(fl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __hip_atomic_store(status + w, st.status, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            atomicAdd(&counters->rk_attempts, (unsigned long long)st.steps);
            atomicAdd(&counters->newton_iters, (unsigned long long)st.newton_iters);
            atomicAdd(&counters->factorisations, (unsigned long long)st.nlu);
            atomicAdd(&counters->completed_solves, 1ULL);
            if (st.status != 0) atomicAdd(&counters->failed_solves, 1ULL);
        }
        // the join of the lane branch above must be a block of its own: merged with the loop's latch (where `continue`,
        // `break` and the end of the body meet) it would make the loop's exit look divergent to the compiler
        __builtin_amdgcn_wave_barrier();
    }
    if (split && lane == 0) atomicAdd(&counters->wave_split, 1ULL);
}

// The same sweep with TWO waves per solve (meth_dae_split.h): wave 0 is the kernel above with the downward chain only, wave 1
// serves the upward chain on wave 0's commands (it never looks at the queue, the rejection bound or the results).
__global__ void __launch_bounds__(kSplitThreads, 2)
meth_particles_dae_split_kernel(MethModel m, const double *__restrict__ theta, int64_t stride, int64_t n,
                                const int64_t *__restrict__ live, double *flows, int *status, const RejectArgs *__restrict__ rej,
                                const int *__restrict__ order, SweepCounters *__restrict__ counters,
                                unsigned long long *__restrict__ queue, int role_policy) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    if (split_role(lds, __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), role_policy)) {
        dae_split_server(lds, lane);
        return;
    }
    const DViewE D{lds + kLdsD, lane};
    const int64_t n_live = live ? (int64_t)queue[1] : n;
    const int64_t count = n_live * m.n_data;
    if (blockIdx.x == 0 && lane == 0) counters->expected_solves = (unsigned long long)count;
    unsigned split = 0;
    for (int64_t it = 0; it <= count; ++it) {
        const int64_t pos = wave_dequeue(&queue[0], lane, split);
        if (pos >= count) break;
        const int e_rank = (int)(pos / n_live);
        const int64_t j = pos - (int64_t)e_rank * n_live;
        const int e = order ? order[e_rank] : e_rank;
        const int64_t particle = live ? live[j] : j;
        const int64_t w = particle * m.n_data + e;
        if (rej) {
            double sigma = m.sigma_fixed;
            if (m.est_sigma) {
                sigma = m.base[8];
                for (int kq = 0; kq < m.dim; ++kq)
                    if (m.est_pos[kq] == 8) sigma = theta[kq * stride + particle];
            }
            if (__builtin_amdgcn_readfirstlane((int)meth_certainly_rejected(rej, m.obs, m.n_data, sigma, particle, flows, status, lane))) {
                __hip_atomic_store(status + w, kStatusCancelled, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                atomicAdd(&counters->cancelled_solves, lane == 0 ? 1ULL : 0ULL);
                continue;
            }
        }
        double p[18];
        for (int q = 0; q < 10; ++q) p[q] = wave_uniform(m.cond[e * 10 + q]);
        for (int jj = 0; jj < 8; ++jj) {  // p_pred_bases[:, est_position] = particle (methanation_functions.py:80)
            double v = m.base[jj];
            for (int kq = 0; kq < m.dim; ++kq)
                if (m.est_pos[kq] == jj) v = theta[kq * stride + particle];
            p[10 + jj] = wave_uniform(v);
        }
        if (lane < kNX)
            for (int f = 0; f < 7; ++f) {
                D(0, f) = m.guess[(int64_t)e * kNS + f * kNX + lane];
                for (int kk = 1; kk < 8; ++kk) D(kk, f) = 0.0;
            }
        DaeStats st;
        dae_split_integrate(lds, lane, p, m.tf, m.rtol, m.atol, m.h0, kDaeMaxAttempts, st);
        if (lane == kNX - 1) {
            const double u = D(0, 6), T = D(0, 5);
            const double P_total = (p[0] + p[1] + p[2] + p[3] + p[4]) * k::R * p[5];
            for (int f = 0; f < 5; ++f) {
                const double cc = D(0, f);
                const double fl = (st.status == 0)
                                      ? cc * m.S * u * 60 * k::R * T / (P_total) * 1e6 * (P_total) / m.P_stp * 298 / T
                                      : -10000.0;   // methanation_set_likelihood.py:244-249
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(flows) + (w * 5 + f),
                                   (unsigned long long)__double_as_longlong(fl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __hip_atomic_store(status + w, st.status, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            atomicAdd(&counters->rk_attempts, (unsigned long long)st.steps);
            atomicAdd(&counters->newton_iters, (unsigned long long)st.newton_iters);
            atomicAdd(&counters->factorisations, (unsigned long long)st.nlu);
            atomicAdd(&counters->completed_solves, 1ULL);
            if (st.status != 0) atomicAdd(&counters->failed_solves, 1ULL);
        }
        __builtin_amdgcn_wave_barrier();
    }
    split_command(lds, kCmdQuit);
    if (split && lane == 0) atomicAdd(&counters->wave_split, 1ULL);
}

// my_loglike per particle from its 5 x n_data flows (:280-300); sigma = the particle's last estimated parameter
// when est_sigma (methanation_functions.py:50-53).  The status array is poisoned (kStatusUnsolved) before every sweep:
// a live particle with an experiment nobody solved gets NaN and is counted, so stale flows can never pass as a result.
__global__ void meth_particle_loglike_kernel(MethModel m, const double *__restrict__ theta, int64_t stride, int64_t n,
                                             const double *__restrict__ flows, const int *__restrict__ status,
                                             uint8_t *__restrict__ p0mask, SweepCounters *__restrict__ counters,
                                             double *__restrict__ lk) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (p0mask && p0mask[p] == 0) {   // masked proposal: nothing was solved, the accept kernel keeps lk1
        lk[p] = __longlong_as_double(0x7ff8000000000000LL);
        return;
    }
    int unsolved = 0, cancelled = 0;
    for (int e = 0; e < m.n_data; ++e) {
        const int st = status[p * m.n_data + e];
        unsolved += (st == kStatusUnsolved);
        cancelled += (st == kStatusCancelled);
    }
    if (cancelled) {   // rejected for certain before all of its experiments were solved: its logL is never formed (and not
        lk[p] = __longlong_as_double(0x7ff8000000000000LL);   // needed: the accept kernel sees the flag and keeps p_filt, lk1)
        if (p0mask) p0mask[p] = 2;
        if (unsolved) atomicAdd(&counters->unsolved_items, (unsigned long long)unsolved);   // cannot happen: every item is dequeued
        return;
    }
    double sigma = m.sigma_fixed;
    if (m.est_sigma) {
        sigma = m.base[8];
        for (int kq = 0; kq < m.dim; ++kq)
            if (m.est_pos[kq] == 8) sigma = theta[kq * stride + p];
    }
    const double c = -(0.5 / (sigma * sigma)), l = m.n_data * log(sigma);
    double total = 0.0;
    for (int i = 0; i < 5; ++i) {
        double acc = 0.0;
        for (int e = 0; e < m.n_data; ++e)
            acc = meth_acc_step(acc, flows[(p * m.n_data + e) * 5 + i] - m.obs[i * m.n_data + e]);
        total = meth_loglike_step(total, c, acc, l);
    }
    if (unsolved) {
        atomicAdd(&counters->unsolved_items, (unsigned long long)unsolved);
        total = __longlong_as_double(0x7ff8000000000000LL);
    }
    lk[p] = total;
}

// proposal + support mask for any dimension d <= SMC_MAX_DIM (SMC_methanation_main.py:312-336 /
// Micmem_SMC_main.py:220-228)
__global__ void __launch_bounds__(256)
generic_propose_kernel(Prior prior, MHParams mh, const double *__restrict__ filt, int64_t stride, int64_t n, int d,
                       double *__restrict__ prop, int64_t pstride, uint8_t *__restrict__ p0_out) {
    if (mh.ctl && mh.ctl->stop) return;      // batch of iterations under device control: the loop has ended
    if (blockIdx.x == 0 && mh.zero_counters && threadIdx.x < sizeof(SweepCounters) / 8)   // (instead of a memset: that could not be skipped)
        reinterpret_cast<unsigned long long *>(mh.zero_counters)[threadIdx.x] = 0ULL;
    if (blockIdx.x == 0 && threadIdx.x == 0 && mh.reject_out) {   // what the early-rejection bound of this sweep's solves reads
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
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (mh.pending_sums)   // user model with early rejection: none of this sweep's items has finished yet (NaN = pending)
        for (int k = 0; k < mh.pending_n_ex; ++k) mh.pending_sums[(int64_t)k * n + p] = __longlong_as_double(0x7ff8000000000000LL);
    double z[SMC_MAX_DIM], g[SMC_MAX_DIM];
    if (mh.device_rng) {
        const uint64_t gi = (uint64_t)(mh.global_offset + p);
        for (int b = 0; b < (SMC_MAX_DIM + 1) / 2; ++b) {
            if (2 * b >= d) break;
            const u32x4 r = philox_block(mh.seed, gi, mh.stream, (uint32_t)b);
            const double u1 = 1.0 - u01_from(r.x, r.y), u2 = u01_from(r.z, r.w);
            const double rad = sqrt(-2.0 * log(u1));
            double sn, cs;
            sincos(6.283185307179586 * u2, &sn, &cs);
            g[2 * b] = rad * cs;
            if (2 * b + 1 < SMC_MAX_DIM) g[2 * b + 1] = rad * sn;
        }
        const double *T = mh.transform_dev ? mh.transform_dev : mh.transform;
        for (int c = 0; c < d; ++c) {
            double s = 0.0;
            for (int kq = 0; kq < d; ++kq) s += g[kq] * T[kq * d + c];
            z[c] = s;
        }
    } else {
        for (int c = 0; c < d; ++c) z[c] = mh.noise[(int64_t)c * n + p];
    }
    double pdf = 1.0, cand[SMC_MAX_DIM], cur[SMC_MAX_DIM];
    const double ratio = mh.ctl ? mh.ctl->ratio : mh.ratio;     // mhstep_ratio: the control kernel's under device control
    for (int c = 0; c < d; ++c) {
        cur[c] = filt[c * stride + p];
        cand[c] = __dadd_rn(cur[c], __dmul_rn(z[c], ratio));
        const double q = prior_pdf(prior.kind[c], prior.a[c], prior.b[c], cand[c]);
        pdf = (c == 0) ? q : pdf * q;
    }
    if (mh.prior_mode != SMC_PRIOR_MODE_MASK) {   // p0_2 / p0_1 (SMC_methanation_main.py:323-324, 343)
        double cur_pdf = 1.0;
        for (int c = 0; c < d; ++c) {
            const double q = prior_pdf(prior.kind[c], prior.a[c], prior.b[c], cur[c]);
            cur_pdf = (c == 0) ? q : cur_pdf * q;
        }
        mh.pratio[p] = pdf / cur_pdf;
    }
    const double p0 = (pdf > 0.0 || mh.prior_mode == SMC_PRIOR_MODE_RATIO) ? 1.0 : 0.0, q0 = 1.0 - p0;
    for (int c = 0; c < d; ++c) prop[c * pstride + p] = __dadd_rn(__dmul_rn(cand[c], p0), __dmul_rn(cur[c], q0));
    p0_out[p] = (uint8_t)(p0 != 0.0);
}

// accept / select for any dimension, from a precomputed lk2 array
__global__ void __launch_bounds__(256)
generic_accept_kernel(MHParams mh, const double *__restrict__ prop, int64_t pstride, int64_t n, int d,
                      const double *__restrict__ lk2_arr, const uint8_t *__restrict__ p0_in, double *lk_io, double *filt,
                      int64_t fstride, uint8_t *__restrict__ r_ac, SweepCounters *__restrict__ counters,
                      double *__restrict__ dbg_lk2, uint8_t *__restrict__ dbg_r) {
    __shared__ unsigned long long s_cnt[4][2];
    if (mh.ctl && mh.ctl->stop) return;   // after the loop's `break`: p_filt, lk1, r_ac and the counters stay as they are
    unsigned long long acc_now = 0, acc_ever = 0;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const bool masked = p0_in[p] == 0;
        const bool cancelled = p0_in[p] == 2;   // methanation: rejected for certain before all experiments were solved
        const double lk1 = lk_io[p];
        const double lk2 = (masked || cancelled) ? lk1 : lk2_arr[p];
        const double p0 = masked ? 0.0 : 1.0;
        double rr;
        if (mh.device_rng) {
            const u32x4 ru = philox_block(mh.seed, (uint64_t)(mh.global_offset + p), mh.stream, SMC_PHILOX_BLOCK_UNIFORM);
            rr = u01_from(ru.x, ru.y);
        } else {
            rr = mh.rr[p];
        }
        double pp = exp((lk2 - lk1) * mh.gamma);
        if (mh.prior_mode != SMC_PRIOR_MODE_MASK) pp = pp * mh.pratio[p];
        if (mh.prior_mode != SMC_PRIOR_MODE_RATIO) pp = pp * p0;
        const double r = (!cancelled && pp >= rr) ? 1.0 : 0.0, nr = 1.0 - r;
        for (int c = 0; c < d; ++c) {
            const double th = prop[c * pstride + p], f = filt[c * fstride + p];
            filt[c * fstride + p] = __dadd_rn(__dmul_rn(th, r), __dmul_rn(f, nr));
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
    for (int off = 32; off > 0; off >>= 1) {
        acc_now += __shfl_down(acc_now, off);
        acc_ever += __shfl_down(acc_ever, off);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_cnt[w][0] = acc_now;
        s_cnt[w][1] = acc_ever;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const unsigned long long v = s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];
        if (v) atomicAdd(threadIdx.x == 0 ? &counters->accepted_now : &counters->accepted_ever, v);
    }
}

static void launch_solves(smc_ctx *ctx, const double *theta, int64_t stride, int64_t n, const uint8_t *p0mask, bool reject,
                          const MHControl *ctl = nullptr) {
    const MethModel &m = ctx->meth;
    int64_t nwaves = (int64_t)ctx->cu_count * 4;
    if (nwaves > n * m.n_data) nwaves = n * m.n_data;
    if (nwaves < 1) nwaves = 1;
    ScopedTimer tm(ctx, SMC_T_SOLVE);
    hipError_t e = hipMemsetAsync(ctx->d_queue, 0, 2 * sizeof(unsigned long long), ctx->stream);
    if (e == hipSuccess)   // 0xff bytes = kStatusUnsolved in every int
        e = hipMemsetAsync(ctx->d_mstatus, 0xff, (size_t)n * m.n_data * sizeof(int), ctx->stream);
    if (e != hipSuccess) {
        ctx->err = std::string("methanation sweep: clearing the work queue / status array failed: ") + hipGetErrorString(e);
        ctx->launch_failed = true;
        return;
    }
    const int64_t *live = nullptr;
    if (p0mask) {
        hipLaunchKernelGGL(meth_livelist_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, p0mask, n,
                           ctx->d_mwork, ctx->d_queue, ctl);
        live = ctx->d_mwork;
    }
    const int *order = nullptr;
    if (reject && ctx->stiff_first && ctx->d_morder && m.n_data <= kWave) {
        // the misfit statistics the previous Metropolis sweep left behind -> this sweep's order (all zero before the first
        // sweep: the index order); meth_order_kernel clears them for this sweep's own statistics
        hipLaunchKernelGGL(meth_order_kernel, dim3(1), dim3(kWave), 0, ctx->stream, ctx->d_mstat, m.n_data, ctx->d_morder, ctl);
        order = ctx->d_morder;
    }
    if (meth_split_enabled())   // two waves per solve: the same number of workgroups (four solves per CU), twice the waves
        hipLaunchKernelGGL(meth_particles_dae_split_kernel, dim3((unsigned)nwaves), dim3(meth::kSplitThreads),
                           meth::kLdsSplitDoubles * sizeof(double), ctx->stream, m, theta, stride, n, live, ctx->d_mflows,
                           ctx->d_mstatus, reject ? ctx->d_reject : nullptr, order, ctx->d_counters, ctx->d_queue, meth_split_role_policy());
    else
        hipLaunchKernelGGL(meth_particles_dae_kernel, dim3((unsigned)nwaves), dim3(64), kLdsDoubles * sizeof(double), ctx->stream,
                           m, theta, stride, n, live, ctx->d_mflows, ctx->d_mstatus, reject ? ctx->d_reject : nullptr, order,
                           ctx->d_counters, ctx->d_queue);
}

void launch_meth_loglik(smc_ctx *ctx, const double *theta, int64_t stride, int64_t n, double *lk) {
    if (n <= 0) return;
    launch_solves(ctx, theta, stride, n, nullptr, false);
    if (ctx->launch_failed) return;
    hipLaunchKernelGGL(meth_particle_loglike_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->meth,
                       theta, stride, n, ctx->d_mflows, ctx->d_mstatus, nullptr, ctx->d_counters, lk);
}

void launch_generic_propose(smc_ctx *ctx, int64_t n, const MHParams &mh) {
    ParticleSet &F = ctx->set[SMC_SET_FILT];
    ParticleSet &P = ctx->set[SMC_SET_PRED];
    hipLaunchKernelGGL(generic_propose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->prior, mh,
                       F.theta, F.stride, n, ctx->dim, P.theta, P.stride, ctx->d_p0);
}

void launch_generic_accept(smc_ctx *ctx, int64_t n, const MHParams &mh, const double *lk2) {
    ParticleSet &F = ctx->set[SMC_SET_FILT];
    ParticleSet &P = ctx->set[SMC_SET_PRED];
    const bool dbg = ctx->debug_capture != 0;
    const int64_t g = (n + 255) / 256;
    hipLaunchKernelGGL(generic_accept_kernel, dim3((unsigned)(g < 1024 ? g : 1024)), dim3(256), 0, ctx->stream, mh, P.theta,
                       P.stride, n, ctx->dim, lk2, ctx->d_p0, F.lk, F.theta, F.stride, ctx->r_ac, ctx->d_counters,
                       dbg ? ctx->dbg_lk2 : nullptr, dbg ? ctx->dbg_r : nullptr);
}

void launch_meth_mh(smc_ctx *ctx, int64_t n, const MHParams &mh_in) {
    if (n <= 0) return;
    ParticleSet &P = ctx->set[SMC_SET_PRED];
    // exact early rejection: off while the proposals' likelihoods are captured for inspection (they would be incomplete),
    // and for data sets with more experiments than a wave has lanes (the bound looks at one experiment per lane)
    const bool reject = ctx->early_reject != 0 && ctx->debug_capture == 0 && mh_in.gamma > 0.0 && ctx->d_reject &&
                        ctx->meth.n_data <= kWave;
    MHParams mh = mh_in;
    if (reject) {
        mh.reject_out = ctx->d_reject;
        mh.reject_lk1 = ctx->set[SMC_SET_FILT].lk;
    }
    launch_generic_propose(ctx, n, mh);
    launch_solves(ctx, P.theta, P.stride, n, ctx->d_p0, reject, mh.ctl);
    if (ctx->launch_failed) return;
    if (reject && ctx->stiff_first && ctx->d_mstat && ctx->meth.n_data <= kWave) {   // misfit per experiment of THIS sweep's proposals: the next sweep's order
        const int64_t g = (n + 255) / 256;
        hipLaunchKernelGGL(meth_experiment_stats_kernel, dim3((unsigned)(g < 512 ? g : 512)), dim3(256),
                           2 * (size_t)ctx->meth.n_data * sizeof(double), ctx->stream, ctx->d_mflows, ctx->d_mstatus, ctx->d_p0,
                           ctx->meth.obs, n, ctx->meth.n_data, ctx->d_mstat, mh.ctl);
    }
    hipLaunchKernelGGL(meth_particle_loglike_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->meth,
                       P.theta, P.stride, n, ctx->d_mflows, ctx->d_mstatus, ctx->d_p0, ctx->d_counters, ctx->d_mlk2);
    launch_generic_accept(ctx, n, mh, ctx->d_mlk2);
}

}  // namespace smc
