// meth_kernels.hip -- kernels of the methanation rows: K7 (batched DAE residual, rate law, Gaussian
// log-likelihood from outlet flows; pinned against the reference's own functions) and K8 (the DAE time
// integration of my_model, meth_dae.h; parity unpinned - no IDA to compare with).  Context-free C-ABI entry
// points operating on host buffers.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/smc_hip.h"
#include "meth_dae.h"
#include "meth_dae_wave.h"
#include "meth_dae_elem.h"
#include "meth_dae_split.h"
#include "meth_model.h"

namespace smc {
namespace meth {

// one wave per state vector, lane = axial node (51 of 64 lanes active): neighbouring nodes are
// neighbouring lanes, field-major rows are read coalesced
__global__ void __launch_bounds__(64) residual_kernel(const double *__restrict__ X, const double *__restrict__ dX,
                                                      const double *__restrict__ params, int64_t n,
                                                      double *__restrict__ res) {
    const int64_t b = blockIdx.x;
    if (b >= n) return;
    __shared__ double sX[NSTATE], sD[NSTATE], sP[NPAR];
    for (int k = threadIdx.x; k < NSTATE; k += 64) {
        sX[k] = X[b * NSTATE + k];
        sD[k] = dX[b * NSTATE + k];
    }
    if (threadIdx.x < NPAR) sP[threadIdx.x] = params[b * NPAR + threadIdx.x];
    __syncthreads();
    const int i = threadIdx.x;
    if (i >= NX) return;
    double out[7];
    node_residual(i, (const double *)sX, (const double *)sD, sP, out);
    for (int f = 0; f < 7; ++f) res[b * NSTATE + f * NX + i] = out[f];
}

__global__ void rate_kernel(const double *__restrict__ in, const double *__restrict__ kin, int64_t n,
                            double *__restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double k[8];
    for (int q = 0; q < 8; ++q) k[q] = kin[j * 8 + q];
    out[j] = rCH4(in[j * 5], in[j * 5 + 1], in[j * 5 + 2], in[j * 5 + 3], in[j * 5 + 4], k);
}

// my_loglike (methanation_set_likelihood.py:280-300): per particle, 5 components x n_data flows
__global__ void loglike_kernel(const double *__restrict__ y, const double *__restrict__ data,
                               const double *__restrict__ sigma, int64_t n, int n_data, double *__restrict__ lk) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const double s = sigma[j];
    const double c = -(0.5 / (s * s)), l = n_data * log(s);
    double total = 0.0;
    for (int i = 0; i < 5; ++i) {
        double acc = 0.0;
        for (int k = 0; k < n_data; ++k) {
            const double d = y[(j * 5 + i) * n_data + k] - data[i * n_data + k];
            acc += d * d;
        }
        total += c * acc - l;
    }
    lk[j] = total;
}

// K8: one thread per (particle, experiment) solve; the per-solve workspace is interleaved across the
// `nslots` threads of the launch (element idx of slot s at ws[idx*nslots + s]) so that the 64 lanes of a wave,
// which execute the same statement on 64 independent solves, touch 64 consecutive doubles.
__global__ void __launch_bounds__(64)
dae_kernel(double *__restrict__ wsbuf, int64_t nslots, const double *__restrict__ p0_all,
           const double *__restrict__ y0_all, int64_t n_solves, double tf, double rtol, double atol, double h0,
           int max_attempts, double S, double P_stp, double *__restrict__ flows, double *__restrict__ y_final,
           int *__restrict__ status, unsigned long long *__restrict__ counters) {
    const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= nslots) return;
    const Ws ws{wsbuf + slot, nslots};
    for (int64_t sidx = slot; sidx < n_solves; sidx += nslots) {
        double p[18];
        for (int q = 0; q < 18; ++q) p[q] = p0_all[sidx * 18 + q];
        for (int x = 0; x < kNS; ++x) ws(OFF_D + x) = y0_all[sidx * kNS + x];
        for (int x = kNS; x < 8 * kNS; ++x) ws(OFF_D + x) = 0.0;
        DaeStats st;
        dae_integrate(ws, p, tf, rtol, atol, h0, max_attempts, st);
        double F[5];
        if (st.status == 0) {
            outlet_flows(ws, p, S, P_stp, F);
        } else {
            for (int f = 0; f < 5; ++f) F[f] = -10000.0;  // the reference's failure sentinel (:244-249)
        }
        for (int f = 0; f < 5; ++f) flows[sidx * 5 + f] = F[f];
        if (y_final)
            for (int x = 0; x < kNS; ++x) y_final[sidx * kNS + x] = ws(OFF_D + x);
        status[sidx] = st.status;
        atomicAdd(&counters[0], (unsigned long long)st.steps);
        atomicAdd(&counters[1], (unsigned long long)st.rejects);
        atomicAdd(&counters[2], (unsigned long long)st.newton_fail);
        atomicAdd(&counters[3], (unsigned long long)st.newton_iters);
    }
}

// K8 v2: one wave per solve, lane = axial node (meth_dae_wave.h).  Persistent waves walk the solve list.
__global__ void __launch_bounds__(64)
dae_wave_kernel(const double *__restrict__ p0_all, const double *__restrict__ y0_all, int64_t n_solves, double tf,
                double rtol, double atol, double h0, int max_attempts, double S, double P_stp,
                double *__restrict__ flows, double *__restrict__ y_final, int *__restrict__ status,
                unsigned long long *__restrict__ counters) {
    extern __shared__ double sD[];  // 8 x 7 x 64
    const int lane = threadIdx.x;
    const DView D{sD, lane};
    for (int64_t sidx = blockIdx.x; sidx < n_solves; sidx += gridDim.x) {
        double p[18];
        for (int q = 0; q < 18; ++q) p[q] = p0_all[sidx * 18 + q];
        if (lane < kNX)
            for (int f = 0; f < 7; ++f) {
                D(0, f) = y0_all[sidx * kNS + f * kNX + lane];
                for (int kk = 1; kk < 8; ++kk) D(kk, f) = 0.0;
            }
        DaeStats st;
        dae_wave_integrate(sD, lane, p, tf, rtol, atol, h0, max_attempts, st);
        if (lane == kNX - 1) {
            const double u = D(0, 6), T = D(0, 5);
            const double P_total = (p[0] + p[1] + p[2] + p[3] + p[4]) * k::R * p[5];
            for (int f = 0; f < 5; ++f) {
                const double cc = D(0, f);
                flows[sidx * 5 + f] = (st.status == 0)
                                          ? cc * S * u * 60 * k::R * T / (P_total) * 1e6 * (P_total) / P_stp * 298 / T
                                          : -10000.0;
            }
            status[sidx] = st.status;
            atomicAdd(&counters[0], (unsigned long long)st.steps);
            atomicAdd(&counters[1], (unsigned long long)st.rejects);
            atomicAdd(&counters[2], (unsigned long long)st.newton_fail);
            atomicAdd(&counters[3], (unsigned long long)st.newton_iters);
            atomicAdd(&counters[4], (unsigned long long)st.nlu);
#ifdef SMC_METH_PROFILE
            for (int q = 0; q < 12; ++q) atomicAdd(&counters[8 + q], (unsigned long long)st.prof[q]);
#endif
        }
        if (y_final && lane < kNX)
            for (int f = 0; f < 7; ++f) y_final[sidx * kNS + f * kNX + lane] = D(0, f);
    }
}

// K8 v3: one wave per solve, scans in element layout (meth_dae_elem.h); solves are handed out by an atomic counter
// (a failed solve runs its whole attempt budget, ~9x an ordinary one, so a static split leaves waves idle).
__global__ void __launch_bounds__(64)
dae_elem_kernel(const double *__restrict__ p0_all, const double *__restrict__ y0_all, int64_t n_solves, double tf,
                double rtol, double atol, double h0, int max_attempts, double S, double P_stp,
                double *__restrict__ flows, double *__restrict__ y_final, int *__restrict__ status,
                unsigned long long *__restrict__ counters) {
    extern __shared__ double lds[];  // kLdsDoubles
    const int lane = threadIdx.x;
    const DViewE D{lds + kLdsD, lane};
    unsigned split = 0;
    for (int64_t it = 0; it <= n_solves; ++it) {   // scalar loop control (meth_dae_wave.h: wave_dequeue)
        const int64_t sidx = wave_dequeue(&counters[5], lane, split);
        if (sidx >= n_solves) break;
        double p[18];
        for (int q = 0; q < 18; ++q) p[q] = p0_all[sidx * 18 + q];
        if (lane < kNX)
            for (int f = 0; f < 7; ++f) {
                D(0, f) = y0_all[sidx * kNS + f * kNX + lane];
                for (int kk = 1; kk < 8; ++kk) D(kk, f) = 0.0;
            }
        DaeStats st;
        dae_elem_integrate(lds, lane, p, tf, rtol, atol, h0, max_attempts, st);
        if (lane == kNX - 1) {
            const double u = D(0, 6), T = D(0, 5);
            const double P_total = (p[0] + p[1] + p[2] + p[3] + p[4]) * k::R * p[5];
            for (int f = 0; f < 5; ++f) {
                const double cc = D(0, f);
                flows[sidx * 5 + f] = (st.status == 0)
                                          ? cc * S * u * 60 * k::R * T / (P_total) * 1e6 * (P_total) / P_stp * 298 / T
                                          : -10000.0;
            }
            status[sidx] = st.status;
            atomicAdd(&counters[0], (unsigned long long)st.steps);
            atomicAdd(&counters[1], (unsigned long long)st.rejects);
            atomicAdd(&counters[2], (unsigned long long)st.newton_fail);
            atomicAdd(&counters[3], (unsigned long long)st.newton_iters);
            atomicAdd(&counters[4], (unsigned long long)st.nlu);
#ifdef SMC_METH_PROFILE
            for (int q = 0; q < 12; ++q) atomicAdd(&counters[8 + q], (unsigned long long)st.prof[q]);
#endif
        }
        if (y_final && lane < kNX)
            for (int f = 0; f < 7; ++f) y_final[sidx * kNS + f * kNX + lane] = D(0, f);
        if (lane == 0) atomicAdd(&counters[6], 1ULL);   // finished solves
    }
    if (split && lane == 0) atomicAdd(&counters[7], 1ULL);
}


// K8 v4: one solve per workgroup of two waves (meth_dae_split.h): wave 0 runs the integrator and the downward chain, wave 1 serves
// the upward chain.  Wave 0 takes the solves from the atomic counter; wave 1 only ever sees commands.
__global__ void __launch_bounds__(kSplitThreads, 2)
dae_split_kernel(const double *__restrict__ p0_all, const double *__restrict__ y0_all, int64_t n_solves, double tf,
                 double rtol, double atol, double h0, int max_attempts, double S, double P_stp,
                 double *__restrict__ flows, double *__restrict__ y_final, int *__restrict__ status,
                 unsigned long long *__restrict__ counters, int role_policy) {
    extern __shared__ double lds[];  // kLdsSplitDoubles
    const int lane = threadIdx.x & 63;
    if (split_role(lds, __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), role_policy)) {
        dae_split_server(lds, lane);
        return;
    }
    const DViewE D{lds + kLdsD, lane};
    unsigned split = 0;
    for (int64_t it = 0; it <= n_solves; ++it) {
        const int64_t sidx = wave_dequeue(&counters[5], lane, split);
        if (sidx >= n_solves) break;
        double p[18];
        for (int q = 0; q < 18; ++q) {
            p[q] = wave_uniform(p0_all[sidx * 18 + q]);
        }
        if (lane < kNX)
            for (int f = 0; f < 7; ++f) {
                D(0, f) = y0_all[sidx * kNS + f * kNX + lane];
                for (int kk = 1; kk < 8; ++kk) D(kk, f) = 0.0;
            }
        DaeStats st;
        dae_split_integrate(lds, lane, p, tf, rtol, atol, h0, max_attempts, st);
        if (lane == kNX - 1) {
            const double u = D(0, 6), T = D(0, 5);
            const double P_total = (p[0] + p[1] + p[2] + p[3] + p[4]) * k::R * p[5];
            for (int f = 0; f < 5; ++f) {
                const double cc = D(0, f);
                flows[sidx * 5 + f] = (st.status == 0)
                                          ? cc * S * u * 60 * k::R * T / (P_total) * 1e6 * (P_total) / P_stp * 298 / T
                                          : -10000.0;
            }
            status[sidx] = st.status;
            atomicAdd(&counters[0], (unsigned long long)st.steps);
            atomicAdd(&counters[1], (unsigned long long)st.rejects);
            atomicAdd(&counters[2], (unsigned long long)st.newton_fail);
            atomicAdd(&counters[3], (unsigned long long)st.newton_iters);
            atomicAdd(&counters[4], (unsigned long long)st.nlu);
#ifdef SMC_METH_PROFILE
            for (int q = 0; q < 12; ++q) atomicAdd(&counters[8 + q], (unsigned long long)st.prof[q]);
#endif
        }
        if (y_final && lane < kNX)
            for (int f = 0; f < 7; ++f) y_final[sidx * kNS + f * kNX + lane] = D(0, f);
        if (lane == 0) atomicAdd(&counters[6], 1ULL);   // finished solves
        __builtin_amdgcn_wave_barrier();
    }
    split_command(lds, kCmdQuit);
    if (split && lane == 0) atomicAdd(&counters[7], 1ULL);
}

}  // namespace meth
}  // namespace smc

static thread_local std::string g_meth_err;
#define MH(call)                                                                 \
    do {                                                                         \
        hipError_t e_ = (call);                                                  \
        if (e_ != hipSuccess) {                                                  \
            g_meth_err = std::string(#call) + ": " + hipGetErrorString(e_);      \
            for (void *q : bufs) (void)hipFree(q);                               \
            return 1;                                                            \
        }                                                                        \
    } while (0)

#include <vector>

extern "C" {

const char *smc_meth_last_error(void) { return g_meth_err.c_str(); }

int smc_meth_residual_host(int device, const double *X, const double *dX, const double *params, int64_t n,
                           double *res) {
    using namespace smc::meth;
    std::vector<void *> bufs;
    if (n <= 0) return 0;
    MH(hipSetDevice(device));
    double *dXs, *dDs, *dP, *dR;
    const size_t sb = (size_t)n * NSTATE * sizeof(double);
    MH(hipMalloc(&dXs, sb)); bufs.push_back(dXs);
    MH(hipMalloc(&dDs, sb)); bufs.push_back(dDs);
    MH(hipMalloc(&dP, (size_t)n * NPAR * sizeof(double))); bufs.push_back(dP);
    MH(hipMalloc(&dR, sb)); bufs.push_back(dR);
    MH(hipMemcpy(dXs, X, sb, hipMemcpyHostToDevice));
    MH(hipMemcpy(dDs, dX, sb, hipMemcpyHostToDevice));
    MH(hipMemcpy(dP, params, (size_t)n * NPAR * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(residual_kernel, dim3((unsigned)n), dim3(64), 0, 0, dXs, dDs, dP, n, dR);
    MH(hipGetLastError());
    MH(hipMemcpy(res, dR, sb, hipMemcpyDeviceToHost));
    for (void *q : bufs) (void)hipFree(q);
    return 0;
}

int smc_meth_rate_host(int device, const double *in, const double *kin, int64_t n, double *out) {
    using namespace smc::meth;
    std::vector<void *> bufs;
    if (n <= 0) return 0;
    MH(hipSetDevice(device));
    double *dI, *dK, *dO;
    MH(hipMalloc(&dI, (size_t)n * 5 * 8)); bufs.push_back(dI);
    MH(hipMalloc(&dK, (size_t)n * 8 * 8)); bufs.push_back(dK);
    MH(hipMalloc(&dO, (size_t)n * 8)); bufs.push_back(dO);
    MH(hipMemcpy(dI, in, (size_t)n * 5 * 8, hipMemcpyHostToDevice));
    MH(hipMemcpy(dK, kin, (size_t)n * 8 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dI, dK, n, dO);
    MH(hipGetLastError());
    MH(hipMemcpy(out, dO, (size_t)n * 8, hipMemcpyDeviceToHost));
    for (void *q : bufs) (void)hipFree(q);
    return 0;
}

int smc_meth_loglike_host(int device, const double *y, const double *data, const double *sigma, int64_t n, int n_data,
                          double *lk) {
    using namespace smc::meth;
    std::vector<void *> bufs;
    if (n <= 0) return 0;
    MH(hipSetDevice(device));
    double *dY, *dD, *dS, *dL;
    MH(hipMalloc(&dY, (size_t)n * 5 * n_data * 8)); bufs.push_back(dY);
    MH(hipMalloc(&dD, (size_t)5 * n_data * 8)); bufs.push_back(dD);
    MH(hipMalloc(&dS, (size_t)n * 8)); bufs.push_back(dS);
    MH(hipMalloc(&dL, (size_t)n * 8)); bufs.push_back(dL);
    MH(hipMemcpy(dY, y, (size_t)n * 5 * n_data * 8, hipMemcpyHostToDevice));
    MH(hipMemcpy(dD, data, (size_t)5 * n_data * 8, hipMemcpyHostToDevice));
    MH(hipMemcpy(dS, sigma, (size_t)n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(loglike_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dY, dD, dS, n, n_data, dL);
    MH(hipGetLastError());
    MH(hipMemcpy(lk, dL, (size_t)n * 8, hipMemcpyDeviceToHost));
    for (void *q : bufs) (void)hipFree(q);
    return 0;
}

int smc_meth_dae_host(int device, const double *p0_all, const double *y0_all, int64_t n_solves, double tf, double rtol,
                      double atol, double h0, double S, double P_stp, double *flows, double *y_final, int32_t *status,
                      int64_t *stats, double *kernel_ms) {
    using namespace smc::meth;
    std::vector<void *> bufs;
    if (n_solves <= 0) return 0;
    MH(hipSetDevice(device));
    hipDeviceProp_t prop;
    MH(hipGetDeviceProperties(&prop, device));
    const bool v1 = getenv("SMC_METH_DAE_V1") != nullptr;   // debug: the thread-per-solve version (meth_dae.h)
    const bool v2 = getenv("SMC_METH_DAE_V2") != nullptr;   // debug: lane = node scans (meth_dae_wave.h)
    const bool v4 = smc::meth_split_enabled();               // two waves per solve (meth_dae_split.h)
    const int budget = getenv("SMC_METH_MAX_ATTEMPTS") ? atoi(getenv("SMC_METH_MAX_ATTEMPTS")) : kDaeMaxAttempts;
    int64_t nslots = ((n_solves + 63) / 64) * 64;
    const int64_t max_slots = (int64_t)prop.multiProcessorCount * 256;   // 4 waves per CU
    if (nslots > max_slots) nslots = max_slots;
    double *dws = nullptr, *dp, *dy0, *dfl, *dyf = nullptr;
    int *dst;
    unsigned long long *dcnt;
    if (v1) { MH(hipMalloc(&dws, (size_t)nslots * kWsDoubles * sizeof(double))); bufs.push_back(dws); }
    MH(hipMalloc(&dp, (size_t)n_solves * 18 * 8)); bufs.push_back(dp);
    MH(hipMalloc(&dy0, (size_t)n_solves * kNS * 8)); bufs.push_back(dy0);
    MH(hipMalloc(&dfl, (size_t)n_solves * 5 * 8)); bufs.push_back(dfl);
    MH(hipMalloc(&dst, (size_t)n_solves * sizeof(int))); bufs.push_back(dst);
    MH(hipMalloc(&dcnt, 24 * sizeof(unsigned long long))); bufs.push_back(dcnt);
    if (y_final) { MH(hipMalloc(&dyf, (size_t)n_solves * kNS * 8)); bufs.push_back(dyf); }
    MH(hipMemcpy(dp, p0_all, (size_t)n_solves * 18 * 8, hipMemcpyHostToDevice));
    MH(hipMemcpy(dy0, y0_all, (size_t)n_solves * kNS * 8, hipMemcpyHostToDevice));
    MH(hipMemset(dcnt, 0, 24 * sizeof(unsigned long long)));
    hipEvent_t e0, e1;
    MH(hipEventCreate(&e0));
    MH(hipEventCreate(&e1));
    MH(hipEventRecord(e0, 0));
    if (v1) {
        hipLaunchKernelGGL(dae_kernel, dim3((unsigned)(nslots / 64)), dim3(64), 0, 0, dws, nslots, dp, dy0, n_solves, tf,
                           rtol, atol, h0, budget, S, P_stp, dfl, dyf, dst, dcnt);
    } else if (v4 && !v2) {
        const int gpc = getenv("SMC_METH_WAVES_PER_CU") ? atoi(getenv("SMC_METH_WAVES_PER_CU")) : 4;   // workgroups (solves) per CU
        int64_t ngroups = (int64_t)prop.multiProcessorCount * (gpc >= 1 && gpc <= 4 ? gpc : 4);
        if (ngroups > n_solves) ngroups = n_solves;
        if (getenv("SMC_K8_SPLIT_DEBUG")) {
            int nb = -1, nb3 = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, dae_split_kernel, kSplitThreads, kLdsSplitDoubles * sizeof(double));
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb3, dae_elem_kernel, 64, kLdsDoubles * sizeof(double));
            fprintf(stderr, "[k8 split] resident workgroups per CU: v4 %d (LDS %zu B), v3 %d (LDS %zu B); shared memory per CU %zu B\n", nb,
                    kLdsSplitDoubles * sizeof(double), nb3, kLdsDoubles * sizeof(double), (size_t)prop.maxSharedMemoryPerMultiProcessor);
        }
        hipLaunchKernelGGL(dae_split_kernel, dim3((unsigned)ngroups), dim3(kSplitThreads), kLdsSplitDoubles * sizeof(double), 0, dp,
                           dy0, n_solves, tf, rtol, atol, h0, budget, S, P_stp, dfl, dyf, dst, dcnt, smc::meth_split_role_policy());
    } else if (!v2) {
        // one wave per SIMD is all that fits (512 VGPRs and 38.8 KB of LDS per wave); SMC_METH_WAVES_PER_CU < 4 thins the grid
        // for the occupancy-scaling measurement of profiles/r02_k8_occupancy.md
        const int wpc = getenv("SMC_METH_WAVES_PER_CU") ? atoi(getenv("SMC_METH_WAVES_PER_CU")) : 4;
        int64_t nwaves = (int64_t)prop.multiProcessorCount * (wpc >= 1 && wpc <= 4 ? wpc : 4);
        if (nwaves > n_solves) nwaves = n_solves;
        hipLaunchKernelGGL(dae_elem_kernel, dim3((unsigned)nwaves), dim3(64), kLdsDoubles * sizeof(double), 0, dp, dy0,
                           n_solves, tf, rtol, atol, h0, budget, S, P_stp, dfl, dyf, dst, dcnt);
    } else {
        int64_t nwaves = (int64_t)prop.multiProcessorCount * 4;
        if (nwaves > n_solves) nwaves = n_solves;
        hipLaunchKernelGGL(dae_wave_kernel, dim3((unsigned)nwaves), dim3(64), 8 * 7 * 64 * sizeof(double), 0, dp, dy0,
                           n_solves, tf, rtol, atol, h0, budget, S, P_stp, dfl, dyf, dst, dcnt);
    }
    MH(hipGetLastError());
    MH(hipEventRecord(e1, 0));
    MH(hipEventSynchronize(e1));
    float ms = 0.f;
    MH(hipEventElapsedTime(&ms, e0, e1));
    if (kernel_ms) *kernel_ms = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    MH(hipMemcpy(flows, dfl, (size_t)n_solves * 5 * 8, hipMemcpyDeviceToHost));
    MH(hipMemcpy(status, dst, (size_t)n_solves * sizeof(int), hipMemcpyDeviceToHost));
    if (y_final) MH(hipMemcpy(y_final, dyf, (size_t)n_solves * kNS * 8, hipMemcpyDeviceToHost));
    {
        unsigned long long h[8];
        MH(hipMemcpy(h, dcnt, sizeof h, hipMemcpyDeviceToHost));
        if (stats)
            for (int q = 0; q < 5; ++q) stats[q] = (int64_t)h[q];     // [4]: factorisations (0 from the debug kernels v1 / v2)
        if (!v1 && !v2 && (h[6] != (unsigned long long)n_solves || h[7] != 0)) {   // every solve exactly once, whole waves only
            g_meth_err = "dae_elem_kernel: " + std::to_string(h[6]) + " of " + std::to_string(n_solves) +
                         " solves finished, " + std::to_string(h[7]) + " waves split at a dequeue";
            for (void *q : bufs) (void)hipFree(q);
            return 1;
        }
    }
#ifdef SMC_METH_PROFILE
    {
        unsigned long long h[24];
        MH(hipMemcpy(h, dcnt, sizeof h, hipMemcpyDeviceToHost));
        static const char *nm[12] = {"build+factor", "residual", "forward", "backward", "total", "change_D", "predictor", "jac+transpose", "factor+newton", "error test", "Dupd+select", "update+norm"};   // v4: "build+factor" = the wave's chain, "forward" = command + both scans, "backward" = wait + middle node
        fprintf(stderr, "[meth profile] solves %lld  nlu/solve %.1f  newton/solve %.1f  steps/solve %.1f\n", (long long)n_solves,
                (double)h[4] / n_solves, (double)h[3] / n_solves, (double)h[0] / n_solves);
        for (int q = 0; q < 12; ++q)
            fprintf(stderr, "[meth profile] %-14s %10.0f cycles/solve  (%.1f %% of total)\n", nm[q], (double)h[8 + q] / n_solves,
                    100.0 * h[8 + q] / (double)h[12]);
        fprintf(stderr, "[meth profile] per factorisation %.0f cycles; per newton iteration: residual %.0f forward %.0f backward %.0f\n",
                (double)h[8] / h[4], (double)h[9] / h[3], (double)h[10] / h[3], (double)h[11] / h[3]);
    }
#endif
    for (void *q : bufs) (void)hipFree(q);
    return 0;
}

}  // extern "C"
