// attempt_probe.hip -- probe (not product code): time per RK45 attempt of ONE stiff Michaelis-Menten solve running alone
// on the GPU (the latency-bound tail of an early-tempering sweep), for the division variants of mm_rk45.h, alternating
// between them inside one process on one box.    hipcc -O3 --offload-arch=gfx950 -ffp-contract=on -o attempt_probe attempt_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "../python-based-sequential-monte-carlo-method-with-likelihood-tempering_amd/csrc/mm_rk45.h"
using namespace smc;

// per-lane operands (as in the product, where every lane holds another particle): the compiler cannot prove the branches
// uniform and manages them with exec masks
template <int DIV>
__global__ void __launch_bounds__(64) solve_lanes(const double *theta, const double *t, const double *P, int n_t, int nlive,
                                                  double *out, int *att) {
    __shared__ double2 s_tp[257];
    mm_table_fill(s_tp, t, P, 1, n_t, threadIdx.x, 64);
    __syncthreads();
    if ((int)threadIdx.x >= nlive) return;
    const double Vmax = theta[threadIdx.x], Km = theta[64 + threadIdx.x], S0 = theta[128 + threadIdx.x];
    MMItem it;
    if (!mm_item_begin<false>(it, Vmax, Km, S0, s_tp, 0, n_t, 1e-3, 1e-6, nullptr)) return;
    int st;
    do { st = mm_item_attempt<false, DIV>(it, s_tp, n_t, 1e-3, 1e-6, nullptr); } while (st == 0);
    out[threadIdx.x] = it.sum_r2;
    att[threadIdx.x] = it.attempts;
}

__global__ void __launch_bounds__(256) busy_kernel(double *out, int iters) {   // dependent FP64 FMAs, ~ iters * 4 * 3.5 ns
    double x = threadIdx.x * 1e-3;
    for (int i = 0; i < iters; ++i) { x = fma(x, 0.999999, 1e-9); x = fma(x, 0.999999, 1e-9); x = fma(x, 0.999999, 1e-9); x = fma(x, 0.999999, 1e-9); }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = x;
}

template <int DIV>
__global__ void __launch_bounds__(64) solve_one(double Vmax, double Km, double S0, const double *t, const double *P, int n_t,
                                                int nlive, double *out, int *att) {
    __shared__ double2 s_tp[257];
    mm_table_fill(s_tp, t, P, 1, n_t, threadIdx.x, 64);
    __syncthreads();
    if ((int)threadIdx.x >= nlive) return;
    MMItem it;
    if (!mm_item_begin<false>(it, Vmax, Km, S0, s_tp, 0, n_t, 1e-3, 1e-6, nullptr)) return;
    int st;
    do { st = mm_item_attempt<false, DIV>(it, s_tp, n_t, 1e-3, 1e-6, nullptr); } while (st == 0);
    out[threadIdx.x] = it.sum_r2;
    att[threadIdx.x] = it.attempts;
}

int main() {
    const int n_t = 40;
    std::vector<double> t(n_t), P(n_t, 0.0);
    for (int i = 0; i < n_t; ++i) t[i] = 10.0 * i / (n_t - 1);
    double *dt, *dP, *dout; int *datt;
    (void)hipMalloc(&dt, n_t * 8); (void)hipMalloc(&dP, n_t * 8); (void)hipMalloc(&dout, 64 * 8); (void)hipMalloc(&datt, 64 * 4);
    (void)hipMemcpy(dt, t.data(), n_t * 8, hipMemcpyHostToDevice); (void)hipMemcpy(dP, P.data(), n_t * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const double Vmax = 10.0, Km = 3e-3, S0s[3] = {0.1, 0.5, 2.0};   // the band where RK45 runs on its stability limit
    for (int nlive : {1, 64}) {
        for (double S0 : S0s) {
            std::vector<float> ms5, ms6;
            int a5 = 0, a6 = 0; double r5 = 0, r6 = 0;
            for (int rep = 0; rep < 12; ++rep) {
                float ms;
                (void)hipEventRecord(e0, 0);
                hipLaunchKernelGGL((solve_one<kDivLean6>), dim3(1), dim3(64), 0, 0, Vmax, Km, S0, dt, dP, n_t, nlive, dout, datt);
                (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1); ms6.push_back(ms);
                (void)hipMemcpy(&a6, datt, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&r6, dout, 8, hipMemcpyDeviceToHost);
                (void)hipEventRecord(e0, 0);
                hipLaunchKernelGGL((solve_one<kDivLean5>), dim3(1), dim3(64), 0, 0, Vmax, Km, S0, dt, dP, n_t, nlive, dout, datt);
                (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1); ms5.push_back(ms);
                (void)hipMemcpy(&a5, datt, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&r5, dout, 8, hipMemcpyDeviceToHost);
            }
            std::sort(ms5.begin(), ms5.end()); std::sort(ms6.begin(), ms6.end());
            printf("live lanes %2d, S0 = %.1f: attempts %d / %d (six / five), sum_r2 equal: %s;  median ms six %.3f five %.3f  ->  "
                   "us per attempt six %.4f five %.4f (%.1f %%)\n", nlive, S0, a6, a5, r5 == r6 ? "yes" : "NO", ms6[6], ms5[6],
                   ms6[6] * 1e3 / a6, ms5[6] * 1e3 / a5, 100.0 * (ms5[6] / a5 - ms6[6] / a6) / (ms6[6] / a6));
        }
    }
    {   // the same solve with per-lane operands
        std::vector<double> th(192);
        for (int l = 0; l < 64; ++l) { th[l] = 10.0; th[64 + l] = 3e-3; th[128 + l] = 0.1; }
        double *dth; (void)hipMalloc(&dth, 192 * 8); (void)hipMemcpy(dth, th.data(), 192 * 8, hipMemcpyHostToDevice);
        for (int nlive : {1, 64}) {
            std::vector<float> ms6;
            int a6 = 0;
            for (int rep = 0; rep < 12; ++rep) {
                float ms;
                (void)hipEventRecord(e0, 0);
                hipLaunchKernelGGL((solve_lanes<kDivLean6>), dim3(1), dim3(64), 0, 0, dth, dt, dP, n_t, nlive, dout, datt);
                (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1); ms6.push_back(ms);
                (void)hipMemcpy(&a6, datt, 4, hipMemcpyDeviceToHost);
            }
            std::sort(ms6.begin(), ms6.end());
            printf("per-lane operands, live lanes %2d: attempts %d, median %.3f ms -> %.4f us per attempt\n", nlive, a6, ms6[6], ms6[6] * 1e3 / a6);
        }
    }
    {   // does a lone stiff solve run faster while the rest of the chip is kept busy (clock / power management)?
        std::vector<double> th(192);
        for (int l = 0; l < 64; ++l) { th[l] = 10.0; th[64 + l] = 3e-3; th[128 + l] = 0.1; }
        double *dth, *dbusy; (void)hipMalloc(&dth, 192 * 8); (void)hipMemcpy(dth, th.data(), 192 * 8, hipMemcpyHostToDevice);
        (void)hipMalloc(&dbusy, 256 * 8 * 4096);
        hipStream_t sa, sb; (void)hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
        for (int busy_blocks : {0, 1, 4, 16, 64, 128, 250, 0}) {
            std::vector<float> msv;
            int a6 = 0;
            for (int rep = 0; rep < 8; ++rep) {
                float ms;
                if (busy_blocks) hipLaunchKernelGGL(busy_kernel, dim3(busy_blocks), dim3(256), 0, sb, dbusy, 3000000);
                (void)hipEventRecord(e0, sa);
                hipLaunchKernelGGL((solve_lanes<kDivLean6>), dim3(1), dim3(64), 0, sa, dth, dt, dP, n_t, 1, dout, datt);
                (void)hipEventRecord(e1, sa); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1); msv.push_back(ms);
                (void)hipDeviceSynchronize();
                (void)hipMemcpy(&a6, datt, 4, hipMemcpyDeviceToHost);
            }
            std::sort(msv.begin(), msv.end());
            printf("lone stiff solve (1 live lane) with %4d busy blocks of 256 threads beside it: median %.3f ms -> %.4f us per attempt\n",
                   busy_blocks, msv[4], msv[4] * 1e3 / a6);
        }
    }
    return 0;
}
