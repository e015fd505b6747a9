// attempt_probe.hip -- probe (not product code): time per RK45 attempt of ONE stiff Michaelis-Menten solve running alone
// on the GPU (the latency-bound tail of an early-tempering sweep), for the division variants of mm_rk45.h, alternating
// between them inside one process on one box.    hipcc -O3 --offload-arch=gfx950 -ffp-contract=on -o attempt_probe attempt_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "../python-based-sequential-monte-carlo-method-with-likelihood-tempering_amd/csrc/mm_rk45.h"
using namespace smc;

template <int DIV>
__global__ void __launch_bounds__(64) solve_one(double Vmax, double Km, double S0, const double *t, const double *P, int n_t,
                                                int nlive, double *out, int *att) {
    __shared__ double s_t[256], s_P[256];
    for (int i = threadIdx.x; i < n_t; i += 64) { s_t[i] = t[i]; s_P[i] = P[i]; }
    __syncthreads();
    if ((int)threadIdx.x >= nlive) return;
    MMItem it;
    if (!mm_item_begin<false>(it, Vmax, Km, S0, s_t, s_P, 0, n_t, 1e-3, 1e-6, nullptr)) return;
    int st;
    do { st = mm_item_attempt<false, DIV>(it, s_t, s_P, n_t, 1e-3, 1e-6, nullptr); } while (st == 0);
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
    return 0;
}
