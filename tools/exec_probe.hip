// exec_probe.hip -- does a dependent FP64 FMA chain run faster when only 16 (or 1) lanes of the wave are active?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64) probe(double *out, long long *cyc, int reps, int nactive) {
    const int lane = threadIdx.x;
    double a = 1.0 + lane * 1e-9;
    long long t0 = 0, t1 = 0;
    if (lane < nactive) {
        t0 = clock64();
#pragma unroll 1
        for (int i = 0; i < reps; ++i) {
            a = fma(a, 0.999999, 1e-7); a = fma(a, 0.999999, 1e-7); a = fma(a, 0.999999, 1e-7); a = fma(a, 0.999999, 1e-7);
            a = fma(a, 0.999999, 1e-7); a = fma(a, 0.999999, 1e-7); a = fma(a, 0.999999, 1e-7); a = fma(a, 0.999999, 1e-7);
        }
        t1 = clock64();
    }
    out[lane] = a;
    if (lane == 0) cyc[0] = t1 - t0;
}
int main() {
    double *d; long long *c;
    (void)hipMalloc(&d, 64 * 8); (void)hipMalloc(&c, 8);
    for (int na : {64, 32, 16, 8, 1}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, c, 20000, na);
        (void)hipDeviceSynchronize();
        long long h; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf("active lanes %2d: %.2f clock ticks per dependent v_fma_f64\n", na, (double)h / (20000.0 * 8));
    }
    // wall-clock for the 64-lane case to calibrate the tick
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, c, 2000000, 64);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("16M dependent FMAs: %.3f ms -> %.2f ns per FMA, %lld ticks -> tick = %.3f ns\n", ms, ms * 1e6 / 16e6, h, ms * 1e6 / (double)h);
    return 0;
}
