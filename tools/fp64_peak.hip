// tools/fp64_peak.hip -- what the FP64 vector pipeline of THIS device delivers on independent FMAs (the compute roof the
// roofline fractions in bench.py are set against is the datasheet's 78.6 TFLOP/s; this probe says how much of that a kernel of
// nothing but v_fma_f64 reaches on the box at hand).  Prints one JSON line.  Build: hipcc -O3 --offload-arch=gfx950 -o fp64_peak fp64_peak.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

constexpr int kChains = 16, kInner = 64;

__global__ void __launch_bounds__(256) fma_kernel(double *out, double a, double b, int iters) {
    double x[kChains];
#pragma unroll
    for (int i = 0; i < kChains; ++i) x[i] = 1.0 + 1e-3 * (threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < kInner; ++k)
#pragma unroll
            for (int i = 0; i < kChains; ++i) x[i] = fma(x[i], a, b);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < kChains; ++i) s += x[i];
    if (s == 12345.678) out[blockIdx.x * blockDim.x + threadIdx.x] = s;   // never true: keeps the chains alive
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    double *out;
    CK(hipMalloc(&out, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double best = 0.0, best_ms = 0.0;
    int best_wps = 0;
    const int iters = 4000;
    for (int blocks_per_cu : {1, 2, 4, 8}) {           // 256-thread blocks: 1, 2, 4, 8 waves per SIMD
        const int grid = cus * blocks_per_cu;
        hipLaunchKernelGGL(fma_kernel, dim3(grid), dim3(256), 0, 0, out, 0.999999, 1e-6, 10);     // warm-up
        CK(hipDeviceSynchronize());
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(fma_kernel, dim3(grid), dim3(256), 0, 0, out, 0.999999, 1e-6, iters);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double flop = 2.0 * (double)grid * 256 * kChains * kInner * iters;
            const double tf = flop / (ms * 1e-3) / 1e12;
            if (tf > best) { best = tf; best_ms = ms; best_wps = blocks_per_cu; }
        }
    }
    printf("{\"probe\": \"fp64_fma_peak\", \"device\": \"%s\", \"arch\": \"%s\", \"cu_count\": %d, \"clock_MHz\": %d, "
           "\"fp64_fma_tflops\": %.2f, \"waves_per_simd_at_best\": %d, \"kernel_ms_at_best\": %.3f, "
           "\"independent_chains_per_lane\": %d, \"datasheet_fp64_vector_tflops\": 78.6}\n",
           p.name, p.gcnArchName, cus, p.clockRate / 1000, best, best_wps, best_ms, kChains);
    return 0;
}
