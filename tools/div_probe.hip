// Probe (not product code): accuracy of v_rcp_f64 and of a lean FP64 division vs the IEEE sequence,
// and the FP64 FMA issue rate (peak check for the roofline).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#include <random>

__device__ __forceinline__ double lean_div(double a, double b) {
    double r = __builtin_amdgcn_rcp(b);
    double e = fma(-b, r, 1.0);
    r = fma(r, e, r);
    e = fma(-b, r, 1.0);
    r = fma(r, e, r);
    double q = a * r;
    double rem = fma(-b, q, a);
    return fma(rem, r, q);
}
__device__ __forceinline__ double lean_div1(double a, double b) {  // one Newton step only
    double r = __builtin_amdgcn_rcp(b);
    double e = fma(-b, r, 1.0);
    r = fma(r, e, r);
    double q = a * r;
    double rem = fma(-b, q, a);
    return fma(rem, r, q);
}
__device__ __forceinline__ double lean_div5(double a, double b) {  // Newton step on the quotient beside the one on r: 5-op chain
    const double r0 = __builtin_amdgcn_rcp(b);
    const double e = fma(-b, r0, 1.0);
    const double q0 = a * r0;
    const double r = fma(r0, e, r0);
    const double q1 = fma(q0, e, q0);
    const double rem = fma(-b, q1, a);
    return fma(rem, r, q1);
}
__global__ void k_div5(const double* a, const double* b, double* q5, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) q5[i] = lean_div5(a[i], b[i]);
}
__global__ void k_lean5chain(double* out, int iters) {
    double x = 1.0 + threadIdx.x * 1e-3;
    for (int i = 0; i < iters; ++i) { x = lean_div5(1.7, x + 0.3); x = lean_div5(1.7, x + 0.3); }
    out[threadIdx.x] = x;
}
__global__ void k_div(const double* a, const double* b, double* q_ieee, double* q_lean, double* q_lean1, double* rcp, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    q_ieee[i] = a[i] / b[i];
    q_lean[i] = lean_div(a[i], b[i]);
    q_lean1[i] = lean_div1(a[i], b[i]);
    rcp[i] = __builtin_amdgcn_rcp(b[i]);
}
__global__ void k_fma(double* out, int iters) {
    double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const double a = 0.999999, c = 1e-9;
    for (int i = 0; i < iters; ++i) {
        x0 = fma(x0, a, c); x1 = fma(x1, a, c); x2 = fma(x2, a, c); x3 = fma(x3, a, c);
        x4 = fma(x4, a, c); x5 = fma(x5, a, c); x6 = fma(x6, a, c); x7 = fma(x7, a, c);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_chain(double* out, int iters) {  // dependent FMA chain latency, one wave
    double x = threadIdx.x * 1e-3;
    const double a = 0.999999, c = 1e-9;
    for (int i = 0; i < iters; ++i) { x = fma(x, a, c); x = fma(x, a, c); x = fma(x, a, c); x = fma(x, a, c); }
    out[threadIdx.x] = x;
}
__global__ void k_divchain(double* out, int iters) {
    double x = 1.0 + threadIdx.x * 1e-3;
    for (int i = 0; i < iters; ++i) { x = 1.7 / (x + 0.3); x = 1.7 / (x + 0.3); }
    out[threadIdx.x] = x;
}
#include "../python-based-sequential-monte-carlo-method-with-likelihood-tempering_amd/csrc/mm_rk45.h"
__global__ void k_lean2chain(double* out, int iters) {
    double x = 1.0 + threadIdx.x * 1e-3;
    for (int i = 0; i < iters; ++i) { x = lean_div(1.7, x + 0.3); x = lean_div(1.7, x + 0.3); }
    out[threadIdx.x] = x;
}
__global__ void k_lean1chain(double* out, int iters) {
    double x = 1.0 + threadIdx.x * 1e-3;
    for (int i = 0; i < iters; ++i) { x = lean_div1(1.7, x + 0.3); x = lean_div1(1.7, x + 0.3); }
    out[threadIdx.x] = x;
}
__global__ void k_powchain(double* out, int iters) {
    double x = 0.5 + threadIdx.x * 1e-3;
    for (int i = 0; i < iters; ++i) { x = smc::pow_minus_fifth(x) * 0.37; x = smc::pow_minus_fifth(x) * 0.37; }
    out[threadIdx.x] = x;
}
__global__ void k_libpowchain(double* out, int iters) {
    double x = 0.5 + threadIdx.x * 1e-3;
    for (int i = 0; i < iters; ++i) { x = pow(x, -0.2) * 0.37; x = pow(x, -0.2) * 0.37; }
    out[threadIdx.x] = x;
}
__global__ void k_addchain(double* out, int iters) {
    double x = threadIdx.x * 1e-3;
    for (int i = 0; i < iters; ++i) { x = x + 1e-9; x = x * 0.999999; x = x + 1e-9; x = x * 0.999999; }
    out[threadIdx.x] = x;
}
int main() {
    const int n = 1 << 22;
    std::mt19937_64 g(1);
    std::vector<double> a(n), b(n);
    std::uniform_real_distribution<double> U(-40, 40);
    for (int i = 0; i < n; ++i) { a[i] = std::ldexp(1.0 + (g() >> 11) * 0x1p-53, (int)U(g)) * ((g() & 1) ? 1 : -1);
                                  b[i] = std::ldexp(1.0 + (g() >> 11) * 0x1p-53, (int)U(g)); }
    double *da, *db, *q0, *q1, *q2, *rc;
    hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&q0, n * 8); hipMalloc(&q1, n * 8); hipMalloc(&q2, n * 8); hipMalloc(&rc, n * 8);
    hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
    k_div<<<n / 256, 256>>>(da, db, q0, q1, q2, rc, n);
    std::vector<double> h0(n), h1(n), h2(n), hr(n);
    hipMemcpy(h0.data(), q0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), q1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(h2.data(), q2, n * 8, hipMemcpyDeviceToHost); hipMemcpy(hr.data(), rc, n * 8, hipMemcpyDeviceToHost);
    long mis_ieee = 0, mis1 = 0, mis2 = 0; double maxr = 0;
    for (int i = 0; i < n; ++i) {
        double ref = a[i] / b[i];
        if (h0[i] != ref) mis_ieee++;
        if (h1[i] != ref) mis1++;
        if (h2[i] != ref) mis2++;
        double e = std::fabs(hr[i] * b[i] - 1.0); if (e > maxr) maxr = e;
    }
    printf("n=%d  device a/b != host a/b: %ld   lean(2 newton) mismatches: %ld   lean(1 newton) mismatches: %ld   max |rcp*b-1| = %.3e (2^%.1f)\n",
           n, mis_ieee, mis1, mis2, maxr, std::log2(maxr));
    {   // the 5-op variant (the product's lean_div this round): wide-exponent operands above, then operand pairs shaped like
        // the Michaelis-Menten right-hand side: a = -Vmax*S, b = Km + S with Vmax, Km in (0,10), S in (0, 2]
        double* q5; hipMalloc(&q5, n * 8);
        std::vector<double> h5(n);
        k_div5<<<n / 256, 256>>>(da, db, q5, n);
        hipMemcpy(h5.data(), q5, n * 8, hipMemcpyDeviceToHost);
        long mis5 = 0;
        for (int i = 0; i < n; ++i) if (h5[i] != a[i] / b[i]) mis5++;
        std::uniform_real_distribution<double> U10(0.0, 10.0), U2(1e-9, 2.0);
        std::vector<double> a2(n), b2(n);
        for (int i = 0; i < n; ++i) { const double Vm = U10(g), Km = (i & 1) ? U10(g) : U10(g) * 1e-3, S = U2(g); a2[i] = -Vm * S; b2[i] = Km + S; }
        hipMemcpy(da, a2.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b2.data(), n * 8, hipMemcpyHostToDevice);
        k_div5<<<n / 256, 256>>>(da, db, q5, n);
        hipMemcpy(h5.data(), q5, n * 8, hipMemcpyDeviceToHost);
        long mis5b = 0;
        for (int i = 0; i < n; ++i) if (h5[i] != a2[i] / b2[i]) mis5b++;
        printf("lean 5-op chain: mismatches vs host a/b: %ld of %d (wide exponents), %ld of %d (RHS-shaped operands)\n", mis5, n, mis5b, n);
        hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
    }
    // FMA peak
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    double* out; hipMalloc(&out, 256 * 8 * 4096 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = p.multiProcessorCount * 8;
    k_fma<<<blocks, 256>>>(out, 1000); hipDeviceSynchronize();
    hipEventRecord(e0); k_fma<<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 2.0 * 8 * iters * 256.0 * blocks;
    printf("FP64 FMA throughput: %.2f TFLOP/s (%d CUs, clock %d MHz)\n", flops / (ms * 1e-3) / 1e12, p.multiProcessorCount, p.clockRate / 1000);
    hipEventRecord(e0); k_chain<<<1, 64>>>(out, 100000); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("dependent FMA chain: %.2f ns per FMA (single wave)\n", ms * 1e6 / (4.0 * 100000));
    hipEventRecord(e0); k_divchain<<<1, 64>>>(out, 100000); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("dependent (add + IEEE div) chain: %.2f ns per add+div (single wave)\n", ms * 1e6 / (2.0 * 100000));
#define TIME(K, label, per) do { hipEventRecord(e0); K<<<1, 64>>>(out, 100000); hipEventRecord(e1); hipEventSynchronize(e1); \
        hipEventElapsedTime(&ms, e0, e1); printf("%s: %.2f ns\n", label, ms * 1e6 / ((per) * 100000.0)); } while (0)
    TIME(k_lean2chain, "dependent (add + lean 2-newton div)", 2);
    TIME(k_lean1chain, "dependent (add + lean 1-newton div)", 2);
    TIME(k_lean5chain, "dependent (add + lean 5-op-chain div)", 2);
    TIME(k_powchain, "dependent (pow_minus_fifth + mul)", 2);
    TIME(k_libpowchain, "dependent (ocml pow + mul)", 2);
    TIME(k_addchain, "dependent add/mul op", 4);
    return 0;
}
