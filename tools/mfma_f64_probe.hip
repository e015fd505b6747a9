// mfma_f64_probe.hip -- (1) the lane <-> element maps of v_mfma_f64_4x4x4_4b_f64 (4 blocks of 4x4x4, one f64 of A, B and D per lane),
// derived from one-hot operands; (2) the latency of a chained 8x8 mat-vec step built on it (MFMA + cross-block add) against the
// element-layout step K8 uses today (multiply + 3-level DPP butterfly).  gfx950.  Prints the maps as JSON + cycles per step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CTRL> __device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double allsum_group8(double v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    return v;
}
__device__ __forceinline__ double swap16_sum(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double swap32_sum(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double mfma444(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

// map[la * 64 + lb] = D lane that is non-zero when A is one-hot at lane la and B one-hot at lane lb (-1: none, -2: several)
__global__ void __launch_bounds__(64) layout_kernel(int *map) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double d = mfma444(lane == la ? 1.0 : 0.0, lane == lb ? 1.0 : 0.0, 0.0);
            const unsigned long long m = __ballot(d != 0.0);
            if (lane == 0) map[la * 64 + lb] = m == 0 ? -1 : (__popcll(m) == 1 ? __ffsll((long long)m) - 1 : -2);
        }
}

__global__ void __launch_bounds__(64) latency_kernel(const double *in, double *out, long long *cyc, int reps) {
    const int lane = threadIdx.x;
    double X0 = in[lane], X1 = in[64 + lane], t = in[128 + lane], a;
    long long c[8];
    // (a) today's element-layout step: multiply + butterfly over 8 lanes, alternating directions
    a = t;
    c[0] = clock64();
    for (int i = 0; i < reps; ++i) {
        a = allsum_group8(X0 * a);
        a = dpp_mov<0x128>(X1 * a) + X1 * a;      // stands in for the across8 reduction's first level ...
        a = swap32_sum(swap16_sum(a));             // ... and its two permlane levels
    }
    c[1] = clock64();
    // (b) MFMA step: one 4-block MFMA + cross-block add, alternating swap16 / swap32 (even / odd node layouts)
    double b = t;
    for (int i = 0; i < reps; ++i) {
        b = swap16_sum(mfma444(X0, b, 0.0));
        b = swap32_sum(mfma444(X1, b, 0.0));
    }
    c[2] = clock64();
    // (c) MFMA alone, chained through B
    double d = t;
    for (int i = 0; i < 2 * reps; ++i) d = mfma444(X0, d, 0.0);
    c[3] = clock64();
    // (d) MFMA chained through C (accumulate)
    double e = t;
    for (int i = 0; i < 2 * reps; ++i) e = mfma444(X0, X1, e);
    c[4] = clock64();
    // (e) MFMA step with the elementwise part of the forward scan: t = b - ld z - lx zx with zx read from a fixed lane
    double f = t;
    for (int i = 0; i < reps; ++i) {
        double zx = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(f), 24), __builtin_amdgcn_readlane(__double2loint(f), 24));
        double tt = fma(-X1, zx, fma(-X0, f, t));
        f = swap16_sum(mfma444(X0, tt, 0.0));
        zx = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(f), 40), __builtin_amdgcn_readlane(__double2loint(f), 40));
        tt = fma(-X0, zx, fma(-X1, f, t));
        f = swap32_sum(mfma444(X1, tt, 0.0));
    }
    c[5] = clock64();
    // (f) today's step with the same elementwise part
    double g = t;
    for (int i = 0; i < reps; ++i) {
        double zx = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(g), 6), __builtin_amdgcn_readlane(__double2loint(g), 6));
        double tt = fma(-X1, zx, fma(-X0, g, t));
        g = allsum_group8(X0 * tt);
        zx = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(g), 48), __builtin_amdgcn_readlane(__double2loint(g), 48));
        tt = fma(-X0, zx, fma(-X1, g, t));
        const double p = X1 * tt;
        g = swap32_sum(swap16_sum(p + dpp_mov<0x128>(p)));
    }
    c[6] = clock64();
    out[lane] = a + b + d + e + f + g;
    if (lane == 0) for (int q = 0; q < 6; ++q) cyc[q] = c[q + 1] - c[q];
}

int main() {
    int *dmap;
    hipMalloc(&dmap, 64 * 64 * sizeof(int));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dmap);
    if (hipDeviceSynchronize() != hipSuccess) { printf("layout kernel failed\n"); return 1; }
    std::vector<int> map(64 * 64);
    hipMemcpy(map.data(), dmap, map.size() * sizeof(int), hipMemcpyDeviceToHost);
    printf("{\"mfma_f64_4x4x4_4b_onehot_map\": [");
    for (int i = 0; i < 64 * 64; ++i) printf("%d%s", map[i], i + 1 < 64 * 64 ? "," : "");
    printf("]}\n");
    std::vector<double> h(192);
    for (int i = 0; i < 192; ++i) h[i] = 0.001 * ((i * 37) % 101) - 0.05;
    double *din, *dout;
    long long *dc;
    hipMalloc(&din, 192 * 8); hipMalloc(&dout, 64 * 8); hipMalloc(&dc, 8 * 8);
    hipMemcpy(din, h.data(), 192 * 8, hipMemcpyHostToDevice);
    const int reps = 20000;
    hipLaunchKernelGGL(latency_kernel, dim3(1), dim3(64), 0, 0, din, dout, dc, reps);
    if (hipDeviceSynchronize() != hipSuccess) { printf("latency kernel failed\n"); return 1; }
    long long c[6];
    hipMemcpy(c, dc, 48, hipMemcpyDeviceToHost);
    printf("cycles per NODE STEP (one wave alone on its SIMD):\n");
    printf("  (a) today: multiply + 3-level butterfly            %.1f\n", (double)c[0] / (2.0 * reps));
    printf("  (b) MFMA 4x4x4_4b + cross-block swap-add           %.1f\n", (double)c[1] / (2.0 * reps));
    printf("  (c) MFMA alone, chained through B                  %.1f\n", (double)c[2] / (2.0 * reps));
    printf("  (d) MFMA alone, chained through C                  %.1f\n", (double)c[3] / (2.0 * reps));
    printf("  (e) MFMA step + readlane + 2 FMA (forward scan)    %.1f\n", (double)c[4] / (2.0 * reps));
    printf("  (f) today's step + readlane + 2 FMA (forward scan) %.1f\n", (double)c[5] / (2.0 * reps));
    return 0;
}
