// xlane_probe.hip -- checks the cross-lane primitives the K8 element-layout scans rely on (gfx950):
// all-reduce over the 8 lanes of a group (DPP quad_perm + row_half_mirror) and over the 8 groups
// (DPP row_ror:8, v_permlane16_swap, v_permlane32_swap); prints max errors and per-primitive latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

template <int CTRL> __device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double allsum_group8(double v) {   // over lane bits 0..2
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);   // row_half_mirror
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
__device__ __forceinline__ double allsum_across8(double v) {  // over lane bits 3..5
    v += dpp_mov<0x128>(v);   // row_ror:8
    v = swap16_sum(v);
    v = swap32_sum(v);
    return v;
}
__global__ void __launch_bounds__(64) probe(const double *in, double *out, long long *cyc, int reps) {
    const int lane = threadIdx.x;
    const double v = in[lane];
    out[lane] = allsum_group8(v);
    out[64 + lane] = allsum_across8(v);
    out[128 + lane] = __shfl(v, (lane & 7) * 8 + (lane >> 3));   // transpose by ds_bpermute
    double a = v;
    long long t0 = clock64();
    for (int i = 0; i < reps; ++i) a = allsum_group8(a) * 0.125;
    long long t1 = clock64();
    for (int i = 0; i < reps; ++i) a = allsum_across8(a) * 0.125;
    long long t2 = clock64();
    for (int i = 0; i < reps; ++i) a = __shfl(a, (lane & 7) * 8 + (lane >> 3)) * 1.0000001;
    long long t3 = clock64();
    for (int i = 0; i < reps; ++i) a = a * 1.0000001 + 1e-9;
    long long t4 = clock64();
    out[192 + lane] = a;
    if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; }
}
int main() {
    std::vector<double> h(64), o(256);
    for (int i = 0; i < 64; ++i) h[i] = std::sin(1.0 + i) * 3.0;
    double *din, *dout; long long *dc;
    hipMalloc(&din, 64 * 8); hipMalloc(&dout, 256 * 8); hipMalloc(&dc, 4 * 8);
    hipMemcpy(din, h.data(), 64 * 8, hipMemcpyHostToDevice);
    const int reps = 10000;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, din, dout, dc, reps);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    long long c[4];
    hipMemcpy(o.data(), dout, 256 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c, dc, 32, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0, e3 = 0;
    for (int l = 0; l < 64; ++l) {
        double s1 = 0, s2 = 0;
        for (int j = 0; j < 8; ++j) { s1 += h[(l & ~7) | j]; s2 += h[(l & 7) | (j << 3)]; }
        e1 = fmax(e1, fabs(o[l] - s1)); e2 = fmax(e2, fabs(o[64 + l] - s2));
        e3 = fmax(e3, fabs(o[128 + l] - h[(l & 7) * 8 + (l >> 3)]));
    }
    printf("group8 all-reduce max err %.3g | across8 all-reduce max err %.3g | transpose err %.3g\n", e1, e2, e3);
    printf("cycles per op: group8+mul %.1f  across8+mul %.1f  bpermute(f64)+mul %.1f  fma %.1f\n", (double)c[0] / reps,
           (double)c[1] / reps, (double)c[2] / reps, (double)c[3] / reps);
    return 0;
}
