// gj_probe.hip -- latency of the element-layout 7x7 Gauss-Jordan inversion (lane = 8 r + c), variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__device__ __forceinline__ double lane_bcast(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
__device__ __forceinline__ double recip1(double a) { const double x = __builtin_amdgcn_rcp(a); return fma(x, fma(-a, x, 1.0), x); }

template <int KK> __device__ __forceinline__ void gj2_step(double &a, double &akk, int lane) {
    const int r = lane >> 3, c = lane & 7;
    const double p = recip1(akk);
    const int ulo = __builtin_amdgcn_ds_swizzle(__double2loint(a), 0x18 | (KK << 5));
    const int uhi = __builtin_amdgcn_ds_swizzle(__double2hiint(a), 0x18 | (KK << 5));
    const double u = __hiloint2double(uhi, ulo);
    const double v = __shfl(a, KK * 8 + c);
    const double gen = fma(-(u * v), p, a);
    akk = lane_bcast(gen, KK < 6 ? 9 * KK + 9 : 0);
    const double ap = a * p;
    const bool rk = r == KK, ck = c == KK;
    const double on_row = ck ? p : ap, off_row = ck ? -ap : gen;
    a = rk ? on_row : off_row;
}
template <int VAR> __device__ __forceinline__ double gj(double a, int lane) {
    const int r = lane >> 3, c = lane & 7;
    if (VAR == 0) {           // rolled, as in meth_dae_elem.h
        double akk = lane_bcast(a, 0);
#pragma unroll 1
        for (int kk = 0; kk < 7; ++kk) {
            const double p = recip1(akk);
            const double u = __shfl(a, (lane & ~7) | kk), v = __shfl(a, kk * 8 + c);
            const double gen = fma(-(u * v), p, a);
            akk = lane_bcast(gen, kk < 6 ? 9 * kk + 9 : 0);
            const double ap = a * p;
            const bool rk = r == kk, ck = c == kk;
            const double on_row = ck ? p : ap, off_row = ck ? -ap : gen;
            a = rk ? on_row : off_row;
        }
    } else if (VAR == 1) {    // fully unrolled
        double akk = lane_bcast(a, 0);
#pragma unroll
        for (int kk = 0; kk < 7; ++kk) {
            const double p = recip1(akk);
            const double u = __shfl(a, (lane & ~7) | kk), v = __shfl(a, kk * 8 + c);
            const double gen = fma(-(u * v), p, a);
            akk = lane_bcast(gen, kk < 6 ? 9 * kk + 9 : 0);
            const double ap = a * p;
            const bool rk = r == kk, ck = c == kk;
            const double on_row = ck ? p : ap, off_row = ck ? -ap : gen;
            a = rk ? on_row : off_row;
        }
    } else if (VAR == 2) {    // unrolled; the element of the pivot column in my row by ds_swizzle (constant pattern)
        double akk = lane_bcast(a, 0);
        gj2_step<0>(a, akk, lane); gj2_step<1>(a, akk, lane); gj2_step<2>(a, akk, lane); gj2_step<3>(a, akk, lane);
        gj2_step<4>(a, akk, lane); gj2_step<5>(a, akk, lane); gj2_step<6>(a, akk, lane);
    }
    return a;
}
template <int VAR> __global__ void __launch_bounds__(64) probe(const double *in, double *out, long long *cyc, int reps) {
    const int lane = threadIdx.x;
    double a = in[lane];
    out[lane] = gj<VAR>(a, lane);
    long long t0 = clock64();
    double x = a;
    for (int i = 0; i < reps; ++i) x = gj<VAR>(x, lane);   // inverse of inverse ... stays bounded
    long long t1 = clock64();
    out[64 + lane] = x;
    if (lane == 0) cyc[0] = t1 - t0;
}
int main() {
    std::vector<double> h(64, 0.0), o(128);
    for (int r = 0; r < 7; ++r) for (int c = 0; c < 7; ++c) h[r * 8 + c] = (r == c ? 4.0 : 0.0) + std::sin(1.0 + 3 * r + 7 * c);
    double *din, *dout; long long *dc;
    (void)hipMalloc(&din, 64 * 8); (void)hipMalloc(&dout, 128 * 8); (void)hipMalloc(&dc, 8);
    (void)hipMemcpy(din, h.data(), 64 * 8, hipMemcpyHostToDevice);
    const int reps = 2000;
    for (int var = 0; var < 3; ++var) {
        if (var == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, din, dout, dc, reps);
        if (var == 1) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, din, dout, dc, reps);
        if (var == 2) hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, din, dout, dc, reps);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        long long c; (void)hipMemcpy(o.data(), dout, 128 * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        double err = 0;
        for (int r = 0; r < 7; ++r) for (int cc = 0; cc < 7; ++cc) {
            double s = 0; for (int k = 0; k < 7; ++k) s += o[r * 8 + k] * h[k * 8 + cc];
            err = fmax(err, fabs(s - (r == cc)));
        }
        printf("variant %d: |X A - I| max %.3g, %.0f cycles per 7x7 inversion (%.0f per pivot)\n", var, err, (double)c / reps, (double)c / reps / 7);
    }
    return 0;
}
