// mfma_occ_probe.hip -- K8 v4's regime: ONE chain per wave, TWO waves per SIMD on every SIMD of the chip (8 waves per CU, forced by
// 20 KB of LDS per wave).  Node steps as today (multiply + 3-level DPP butterfly) against steps on v_mfma_f64_4x4x4_4b_f64 (one
// MFMA + one cross-block DPP add; tools/mfma_pair_probe.hip has the block layouts).  At this occupancy the kernel is paced by
// instruction issue, so what is compared is the whole-chip time of N dependent steps per wave, not a lone wave's latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CTRL, int BANK = 0xF> __device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, BANK, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, BANK, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int BANK> __device__ __forceinline__ double dpp_upd(double old, double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, 0xF, BANK, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, 0xF, BANK, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double swap_add16(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double swap_add32(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double bcast(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double mfma444(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ double sum8_even(double p) { p += dpp_mov<0xB1>(p); p += dpp_mov<0x4E>(p); p += dpp_mov<0x141>(p); return p; }
__device__ __forceinline__ double sum8_odd(double p) { p += dpp_mov<0x128>(p); p = swap_add16(p); return swap_add32(p); }
__device__ __forceinline__ double xblock_A(double p) { return p + dpp_mov<0x128>(p); }
__device__ __forceinline__ double xblock_B(double p) {
    double t = dpp_upd<0x124, 0xA>(p, p);
    t = dpp_upd<0x12C, 0x5>(t, p);
    return p + t;
}

// MODE 0: today's forward + backward steps;  MODE 1: MFMA steps.  LDS traffic as in the kernel: 4 operand loads + 1 store per
// forward step, 1 load + 1 store per backward step (lds: the wave's own 20 KB).
template <int MODE>
__global__ void __launch_bounds__(64) probe(const double *in, double *out, int reps) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2560; i += 64) lds[i] = in[i & 511] * 0.01;
    __syncthreads();
    const double X0 = in[lane], X1 = in[64 + lane], G0 = in[128 + lane], G1 = in[192 + lane];
    double z = in[256 + lane];
    for (int r = 0; r < reps; ++r) {
        // inward scan, two steps (the two parities)
        for (int q = 0; q < 12; ++q) {
            const double *row = lds + (2 * q) * 100;
            double b0 = row[lane & 7], c1 = row[8 + (lane & 7)], c2 = row[16 + (lane & 7)], c3 = row[24 + (lane & 7)];
            double z6 = bcast(z, MODE ? 40 : 6), z5 = bcast(z, MODE ? 24 : 5);
            double t = fma(-c3, z5, fma(-c2, z6, fma(-c1, z, b0)));
            z = MODE ? xblock_A(mfma444(X0, t, 0.0)) : sum8_even(X0 * t);
            lds[2000 + q * 18 + (lane & 7)] = z;
            row += 100;
            b0 = row[lane >> 3]; c1 = row[8 + (lane >> 3)]; c2 = row[16 + (lane >> 3)]; c3 = row[24 + (lane >> 3)];
            z6 = bcast(z, MODE ? 36 : 48); z5 = bcast(z, MODE ? 20 : 40);
            t = fma(-c3, z5, fma(-c2, z6, fma(-c1, z, b0)));
            z = MODE ? xblock_B(mfma444(X1, t, 0.0)) : sum8_odd(X1 * t);
            lds[2000 + q * 18 + 9 + (lane >> 3)] = z;
        }
        // outward scan
        for (int q = 11; q >= 0; --q) {
            double zk = lds[2000 + q * 18 + 9 + (lane >> 3)];
            z = MODE ? xblock_A(mfma444(G0, z, (lane & 4) ? 0.0 : zk)) : zk - sum8_even(G0 * z);
            lds[2300 + q * 18 + (lane & 7)] = z;
            zk = lds[2000 + q * 18 + (lane & 7)];
            z = MODE ? xblock_B(mfma444(G1, z, (lane & 8) ? 0.0 : zk)) : zk - sum8_odd(G1 * z);
            lds[2300 + q * 18 + 9 + (lane >> 3)] = z;
        }
        z = z * 1e-3 + in[256 + lane];
    }
    out[blockIdx.x * 64 + lane] = z;
}

int main() {
    std::vector<double> h(512);
    for (int i = 0; i < 512; ++i) h[i] = 0.001 * ((i * 37) % 101) - 0.05;
    double *din, *dout;
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int waves = prop.multiProcessorCount * 8;
    (void)hipMalloc(&din, 512 * 8); (void)hipMalloc(&dout, (size_t)waves * 64 * 8);
    (void)hipMemcpy(din, h.data(), 512 * 8, hipMemcpyHostToDevice);
    const int reps = 4000;
    for (int pass = 0; pass < 2; ++pass)
        for (int mode = 0; mode < 2; ++mode)
            for (int occ = 0; occ < 2; ++occ) {
                const int grid = occ ? waves : prop.multiProcessorCount * 4;      // two waves per SIMD, or one
                hipEvent_t e0, e1;
                (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                (void)hipEventRecord(e0, 0);
                if (mode) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(64), 20480, 0, din, dout, reps);
                else hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(64), 20480, 0, din, dout, reps);
                (void)hipEventRecord(e1, 0);
                if (hipEventSynchronize(e1) != hipSuccess) { printf("kernel failed\n"); return 1; }
                float ms = 0;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (pass) printf("%s steps, %d wave(s) per SIMD: %.2f ms for %d x 48 node steps per wave = %.1f ns per step per wave, %.2f G steps/s on the chip\n",
                                 mode ? "MFMA" : "today's", occ ? 2 : 1, ms, reps, 1e6 * ms / (reps * 48.0), grid * reps * 48.0 / (ms * 1e6));
            }
    return 0;
}
