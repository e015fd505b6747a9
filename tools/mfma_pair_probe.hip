// mfma_pair_probe.hip -- TWO interleaved chains of K8-like node steps in one wave (the two-ended elimination's situation): today's
// multiply + DPP butterfly steps against steps built on v_mfma_f64_4x4x4_4b_f64 (one MFMA + one cross-block DPP add).  A lone wave
// issues one instruction per >= 4 cycles, so with the latency of one chain hidden by the other the instruction count decides.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define FENCE() __builtin_amdgcn_sched_barrier(0)
template <int CTRL, int BANK = 0xF> __device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, BANK, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, BANK, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int BANK> __device__ __forceinline__ double dpp_upd(double old, double v) {   // lanes outside BANK keep `old`
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, 0xF, BANK, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, 0xF, BANK, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void swap16_pair(double v, double &x, double &y) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    x = __hiloint2double((int)b[0], (int)a[0]); y = __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ void swap32_pair(double v, double &x, double &y) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    x = __hiloint2double((int)b[0], (int)a[0]); y = __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double bcast(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double mfma444(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

// even-parity butterfly (over lane bits 0..2) and odd-parity one (bits 3..5) for two values side by side
__device__ __forceinline__ void sum8_pair_even(double &p, double &q) {
    FENCE(); double tp = dpp_mov<0xB1>(p), tq = dpp_mov<0xB1>(q); p += tp; q += tq;
    FENCE(); tp = dpp_mov<0x4E>(p); tq = dpp_mov<0x4E>(q); p += tp; q += tq;
    FENCE(); tp = dpp_mov<0x141>(p); tq = dpp_mov<0x141>(q); p += tp; q += tq; FENCE();
}
__device__ __forceinline__ void sum8_pair_odd(double &p, double &q) {
    FENCE(); double tp = dpp_mov<0x128>(p), tq = dpp_mov<0x128>(q); p += tp; q += tq;
    FENCE(); double px, py, qx, qy; swap16_pair(p, px, py); swap16_pair(q, qx, qy); p = px + py; q = qx + qy;
    FENCE(); swap32_pair(p, px, py); swap32_pair(q, qx, qy); p = px + py; q = qx + qy; FENCE();
}
// cross-block adds after the MFMA: layout A: partner block b ^ 2 (row_ror:8); layout B: partner b ^ 1 (two bank-masked rotations)
__device__ __forceinline__ void xblock_pair_A(double &p, double &q) {
    FENCE(); const double tp = dpp_mov<0x128>(p), tq = dpp_mov<0x128>(q); p += tp; q += tq; FENCE();
}
__device__ __forceinline__ void xblock_pair_B(double &p, double &q) {
    FENCE();
    double tp = dpp_upd<0x124, 0xA>(p, p), tq = dpp_upd<0x124, 0xA>(q, q);      // row_ror:4 into banks 1, 3
    tp = dpp_upd<0x12C, 0x5>(tp, p); tq = dpp_upd<0x12C, 0x5>(tq, q);            // row_ror:12 into banks 0, 2
    p += tp; q += tq; FENCE();
}

__global__ void __launch_bounds__(64) probe(const double *in, double *out, long long *cyc, int reps) {
    const int lane = threadIdx.x;
    const double X0 = in[lane], X1 = in[64 + lane], G0 = in[128 + lane], G1 = in[192 + lane], b0 = in[256 + lane], c1 = in[320 + lane] * 0.01,
                 c2 = in[384 + lane] * 0.01, c3 = in[448 + lane] * 0.01;
    long long c[6];
    // (a) today's two-ended forward: two chains, per step 2 x (3 FMA with broadcasts + multiply + butterfly), alternating parities
    double zT = b0, zB = b0 * 0.5;
    c[0] = clock64();
    for (int i = 0; i < reps; ++i) {
        double z6 = bcast(zT, 6), z5 = bcast(zT, 5), w5 = bcast(zB, 5);
        double tT = fma(-c3, z5, fma(-c2, z6, fma(-c1, zT, b0))), tB = fma(-c3, w5, fma(-c1, zB, b0));
        zT = X0 * tT; zB = X1 * tB; sum8_pair_even(zT, zB);
        z6 = bcast(zT, 48); z5 = bcast(zT, 40); w5 = bcast(zB, 40);
        tT = fma(-c3, z5, fma(-c2, z6, fma(-c1, zT, b0))); tB = fma(-c3, w5, fma(-c1, zB, b0));
        zT = X1 * tT; zB = X0 * tB; sum8_pair_odd(zT, zB);
    }
    c[1] = clock64();
    // (b) the same with MFMA steps: elementwise part + one MFMA + cross-block add
    double yT = b0, yB = b0 * 0.5;
    for (int i = 0; i < reps; ++i) {
        double z6 = bcast(yT, 40), z5 = bcast(yT, 24), w5 = bcast(yB, 24);
        double tT = fma(-c3, z5, fma(-c2, z6, fma(-c1, yT, b0))), tB = fma(-c3, w5, fma(-c1, yB, b0));
        yT = mfma444(X0, tT, 0.0); yB = mfma444(X1, tB, 0.0); xblock_pair_A(yT, yB);
        z6 = bcast(yT, 36); z5 = bcast(yT, 20); w5 = bcast(yB, 20);
        tT = fma(-c3, z5, fma(-c2, z6, fma(-c1, yT, b0))); tB = fma(-c3, w5, fma(-c1, yB, b0));
        yT = mfma444(X1, tT, 0.0); yB = mfma444(X0, tB, 0.0); xblock_pair_B(yT, yB);
    }
    c[2] = clock64();
    // (c) today's two-ended backward: x = z - sum(G * x_next)
    double xT = b0, xB = b0 * 0.5;
    for (int i = 0; i < reps; ++i) {
        double sT = G0 * xT, sB = G1 * xB; sum8_pair_even(sT, sB); xT = b0 - sT; xB = c1 - sB;
        sT = G1 * xT; sB = G0 * xB; sum8_pair_odd(sT, sB); xT = b0 - sT; xB = c1 - sB;
    }
    c[3] = clock64();
    // (d) MFMA backward: the right-hand side enters as the accumulator of a MFMA on -G (half of it per block of a pair)
    double uT = b0, uB = b0 * 0.5;
    for (int i = 0; i < reps; ++i) {
        uT = mfma444(G0, uT, b0); uB = mfma444(G1, uB, c1); xblock_pair_A(uT, uB);
        uT = mfma444(G1, uT, b0); uB = mfma444(G0, uB, c1); xblock_pair_B(uT, uB);
    }
    c[4] = clock64();
    out[lane] = zT + zB + yT + yB + xT + xB + uT + uB;
    if (lane == 0) for (int q = 0; q < 4; ++q) cyc[q] = c[q + 1] - c[q];
}
int main() {
    std::vector<double> h(512);
    for (int i = 0; i < 512; ++i) h[i] = 0.001 * ((i * 37) % 101) - 0.05;
    double *din, *dout; long long *dc;
    (void)hipMalloc(&din, 512 * 8); (void)hipMalloc(&dout, 64 * 8); (void)hipMalloc(&dc, 8 * 8);
    (void)hipMemcpy(din, h.data(), 512 * 8, hipMemcpyHostToDevice);
    const int reps = 20000;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, din, dout, dc, reps);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    long long c[4];
    (void)hipMemcpy(c, dc, 32, hipMemcpyDeviceToHost);
    printf("cycles per PAIR of node steps (two interleaved chains, one wave alone on its SIMD):\n");
    printf("  (a) forward, multiply + DPP butterfly (today)   %.1f\n", (double)c[0] / (2.0 * reps));
    printf("  (b) forward, MFMA 4x4x4_4b + cross-block add     %.1f\n", (double)c[1] / (2.0 * reps));
    printf("  (c) backward, multiply + DPP butterfly (today)  %.1f\n", (double)c[2] / (2.0 * reps));
    printf("  (d) backward, MFMA with the rhs as accumulator   %.1f\n", (double)c[3] / (2.0 * reps));
    return 0;
}
