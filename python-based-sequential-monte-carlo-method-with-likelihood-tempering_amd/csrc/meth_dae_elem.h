// meth_dae_elem.h -- K8 v3: one solve per wave; block elimination and the two substitution scans in ELEMENT layout.
//
// Measured on v2 (meth_dae_wave.h, cycle counters, tools/meth_dae_bench.py): 78 % of a solve is the three scans
// over the 51 nodes (forward 35 %, factorisation 27 %, backward 16 %), and a scan step is ISSUE-bound: with
// lane = node only ONE lane works on the 7x7 block of the current node, yet every wave64 FP64 instruction
// occupies the SIMD for >= 4 cycles whatever the exec mask says (174 instructions = 886 cycles per forward step).
//
// v3 keeps lane = node for everything that is parallel over nodes (residuals, Jacobian blocks, predictor, norms)
// and gives the scans a second layout: lane = 8 r + c holds ELEMENT (r, c) of the 7x7 block of EVERY node,
//      X[i] = (D'_i)^{-1}[r][c]      G[i] = ((D'_i)^{-1} U_i)[r][c]      i = 0..50   (2 x 51 registers per lane)
//   * a scan step is then one multiply, a 3-level butterfly over the 8 lanes that share a row, and one subtract:
//         forward   z_i = X_i (b_i - L_i z_{i-1})      backward   x_i = z_i - G_i x_{i+1}
//     (L_i has three non-zeros per row: diagonal, the u column, one T entry; its coefficients come from LDS);
//   * consecutive nodes ALTERNATE between the layout above and its transpose, so the reduced vector lands exactly
//     where the next step needs it: even nodes reduce over c with DPP (quad_perm, row_half_mirror), odd nodes over r
//     with DPP row_ror:8 + v_permlane16_swap + v_permlane32_swap (gfx950).  No transposition on the chain;
//   * the factorisation is a Gauss-Jordan inversion of D'_i = D_i - L_i G_{i-1} spread over the 49 lanes (pivot
//     row / column by ds_bpermute, reciprocal pivot wave-uniform); rows 5/6 are already swapped by node_eval so
//     that no pivoting is needed (meth_dae.h).  The inverse is explicit because the iteration matrix of a modified
//     Newton method only steers convergence: the converged step does not depend on it;
//   * node-layout results reach the element layout through a 2.8 KB LDS staging row (Jacobian rows, right-hand
//     sides) and come back the same way.
// LDS per wave: differences array 8 x 7 x 51, L/U coefficients 51 x 24, staging 357 + 408 doubles = 38.8 KB, so four
// waves (one per SIMD) still share a CU.  Time stepping, Newton control and error tests are those of meth_dae.h.
// PARITY UNPINNED against the reference's IDA (see meth_dae.h); checked against v2, the CPU build and the CPU checker.
//
// Round 4: TWO-ENDED elimination.  K8 is bound by dependent chains at one wave per SIMD (vector ALUs busy 44 % of the cycles,
// profiles/r04_k8_oneway_pmc_sq_summary.json): a scan step waits ~13 cycles for every dependent FP64 operation while the issue slots
// stay empty, and neither a second wave per SIMD (204 VGPRs of factors, 38.8 KB of LDS per solve - for THIS kernel: meth_dae_split.h
// later gives each chain a wave of its own and gets there) nor the matrix cores (a chained v_mfma_f64_4x4x4 costs 48 cycles) are a
// way out.  What a single wave CAN do is run two independent chains in one instruction
// stream.  The block-tridiagonal system is therefore eliminated from BOTH ends at once ("twisted" / burn-at-both-ends
// factorisation): nodes 0 .. 24 downwards as before (D'_i = D_i - L_i G_{i-1},  G_i = D'_i^{-1} U_i), nodes 50 .. 26 upwards
// (D''_i = D_i - U_i H_{i+1},  H_i = D''_i^{-1} L_i), meeting in node 25 (D*_25 = D_25 - L_25 G_24 - U_25 H_26); the right-hand
// side runs inwards on both chains (z_i = X_i (b_i - L_i z_{i-1}),  w_i = X_i (b_i - U_i w_{i+1})), the middle node is solved,
// and the solution runs outwards on both (x_i = z_i - G_i x_{i+1},  x_i = w_i - H_i x_{i-1}).  Same flops, same storage (one
// inverse and one coupling factor per node), half the chain length; node k and node 50 - k have the same parity, hence the same
// lane layout, and are written side by side in one basic block so that the compiler interleaves the two dependency chains.
// Without pivoting the two-ended elimination is as accurate as the one-way one on this matrix (180 iteration matrices over
// prior-box parameters, states along solves and c = 1e-5 .. 10: worst relative error 3.8e-10 for both against a pivoted dense
// solve; the check is described in DESIGN.md 4.5).  SMC_K8_TWISTED=0 builds the one-way scans of rounds 1-3 (A/B).
#pragma once
#include <hip/hip_runtime.h>

#include "meth_dae.h"
#include "meth_dae_wave.h"

#ifndef SMC_K8_TWISTED
#define SMC_K8_TWISTED 1
#endif

namespace smc {
namespace meth {

constexpr int kMid = kNX / 2;    // node where the two elimination chains meet (25)
static_assert(kNX == 2 * kMid + 1 && (kMid & 1) == 1, "the paired scans assume 51 nodes: pairs (k, 50 - k), k = 0 .. 24, and an odd middle node");

// Row strides chosen against LDS bank conflicts where lane = node touches its own row (64 banks of 4 bytes): a stride of 24
// doubles (48 banks) or 8 doubles (16 banks) puts every fourth lane on the same banks - 16-way conflicts on every write of the
// coefficient rows and every read of the solution rows; 25 and 9 doubles (50 / 18 banks) leave 2-way conflicts.
constexpr int kCfRow = 25, kZRow = 9;
constexpr int kLdsD = 0;                         // D[k][f][node], k < 8
constexpr int kLdsCf = kLdsD + 8 * 7 * kNX;      // per node kCfRow: Ld[0..6],0 | Lx[0..6],0 | Ud[0..5],0,U65 | pad
constexpr int kLdsB = kLdsCf + kNX * kCfRow;         // b[node][7]  (also the staging row of the Jacobian transposition)
constexpr int kLdsZ = kLdsB + kNX * 7;           // z / x [node][kZRow]; slot 7 of a node takes the writes of the lanes that hold no result
constexpr int kLdsDoubles = kLdsZ + kNX * kZRow; // 4947 doubles = 39576 bytes: four waves (one per SIMD) still share a CU's 160 KB
static_assert(kLdsDoubles * 8 * 4 <= 160 * 1024, "four solves per CU");

struct DViewE {   // differences array, node-major within a row (only lanes < kNX may touch it)
    double *s;
    int lane;
    __device__ __forceinline__ double &operator()(int k, int f) const { return s[(k * 7 + f) * kNX + lane]; }
};

__device__ __forceinline__ void wave_lds_sync() {   // one wave per workgroup: LDS operations complete in order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// all-reduce over lane bits 0..2 (the 8 lanes of a group)
__device__ __forceinline__ double allsum_group8(double v) {
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);   // row_half_mirror
    return v;
}
// all-reduce over lane bits 3..5 (same position in the 8 groups)
__device__ __forceinline__ double allsum_across8(double v) {
    v += dpp_mov<0x128>(v);   // row_ror:8
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    }
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    }
    return v;
}
// all 64 lanes (DPP + permlane swaps instead of six ds_bpermute rounds)
__device__ __forceinline__ double allsum_wave(double v) { return wave_uniform(allsum_across8(allsum_group8(v))); }   // scalar: see wave_uniform
template <int Q>
__device__ __forceinline__ double allsum_over_mc(double v) {   // Q = node parity: the column index is c (0) or r (1)
    if (Q == 0) return allsum_group8(v);
    return allsum_across8(v);
}

// The same reduction for TWO independent values, level by level side by side: each DPP / permlane move of one chain fills the
// wait states and the latency of the other's add.  The scheduling barriers keep the compiler from putting one butterfly after
// the other again (its scheduler minimises register pressure, not latency, in a kernel at the register limit).
#define SMC_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ void swap16_pair(double v, double &x, double &y) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    x = __hiloint2double((int)b[0], (int)a[0]);
    y = __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ void swap32_pair(double v, double &x, double &y) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    x = __hiloint2double((int)b[0], (int)a[0]);
    y = __hiloint2double((int)b[1], (int)a[1]);
}
template <int Q>
__device__ __forceinline__ void allsum_over_mc_pair(double &p, double &q) {
    SMC_SCHED_FENCE();
    if (Q == 0) {
        double tp = dpp_mov<0xB1>(p), tq = dpp_mov<0xB1>(q);
        p += tp; q += tq;
        SMC_SCHED_FENCE();
        tp = dpp_mov<0x4E>(p); tq = dpp_mov<0x4E>(q);
        p += tp; q += tq;
        SMC_SCHED_FENCE();
        tp = dpp_mov<0x141>(p); tq = dpp_mov<0x141>(q);
        p += tp; q += tq;
    } else {
        double tp = dpp_mov<0x128>(p), tq = dpp_mov<0x128>(q);
        p += tp; q += tq;
        SMC_SCHED_FENCE();
        double px, py, qx, qy;
        swap16_pair(p, px, py); swap16_pair(q, qx, qy);
        p = px + py; q = qx + qy;
        SMC_SCHED_FENCE();
        swap32_pair(p, px, py); swap32_pair(q, qx, qy);
        p = px + py; q = qx + qy;
    }
    SMC_SCHED_FENCE();
}

struct ElemLane {   // lane = 8 r + c
    int r, c, lane;
    __device__ __forceinline__ explicit ElemLane(int l) : r(l >> 3), c(l & 7), lane(l) {}
    template <int Q> __device__ __forceinline__ int mr() const { return Q ? c : r; }   // block row held for parity Q
    template <int Q> __device__ __forceinline__ int mc() const { return Q ? r : c; }   // block column
};
__device__ __forceinline__ int min6(int v) { return v < 6 ? v : 6; }
__device__ __forceinline__ double ipow_small(double x, int n) {   // x^n, 1 <= n <= kNewtonMaxIter
    double r = x;
    for (int q = 1; q < n; ++q) r *= x;
    return r;
}

// ---- control policy of the integrator (round 5; its CPU statement and the study behind it: the test checker's dae_policy,
// tools/k8_policy_study.py) ----------------------------------------------------------------------------------------------
// SMC_K8_POLICY 1 (default): IDA's policy for the iteration matrix and the Newton iteration, restated inside the
// quasi-constant-step BDF (the reference integrates with IDA: methanation_set_likelihood.py:167-198; SUNDIALS IDA,
// "Mathematical considerations", idaNls / idaNewtonIter / idaLsSolve):
//   * the factored iteration matrix is kept while cj / cj_at_evaluation stays inside ((1 - xrate) / (1 + xrate), its reciprocal),
//     xrate = SMC_K8_XRATE (IDA: 0.25; here 0.15, by measurement), and the Newton correction is scaled by 2 / (1 + cjratio); a Newton failure on a kept matrix repeats
//     the attempt with a fresh one;
//   * Newton converges when ss * |dy| <= 0.33 (|.| = RMS over all unknowns in tolerance units), ss = rate / (1 - rate) CARRIED
//     from step to step (20 after an evaluation of the matrix, 100 when cj changed since the previous attempt); the first
//     iteration also converges on |dy| <= 0.33e-4; rate = (|dy_m| / |dy_0|)^(1/m) > 0.9 ends the iteration as failed.
// IDA's step-size rule (double at a factor >= 2, hold below that) was studied as well and is NOT adopted: in this formulation it
// costs 30 % more steps (profiles/r05_k8_policy_study_cpu.txt).
// SMC_K8_POLICY 0: rounds 1-4 (SciPy's bdf.py: matrix kept only while c is unchanged, convergence rate from two iterations of
// the same step, newton_tol = 1e-3) - A/B builds.
#ifndef SMC_K8_POLICY
#define SMC_K8_POLICY 1
#endif
#ifndef SMC_K8_XRATE
#define SMC_K8_XRATE 0.15    // IDA's own window is 0.25; measured on one box (profiles/r05_ab_k8_policy.log): 2048 x 30 solves at
#endif                       // 0.25 -> 137.8 k solves/s, 0.15 -> 145.1 k, 0.10 -> 141.8 k (rounds 1-4's policy: 103.8 k)
constexpr double kEpcon = 0.33, kRateMax = 0.9, kSsAfterSetup = 20.0, kSsAfterCjChange = 100.0;
constexpr double kCjRatioLo = (1.0 - SMC_K8_XRATE) / (1.0 + SMC_K8_XRATE), kCjRatioHi = (1.0 + SMC_K8_XRATE) / (1.0 - SMC_K8_XRATE);
// must the matrix evaluated at c_lu be evaluated again for an attempt with c?  (cj = 1 / c: cjratio = c_lu / c)
__device__ __forceinline__ bool matrix_is_stale(double c, double c_lu) {
#if SMC_K8_POLICY
    const double cjratio = c_lu / c;
    return !(cjratio > kCjRatioLo && cjratio < kCjRatioHi);
#else
    return c != c_lu;
#endif
}
// scale of the Newton correction computed with the matrix of c_lu in an attempt with c (exactly 1 when c == c_lu)
__device__ __forceinline__ double correction_scale(double c, double c_lu) {
#if SMC_K8_POLICY
    return 2.0 / (1.0 + c_lu / c);
#else
    return 1.0;
#endif
}
// IDA's convergence test after Newton iteration kk (0-based) with correction norm dy_norm: 1 converged, -1 failed, 0 go on
__device__ __forceinline__ int newton_verdict_ida(int kk, double dy_norm, double &first, double &ss) {
    if (kk == 0) {
        first = dy_norm;
        if (dy_norm <= 1e-4 * kEpcon) return 1;
    } else {
        const double q = dy_norm / first;
        const double rate = (kk == 1) ? q : (kk == 2) ? sqrt(q) : cbrt(q);
        if (!(rate <= kRateMax)) return -1;
        ss = rate / (1.0 - rate);
    }
    return (ss * dy_norm <= kEpcon) ? 1 : 0;
}

// reciprocal of a wave-uniform value: v_rcp_f64 (2^-24) + one Newton step (2^-48: the iteration matrix of a
// modified Newton method needs no more)
__device__ __forceinline__ double recip1(double a) {
    double x = __builtin_amdgcn_rcp(a);
    x = fma(x, fma(-a, x, 1.0), x);
    return fma(x, fma(-a, x, 1.0), x);
}

// ... with ONE Newton step (2^-48): what the two-ended elimination's Gauss-Jordan uses - the explicit inverse of a modified
// Newton iteration's matrix steers convergence only, and two instructions per pivot and chain are 4 % of a factorisation
__device__ __forceinline__ double recip1_short(double a) {
    const double x = __builtin_amdgcn_rcp(a);
    return fma(x, fma(-a, x, 1.0), x);
}

// one node of the block elimination; Q = I & 1
template <int I>
__device__ __forceinline__ bool elem_factor_node(const ElemLane &L, const double *cf, double (&X)[kNX], double (&G)[kNX]) {
    constexpr int Q = I & 1;
    const int mr = L.template mr<Q>(), mc = L.template mc<Q>();
    const double *cfi = cf + I * kCfRow;
    double a = X[I];
    if (I > 0) {
        constexpr int IP = (I > 0) ? I - 1 : 0;
        const int kap = mr < 6 ? 6 : 5;
        const int srcK = Q ? (kap * 8 + L.r) : (L.c * 8 + kap);   // holder of G_{I-1}[kap][mc] (parity 1-Q)
        const double gT = __shfl(G[IP], L.c * 8 + L.r);            // G_{I-1}[mr][mc]
        const double gK = __shfl(G[IP], srcK);
        const double ld = cfi[min6(mr)], lx = cfi[8 + min6(mr)];
        a = fma(-lx, gK, fma(-ld, gT, a));
    }
    int ok = 1;
    const double rowsign = Q ? -1.0 : 1.0;
    // The next pivot element always takes the generic update, so its reciprocal starts from `gen` and runs
    // beside the selects and the pivot row / column exchange of the next round (two chains instead of one).
    double akk = lane_bcast(a, 0);
    SMC_UNROLL
    for (int kk = 0; kk < 7; ++kk) {
        if (!(fabs(akk) > 1e-300) || !(fabs(akk) < 1e300)) ok = 0;
        const double p = recip1(akk);
        const double u = __shfl(a, (L.lane & ~7) | kk);   // lane (r, k)
        const double v = __shfl(a, kk * 8 + L.c);         // lane (k, c)
        const double gen = fma(-(u * v), p, a);
        akk = lane_bcast(gen, kk < 6 ? 9 * kk + 9 : 0);
        const double ap = a * p * rowsign;
        const bool rk = L.r == kk, ck = L.c == kk;
        const double on_row = ck ? p : ap, off_row = ck ? -ap : gen;
        a = rk ? on_row : off_row;
    }
    X[I] = a;
    {
        const int src6 = Q ? (48 + L.c) : ((L.lane & ~7) | 6);   // holder of X_I[mr][6]
        const double tU = __shfl(a, src6);
        const double ud = cfi[16 + min6(mc)], u65 = cfi[23];
        G[I] = fma(a, ud, (mc == 5) ? tU * u65 : 0.0);
    }
    return ok != 0;
}

template <int I>
struct ElemFactorLoop {
    static __device__ __forceinline__ bool run(const ElemLane &L, const double *cf, double (&X)[kNX], double (&G)[kNX]) {
        const bool below = ElemFactorLoop<I - 1>::run(L, cf, X, G);
        return elem_factor_node<I>(L, cf, X, G) && below;
    }
};
template <>
struct ElemFactorLoop<-1> {
    static __device__ __forceinline__ bool run(const ElemLane &, const double *, double (&)[kNX], double (&)[kNX]) { return true; }
};

// ---------------------------------------------------------------------------------------------
// two-ended elimination: node pair (K, 50 - K) side by side; Q = K & 1 is the parity of both
// ---------------------------------------------------------------------------------------------
// Layout reminders (Q = node parity): lane (r, c) holds element [mr][mc] with (mr, mc) = Q ? (c, r) : (r, c); the factor of a
// NEIGHBOUR node (parity 1 - Q) element [a][b] sits in lane (1-Q ? (b, a) : (a, b)).  L_i = diag(ld) + column 6 (rows 0..5: lx) +
// L[6][5] (lx[6]);  U_i = diag(ud[0..5]) + U[6][5] (u65)  (cf row of the node: [0..6] ld, [8..14] lx, [16..21] ud, [23] u65).
// branch-free (bitwise &): a scalar branch per pivot would cut the basic block in which the two chains are interleaved
__device__ __forceinline__ int gj_pivot_ok(double v) { return (int)(fabs(v) > 1e-300) & (int)(fabs(v) < 1e300); }

template <int K>
__device__ __forceinline__ bool elem_factor_pair(const ElemLane &L, const double *cf, double (&X)[kNX], double (&G)[kNX]) {
    constexpr int T = K, B = kNX - 1 - K, Q = K & 1;
    const int mr = L.template mr<Q>(), mc = L.template mc<Q>();
    const double *cT = cf + T * kCfRow, *cB = cf + B * kCfRow;
    double aT = X[T], aB = X[B];
    if (K > 0) {
        constexpr int TP = (K > 0) ? T - 1 : 0, BP = (K > 0) ? B + 1 : kNX - 1;
        const int srcT = L.c * 8 + L.r;                              // neighbour's [mr][mc]
        {   // top:  D' = D - L_T G_{T-1}:  (L G)[mr][mc] = ld[mr] G[mr][mc] + lx[mr] G[mr < 6 ? 6 : 5][mc]
            const int kap = mr < 6 ? 6 : 5;
            const int srcK = Q ? (kap * 8 + L.r) : (L.c * 8 + kap);
            const double gT = __shfl(G[TP], srcT), gK = __shfl(G[TP], srcK);
            aT = fma(-cT[8 + min6(mr)], gK, fma(-cT[min6(mr)], gT, aT));
        }
        {   // bottom:  D'' = D - U_B H_{B+1}:  (U H)[mr][mc] = ud[mr] H[mr][mc] (mr < 6),  u65 H[5][mc] (mr == 6)
            const int src5 = Q ? (5 * 8 + L.r) : (L.c * 8 + 5);
            const double hT = __shfl(G[BP], srcT), h5 = __shfl(G[BP], src5);
            const double cu = cB[(mr == 6) ? 23 : 16 + min6(mr)];         // ONE load (padding row 7: cB[22] == 0)
            aB = fma(-cu, (mr == 6) ? h5 : hT, aB);
        }
    }
    // two Gauss-Jordan inversions, pivot by pivot side by side (see elem_factor_node for the single form).  No test per pivot
    // (two compares and the scalar bookkeeping per pivot and chain were a fifth of the loop): a vanishing pivot shows in the
    // inverse - its reciprocal sits on the diagonal, and 0 or a non-finite pivot turns the block into inf / NaN - and is caught
    // by one magnitude test per element at the end.
    const double rowsign = Q ? -1.0 : 1.0;
    double akT = lane_bcast(aT, 0), akB = lane_bcast(aB, 0);
    SMC_UNROLL
    for (int kk = 0; kk < 7; ++kk) {
        const double pT = recip1_short(akT), pB = recip1_short(akB);
        const int su = (L.lane & ~7) | kk, sv = kk * 8 + L.c;
        const double uT = __shfl(aT, su), uB = __shfl(aB, su);
        const double vT = __shfl(aT, sv), vB = __shfl(aB, sv);
        const double genT = fma(-(uT * vT), pT, aT), genB = fma(-(uB * vB), pB, aB);
        akT = lane_bcast(genT, kk < 6 ? 9 * kk + 9 : 0);
        akB = lane_bcast(genB, kk < 6 ? 9 * kk + 9 : 0);
        const double apT = aT * pT * rowsign, apB = aB * pB * rowsign;
        const bool rk = L.r == kk, ck = L.c == kk;
        aT = rk ? (ck ? pT : apT) : (ck ? -apT : genT);
        aB = rk ? (ck ? pB : apB) : (ck ? -apB : genB);
    }
    X[T] = aT;
    X[B] = aB;
    const int ok = (int)(fabs(aT) < 1e300) & (int)(fabs(aB) < 1e300);      // false for NaN
    const int src6 = Q ? (48 + L.c) : ((L.lane & ~7) | 6);          // holder of X[mr][6]
    {   // G_T = X_T U_T:  [mr][mc] = X[mr][mc] ud[mc] + (mc == 5) X[mr][6] u65
        const double tU = __shfl(aT, src6), ud = cT[16 + min6(mc)], u65 = cT[23];     // loads outside the selects: no exec-mask branches
        const double extra = tU * u65;
        G[T] = fma(aT, ud, (mc == 5) ? extra : 0.0);
    }
    {   // H_B = X_B L_B:  [mr][mc] = X[mr][mc] ld[mc] + (mc == 6) sum_{k<6} X[mr][k] lx[k] + (mc == 5) X[mr][6] lx[6]
        const double x6 = __shfl(aB, src6), lxv = cB[8 + min6(mc)], ldv = cB[min6(mc)], lx6 = cB[8 + 6];
        const double pr = aB * lxv, e5 = x6 * lx6;
        const double S = allsum_over_mc<Q>((mc < 6) ? pr : 0.0);
        G[B] = fma(aB, ldv, (mc == 6) ? S : ((mc == 5) ? e5 : 0.0));
    }
    return ok != 0;
}
template <int K>
struct ElemFactorPairLoop {
    static __device__ __forceinline__ bool run(const ElemLane &L, const double *cf, double (&X)[kNX], double (&G)[kNX]) {
        const bool before = ElemFactorPairLoop<K - 1>::run(L, cf, X, G);
        return elem_factor_pair<K>(L, cf, X, G) && before;
    }
};
template <>
struct ElemFactorPairLoop<-1> {
    static __device__ __forceinline__ bool run(const ElemLane &, const double *, double (&)[kNX], double (&)[kNX]) { return true; }
};
// the node where the chains meet:  D* = D - L G_{m-1} - U H_{m+1},  X_m = D*^{-1}  (no coupling factor)
__device__ __forceinline__ bool elem_factor_middle(const ElemLane &L, const double *cf, double (&X)[kNX], const double (&G)[kNX]) {
    constexpr int I = kMid, Q = I & 1;
    const int mr = L.template mr<Q>();
    const double *cfi = cf + I * kCfRow;
    double a = X[I];
    {
        const int srcT = L.c * 8 + L.r;
        const int kap = mr < 6 ? 6 : 5;
        const int srcK = Q ? (kap * 8 + L.r) : (L.c * 8 + kap), src5 = Q ? (5 * 8 + L.r) : (L.c * 8 + 5);
        const double gT = __shfl(G[I - 1], srcT), gK = __shfl(G[I - 1], srcK);
        const double hT = __shfl(G[I + 1], srcT), h5 = __shfl(G[I + 1], src5);
        a = fma(-cfi[8 + min6(mr)], gK, fma(-cfi[min6(mr)], gT, a));
        const double cu = cfi[(mr == 6) ? 23 : 16 + min6(mr)];
        a = fma(-cu, (mr == 6) ? h5 : hT, a);
    }
    int ok = 1;
    const double rowsign = Q ? -1.0 : 1.0;
    double akk = lane_bcast(a, 0);
    SMC_UNROLL
    for (int kk = 0; kk < 7; ++kk) {
        ok &= gj_pivot_ok(akk);
        const double p = recip1(akk);
        const double u = __shfl(a, (L.lane & ~7) | kk), v = __shfl(a, kk * 8 + L.c);
        const double gen = fma(-(u * v), p, a);
        akk = lane_bcast(gen, kk < 6 ? 9 * kk + 9 : 0);
        const double ap = a * p * rowsign;
        const bool rk = L.r == kk, ck = L.c == kk;
        a = rk ? (ck ? p : ap) : (ck ? -ap : gen);
    }
    X[I] = a;
    return ok != 0;
}

// Right-hand side inwards on both chains.  zprev / wprev: results of the previous pair on this lane's input index mc; the
// operands of the NEXT pair are loaded before this pair's results are stored (the compiler must assume the stores alias them).
//   top:     t = b - ld z_prev - lxa z_prev[6] - lxb z_prev[5]      lxa = mc < 6 ? L[mc][6] : 0,  lxb = mc >= 6 ? L[6][5] : 0
//   bottom:  t = b - cua w_prev - cub w_prev[5]                      cua = mc < 6 ? U[mc][mc] : 0, cub = mc >= 6 ? U[6][5] : 0
// The masks live in the per-lane ADDRESSES of the coefficient loads (a zero slot of the node's cf row where the coefficient
// does not apply: FwdOffsets), so the broadcasts stay scalar operands of the FMAs - no v_mov from the SGPRs, no selects.
struct FwdOffsets {     // per lane and parity: offsets into a node's cf row of 24 doubles ([7], [15], [22] hold zeros)
    int ld, lxa, lxb, cua, cub, b;
};
template <int Q>
__device__ __forceinline__ FwdOffsets fwd_offsets(const ElemLane &L) {
    const int mc = L.template mc<Q>(), m6 = min6(mc);
    FwdOffsets o;
    o.ld = m6;
    o.lxa = (mc < 6) ? 8 + mc : 7;
    o.lxb = (mc < 6) ? 7 : 8 + 6;
    o.cua = 16 + m6;                  // [22] == 0 for mc >= 6
    o.cub = (mc < 6) ? 22 : 23;
    o.b = m6;
    return o;
}
struct FwdOperands {    // of one pair: loaded one pair ahead
    double bT, ldT, lxaT, lxbT, bB, cuaB, cubB;
};
template <int K>
struct ElemForwardPair {
    static __device__ __forceinline__ void run(const ElemLane &L, const double *cf, const double *b, double *z, const double (&X)[kNX],
                                               const FwdOffsets &o0, const FwdOffsets &o1, double zprev, double wprev,
                                               const FwdOperands &op, double &zlast, double &wlast) {
        constexpr int T = K, B = kNX - 1 - K, Q = K & 1, TN = T + 1, BN = B - 1;
        const int mr = L.template mr<Q>(), mc = L.template mc<Q>();
        const FwdOffsets &on = Q ? o0 : o1;                  // offsets of the next pair's parity
        FwdOperands nx;
        nx.bT = b[TN * 7 + on.b];
        nx.ldT = cf[TN * kCfRow + on.ld];
        nx.lxaT = cf[TN * kCfRow + on.lxa];
        nx.lxbT = cf[TN * kCfRow + on.lxb];
        nx.bB = b[BN * 7 + on.b];
        nx.cuaB = cf[BN * kCfRow + on.cua];
        nx.cubB = cf[BN * kCfRow + on.cub];
        double tT = op.bT, tB = op.bB;
        if (K > 0) {
            const double z6 = lane_bcast(zprev, Q ? 48 : 6), z5 = lane_bcast(zprev, Q ? 40 : 5);
            const double w5 = lane_bcast(wprev, Q ? 40 : 5);
            tT = fma(-op.lxbT, z5, fma(-op.lxaT, z6, fma(-op.ldT, zprev, tT)));
            tB = fma(-op.cubB, w5, fma(-op.cuaB, wprev, tB));
        }
        double zi = X[T] * tT, wi = X[B] * tB;
        allsum_over_mc_pair<Q>(zi, wi);
        const int slot = (mc == 0 && mr < 7) ? mr : 7;       // one unconditional store each: no exec-mask round trip per step
        z[T * kZRow + slot] = zi;
        z[B * kZRow + slot] = wi;
        if constexpr (K + 1 < kMid) {
            ElemForwardPair<K + 1>::run(L, cf, b, z, X, o0, o1, zi, wi, nx, zlast, wlast);
        } else {
            zlast = zi;
            wlast = wi;
        }
    }
};
// Solution outwards on both chains, K = 24 .. 0:  x_T = z_T - G_T x_{T+1},  x_B = w_B - H_B x_{B-1}  (H is kept in G[B]).
// xT / xB: the inner neighbours' solutions on this lane's input index mc; zT / wB: this pair's z / w on the lane's row index mr.
template <int K>
struct ElemBackwardPair {
    static __device__ __forceinline__ void run(const ElemLane &L, double *z, const double (&G)[kNX], double xT, double zT, double xB,
                                               double wB) {
        constexpr int T = K, B = kNX - 1 - K, Q = K & 1, QN = 1 - Q, KN = (K > 0) ? K - 1 : 0;
        const int mr = L.template mr<Q>(), mc = L.template mc<Q>();
        const int mrn = min6(L.template mr<QN>());
        const double zTn = z[KN * kZRow + mrn], wBn = z[(kNX - 1 - KN) * kZRow + mrn];
        double st = G[T] * xT, sb = G[B] * xB;
        allsum_over_mc_pair<Q>(st, sb);
        const double xt = zT - st, xb = wB - sb;
        const int slot = (mc == 0 && mr < 7) ? mr : 7;
        z[T * kZRow + slot] = xt;
        z[B * kZRow + slot] = xb;
        if constexpr (K > 0) ElemBackwardPair<KN>::run(L, z, G, xt, zTn, xb, wBn);
    }
};
// one linear solve with the two-ended factors: b (LDS, node-major) -> x in z (LDS, [node][8])
__device__ __forceinline__ void elem_solve_twisted(const ElemLane &L, const double *cf, const double *b, double *z,
                                                   const double (&X)[kNX], const double (&G)[kNX]) {
    const FwdOffsets o0 = fwd_offsets<0>(L), o1 = fwd_offsets<1>(L);
    double zl, wl;
    {
        FwdOperands op{};
        op.bT = b[o0.b];
        op.bB = b[(kNX - 1) * 7 + o0.b];
        ElemForwardPair<0>::run(L, cf, b, z, X, o0, o1, 0.0, 0.0, op, zl, wl);
    }
    constexpr int I = kMid, Q = I & 1;
    static_assert(Q == 1, "the middle node's offsets are o1");
    const int mr = L.template mr<Q>(), mc = L.template mc<Q>();
    const double *cfi = cf + I * kCfRow;
    const double z6 = lane_bcast(zl, Q ? 48 : 6), z5 = lane_bcast(zl, Q ? 40 : 5), w5 = lane_bcast(wl, Q ? 40 : 5);
    double t = b[I * 7 + o1.b];
    t = fma(-cfi[o1.lxb], z5, fma(-cfi[o1.lxa], z6, fma(-cfi[o1.ld], zl, t)));
    t = fma(-cfi[o1.cub], w5, fma(-cfi[o1.cua], wl, t));
    const int mrn = min6(L.template mr<1 - Q>());
    const double zT = z[(I - 1) * kZRow + mrn], wB = z[(I + 1) * kZRow + mrn];      // loaded before x_m is stored
    const double xm = allsum_over_mc<Q>(X[I] * t);
    z[I * kZRow + ((mc == 0 && mr < 7) ? mr : 7)] = xm;
    ElemBackwardPair<kMid - 1>::run(L, z, G, xm, zT, xm, wB);
}

// iteration matrix at the predictor (parallel over nodes), transposition into the element layout, block elimination
__device__ __forceinline__ bool elem_build_and_factor(int lane, double *lds, const double *yp, const double *psi,
                                                      const double *p, double c, double (&X)[kNX], double (&G)[kNX], DaeStats &st) {
    SMC_PROF_BEGIN();
    const double cj = 1.0 / c;
    const bool node = lane < kNX;
    const ElemLane L(lane);
    double *cf = lds + kLdsCf, *stage = lds + kLdsB;
    {
        double wm[7], wp[7], yd0[7], res[7], Lb[kNB], Db[kNB], Ub[kNB];
        neighbours(yp, wm, wp);
        SMC_UNROLL
        for (int f = 0; f < 7; ++f) yd0[f] = psi[f] * cj;
        SMC_UNROLL
        for (int q = 0; q < kNB; ++q) Lb[q] = Db[q] = Ub[q] = 0.0;
        if (node) node_eval<true>(lane, wm, yp, wp, yd0, p, cj, res, Lb, Db, Ub);
        SMC_UNROLL
        for (int i = 0; i < kNX; ++i) X[i] = 0.0;
        if (node) {
            double *o = cf + lane * kCfRow;
            SMC_UNROLL
            for (int r = 0; r < 7; ++r) {
                o[r] = Lb[r * 7 + r];
                o[8 + r] = (r < 6) ? Lb[r * 7 + 6] : Lb[6 * 7 + 5];
                o[16 + r] = (r < 6) ? Ub[r * 7 + r] : 0.0;
            }
            o[7] = 0.0;
            o[15] = 0.0;
            o[23] = Ub[6 * 7 + 5];
        }
        SMC_UNROLL
        for (int rho = 0; rho < 7; ++rho) {   // one block row of all nodes per pass through the staging row
            if (node)
                SMC_UNROLL
                for (int cc = 0; cc < 7; ++cc) stage[lane * 7 + cc] = Db[rho * 7 + cc];
            wave_lds_sync();
            if (L.r == rho && L.c < 7)        // even nodes: lane (r, c) holds [r][c]
                SMC_UNROLL
                for (int i = 0; i < kNX; i += 2) X[i] = stage[i * 7 + L.c];
            if (L.c == rho && L.r < 7)        // odd nodes: lane (r, c) holds [c][r]
                SMC_UNROLL
                for (int i = 1; i < kNX; i += 2) X[i] = stage[i * 7 + L.r];
            wave_lds_sync();
        }
    }
    SMC_PROF_ADD(st, 7);   // Jacobian blocks + transposition
#if SMC_K8_TWISTED
    bool ok = ElemFactorPairLoop<kMid - 1>::run(L, cf, X, G);
    ok = elem_factor_middle(L, cf, X, G) && ok;
#else
    const bool ok = ElemFactorLoop<kNX - 1>::run(L, cf, X, G);
#endif
    SMC_PROF_ADD(st, 0);   // block elimination
    return __all(ok);
}

// Forward scan, tail-recursive over the nodes: z_I = X_I (b_I - L_I z_{I-1}).  The right-hand side and the L
// coefficients of node I+1 are loaded BEFORE z_I is stored (the compiler must assume the store aliases them, and an
// LDS round trip on the chain costs as much as the butterfly).  Returns z_50[mr] on every lane.
template <int I>
struct ElemForward {
    static __device__ __forceinline__ double run(const ElemLane &L, const double *cf, const double *b, double *z,
                                                 const double (&X)[kNX], double zprev, double bI, double ldI, double lxI) {
        constexpr int Q = I & 1, QN = 1 - Q, IN = (I + 1 < kNX) ? I + 1 : I;
        const int mr = L.template mr<Q>(), mc = L.template mc<Q>();
        const int mcn = min6(L.template mc<QN>());
        const double bN = b[IN * 7 + mcn], ldN = cf[IN * kCfRow + mcn], lxN = cf[IN * kCfRow + 8 + mcn];
        double t = bI;
        if (I > 0) {
            const double z6 = lane_bcast(zprev, Q ? 48 : 6), z5 = lane_bcast(zprev, Q ? 40 : 5);
            const double zx = (mc == 6) ? z5 : z6;
            t = fma(-lxI, zx, fma(-ldI, zprev, t));
        }
        const double zi = allsum_over_mc<Q>(X[I] * t);
        z[I * kZRow + ((mc == 0 && mr < 7) ? mr : 7)] = zi;   // one unconditional store: no exec-mask round trip per step
        if constexpr (I + 1 < kNX) return ElemForward<IN>::run(L, cf, b, z, X, zi, bN, ldN, lxN);
        else return zi;
    }
};

// Backward scan: x_I = z_I - G_I x_{I+1}, I = 49 .. 0; `xnext` is x_{I+1}[mc], zI = z_I[mr] (loaded one step ahead)
template <int I>
struct ElemBackward {
    static __device__ __forceinline__ void run(const ElemLane &L, double *z, const double (&G)[kNX], double xnext, double zI) {
        constexpr int Q = I & 1, QN = 1 - Q, IN = (I > 0) ? I - 1 : 0;
        const int mr = L.template mr<Q>(), mc = L.template mc<Q>();
        const double zN = z[IN * kZRow + min6(L.template mr<QN>())];
        const double xi = zI - allsum_over_mc<Q>(G[I] * xnext);
        z[I * kZRow + ((mc == 0 && mr < 7) ? mr : 7)] = xi;
        if constexpr (I > 0) ElemBackward<IN>::run(L, z, G, xi, zN);
    }
};

// one modified-Newton iteration; returns RMS(dy/scale) over all unknowns, or -1 if the residual is not finite
__device__ __forceinline__ double elem_newton_iteration(int lane, double *lds, double *y, double *dd, const double *yp,
                                                        const double *psi, const double *p, double c, double corr, double rtol,
                                                        double atol, const double (&X)[kNX], const double (&G)[kNX],
                                                        DaeStats &st) {
    SMC_PROF_BEGIN();
    const double cj = 1.0 / c;
    const bool node = lane < kNX;
    const ElemLane L(lane);
    double *b = lds + kLdsB, *z = lds + kLdsZ;
    const double *cf = lds + kLdsCf;
    int finite = 1;
    {
        double wm[7], wp[7], yd0[7], res[7];
        neighbours(y, wm, wp);
        SMC_UNROLL
        for (int f = 0; f < 7; ++f) yd0[f] = (psi[f] + dd[f]) * cj;
        if (node) {
            node_eval<false>(lane, wm, y, wp, yd0, p, cj, res, nullptr, nullptr, nullptr);
            SMC_UNROLL
            for (int r = 0; r < 7; ++r) {
                if (!(res[r] - res[r] == 0.0)) finite = 0;
                b[lane * 7 + r] = -res[r];
            }
        }
    }
    if (!__all(finite)) return -1.0;
    wave_lds_sync();
    SMC_PROF_ADD(st, 1);
#if SMC_K8_TWISTED
    elem_solve_twisted(L, cf, b, z, X, G);
    SMC_PROF_ADD(st, 2);   // both scans (slot 3 stays empty)
#else
    const double zlast = ElemForward<0>::run(L, cf, b, z, X, 0.0, b[min6(L.c)], 0.0, 0.0);
    SMC_PROF_ADD(st, 2);
    ElemBackward<kNX - 2>::run(L, z, G, zlast, z[(kNX - 2) * kZRow + min6(L.template mr<(kNX - 2) & 1>())]);
#endif
    wave_lds_sync();
    SMC_PROF_ADD(st, 3);
    double sumsq = 0.0;
    if (node)
        SMC_UNROLL
        for (int f = 0; f < 7; ++f) {
#if SMC_K8_POLICY
            const double dx = z[lane * kZRow + f] * corr;      // matrix of another cj: 2 / (1 + cjratio)
#else
            const double dx = z[lane * kZRow + f];
#endif
            const double sc = atol + rtol * fabs(yp[f]);
            // weights of a convergence norm: the reciprocal with two Newton steps (~1 ulp) instead of the IEEE division's
            // scaling / fix-up sequence (13 instructions per component and iteration; sc is within [atol, atol + rtol |y|])
            const double q = dx * recip1(sc);
            sumsq += q * q;
            y[f] += dx;
            dd[f] += dx;
        }
    wave_lds_sync();
    return sqrt(allsum_wave(sumsq) / kNS);
}

// bdf.py compute_R / change_D with static loops (no scratch): D[0..order] <- (R(factor) U)^T D[0..order].
// U = R(1) is upper triangular, so the 6 x 6 product does not depend on the order; rows above it are masked.
__device__ __forceinline__ void elem_change_D(const DViewE &D, int order, double factor, bool node) {
    double R[6][6], RU[6][6];
    SMC_UNROLL
    for (int j = 0; j < 6; ++j) R[0][j] = 1.0;
    SMC_UNROLL
    for (int i = 1; i < 6; ++i) {
        R[i][0] = 0.0;
        SMC_UNROLL
        for (int j = 1; j < 6; ++j) R[i][j] = R[i - 1][j] * ((i - 1 - factor * j) * (1.0 / i));
    }
    SMC_UNROLL
    for (int i = 0; i < 6; ++i)
        SMC_UNROLL
        for (int j = 0; j < 6; ++j) {
            double s = 0.0;
            SMC_UNROLL
            for (int q = 0; q <= j; ++q) {
                double u = (q == 0) ? 1.0 : 0.0;   // U[q][j] = prod_{m=1..q} (m-1-j)/m for j >= 1; U[q][0] = delta_q0
                if (j >= 1) {
                    u = 1.0;
                    SMC_UNROLL
                    for (int m = 1; m <= q; ++m) u *= (double)(m - 1 - j) / (double)m;
                }
                s += R[i][q] * u;
            }
            RU[i][j] = (i <= order) ? s : 0.0;
        }
    if (!node) return;
    SMC_UNROLL
    for (int f = 0; f < 7; ++f) {
        double dcol[6];
        SMC_UNROLL
        for (int i = 0; i < 6; ++i) dcol[i] = D(i, f);
        SMC_UNROLL
        for (int j = 0; j < 6; ++j) {
            double s = 0.0;
            SMC_UNROLL
            for (int i = 0; i < 6; ++i) s += RU[i][j] * dcol[i];
            if (j <= order) D(j, f) = s;
        }
    }
}

// Integrate one solve (the whole wave cooperates).  lds: this wave's region of kLdsDoubles doubles, holding y0 in
// row 0 of the differences array and zeros in rows 1..7 on entry; the state at tf is left in row 0.
__device__ __forceinline__ void dae_elem_integrate(double *lds, int lane, const double *p, double tf, double rtol,
                                                   double atol, double h0, int max_attempts, DaeStats &st) {
    const double newton_tol = fmax(10 * 2.220446049250313e-16 / rtol, fmin(0.03, sqrt(rtol)));   // SMC_K8_POLICY 0 only
    (void)newton_tol;
    const bool node = lane < kNX;
    const DViewE D{lds + kLdsD, lane};
    st.steps = st.rejects = st.newton_fail = st.nlu = st.newton_iters = 0;
    st.status = 0;
#ifdef SMC_METH_PROFILE
    for (int q = 0; q < 12; ++q) st.prof[q] = 0;
    const long long prof_start_ = clock64();
#endif
    double t = 0.0, h_abs = h0;
    int order = 1, n_equal = 0, attempts = 0;
    double X[kNX], G[kNX];
    bool lu_valid = false, force_rebuild = false;
    double c_lu = 0.0;
    double ss = kSsAfterSetup, c_last = 0.0;   // SMC_K8_POLICY 1: the carried convergence-rate factor, c of the previous attempt
    double yp[7], y[7], psi[7], dd[7];
    for (;;) {  // one iteration = one step attempt
        // The state that steers the attempt, pinned to scalars (meth_dae_wave.h: wave_uniform): the values are equal in all
        // lanes anyway; this tells the compiler, so that every branch below is a scalar branch taken by the whole wave.
        t = wave_uniform(t);
        h_abs = wave_uniform(h_abs);
        c_lu = wave_uniform(c_lu);
#if SMC_K8_POLICY
        ss = wave_uniform(ss);
        c_last = wave_uniform(c_last);
#endif
        order = __builtin_amdgcn_readfirstlane(order);
        n_equal = __builtin_amdgcn_readfirstlane(n_equal);
        attempts = __builtin_amdgcn_readfirstlane(attempts);
        lu_valid = __builtin_amdgcn_readfirstlane((int)lu_valid) != 0;
        force_rebuild = __builtin_amdgcn_readfirstlane((int)force_rebuild) != 0;
        if (!(t < tf)) break;
        if (h_abs < 1e-14 * fmax(1.0, t) || attempts >= max_attempts) { st.status = 1; break; }
        ++attempts;
        double t_new = t + h_abs;
        if (t_new - tf > 0) {
            t_new = tf;
            { SMC_PROF_BEGIN(); elem_change_D(D, order, fabs(t_new - t) / h_abs, node); SMC_PROF_ADD(st, 5); }
            n_equal = 0;
        }
        t_new = wave_uniform(t_new);   // elem_change_D branches on the lane: what joins behind it is re-pinned
        n_equal = __builtin_amdgcn_readfirstlane(n_equal);
        const double h = t_new - t;
        h_abs = fabs(h);
        const double c = h / bdf_alpha(order);
        SMC_PROF_BEGIN();
        {
            double s[7], q[7];
            SMC_UNROLL
            for (int f = 0; f < 7; ++f) s[f] = q[f] = 0.0;
            if (node) {
                SMC_UNROLL
                for (int kk = 0; kk <= kMaxOrder; ++kk)
                    if (kk <= order)
                        SMC_UNROLL
                        for (int f = 0; f < 7; ++f) {
                            const double dv = D(kk, f);
                            s[f] += dv;
                            if (kk >= 1) q[f] += dv * bdf_gamma(kk);
                        }
            }
            const double inv_alpha = 1.0 / bdf_alpha(order);
            SMC_UNROLL
            for (int f = 0; f < 7; ++f) {
                yp[f] = y[f] = s[f];
                psi[f] = q[f] * inv_alpha;
                dd[f] = 0.0;
            }
        }
        // The factored iteration matrix is kept while it is not stale (matrix_is_stale: policy 1 - cj within (0.6, 1.67) of the
        // cj it was evaluated with; policy 0 - c unchanged, bdf.py:343-357); rebuilt at the current predictor otherwise, or - by
        // repeating this attempt - when Newton failed on a kept matrix.
        // (Rounds 2-4 tried the pieces of IDA's policy one at a time inside SciPy's control and dropped each: keeping the matrix
        // while c drifts cut the factorisations but raised SciPy-test Newton iterations 720 -> 913; the carried rate alone cut
        // the iterations 720 -> 435.  Together, with IDA's own convergence constant, they pay: CPU study, then the GPU A/B
        // in profiles/r05_ab_k8_policy.log.)
        const bool fresh = !lu_valid || force_rebuild || matrix_is_stale(c, c_lu);
#if SMC_K8_POLICY
        if (c != c_last) ss = kSsAfterCjChange;
        c_last = c;
#endif
        SMC_PROF_ADD(st, 6);
        if (fresh) {
            ++st.nlu;
            lu_valid = elem_build_and_factor(lane, lds, yp, psi, p, c, X, G, st);
            c_lu = c;
            force_rebuild = false;
            ss = kSsAfterSetup;
        }
        bool converged = false;
        int n_iter = 0;
        if (lu_valid) {
            const double corr = wave_uniform(correction_scale(c, c_lu));
#if SMC_K8_POLICY
            double dy_first = 0.0;
#pragma unroll 1
            for (int kk = 0; kk < kNewtonMaxIter; ++kk) {
                const double dy_norm = elem_newton_iteration(lane, lds, y, dd, yp, psi, p, c, corr, rtol, atol, X, G, st);
                n_iter = kk + 1;
                ++st.newton_iters;
                if (dy_norm < 0) break;
                const int verdict = newton_verdict_ida(kk, dy_norm, dy_first, ss);
                if (verdict != 0) { converged = verdict > 0; break; }
            }
#else
            // bdf.py:365-382
            double dy_norm_old = -1.0;
#pragma unroll 1
            for (int kk = 0; kk < kNewtonMaxIter; ++kk) {
                const double dy_norm = elem_newton_iteration(lane, lds, y, dd, yp, psi, p, c, corr, rtol, atol, X, G, st);
                n_iter = kk + 1;
                ++st.newton_iters;
                if (dy_norm < 0) break;
                const double rate = (dy_norm_old >= 0) ? dy_norm / dy_norm_old : -1.0;
                const double scaled = dy_norm / (1 - rate);      // dy_norm / (1 - rate): shared by both tests (one division, not two)
                if (rate >= 0 && (rate >= 1 || ipow_small(rate, kNewtonMaxIter - kk) * scaled > newton_tol)) break;
                if (dy_norm == 0 || (rate >= 0 && rate * scaled < newton_tol)) { converged = true; break; }
                dy_norm_old = dy_norm;
            }
#endif
        }
        SMC_PROF_ADD(st, 8);   // factorisation + Newton loop incl. control (slots 0,7,1,2,3 are inside)
        if (!converged && !fresh) {   // stale matrix: same step again with a fresh one
            force_rebuild = true;
            continue;
        }
        if (!converged) {
            ++st.newton_fail;
            lu_valid = false;
            h_abs *= 0.5;
            { SMC_PROF_BEGIN(); elem_change_D(D, order, 0.5, node); SMC_PROF_ADD(st, 5); }
            n_equal = 0;
            continue;
        }
        const double safety = 0.9 * (2 * kNewtonMaxIter + 1) / (2.0 * kNewtonMaxIter + n_iter);
        double se = 0.0;
        if (node)
            SMC_UNROLL
            for (int f = 0; f < 6; ++f) {
                const double isc = recip1(atol + rtol * fabs(y[f]));      // weights of an error norm: see elem_newton_iteration
                const double e = bdf_error_const(order) * dd[f] * isc;
                se += e * e;
            }
        const double error_norm = sqrt(allsum_wave(se) / (6 * kNX));
        if (!(error_norm <= 1)) {
            ++st.rejects;
            const double factor = (error_norm == error_norm) ? fmax(0.2, safety * pow(error_norm, -1.0 / (order + 1))) : 0.2;
            h_abs *= factor;
            { SMC_PROF_BEGIN(); elem_change_D(D, order, factor, node); SMC_PROF_ADD(st, 5); }
            n_equal = 0;
            continue;
        }
        SMC_PROF_ADD(st, 9);   // error test
        ++n_equal;
        t = t_new;
        ++st.steps;
        const bool select = n_equal >= order + 1;
        double sm = 0.0, sp = 0.0;
        if (node) {
            double acc[7], d_order[7], dnew2[7];
            SMC_UNROLL
            for (int f = 0; f < 7; ++f) {
                dnew2[f] = dd[f] - D(order + 1, f);
                D(order + 2, f) = dnew2[f];
                D(order + 1, f) = dd[f];
                acc[f] = dd[f];
                d_order[f] = 0.0;
            }
            SMC_UNROLL
            for (int kk = kMaxOrder; kk >= 0; --kk)
                if (kk <= order)
                    SMC_UNROLL
                    for (int f = 0; f < 7; ++f) {
                        acc[f] += D(kk, f);
                        D(kk, f) = acc[f];
                        if (kk == order) d_order[f] = acc[f];
                    }
            if (select)
                SMC_UNROLL
                for (int f = 0; f < 6; ++f) {
                    const double isc = recip1(atol + rtol * fabs(y[f]));
                    if (order > 1) { const double e = bdf_error_const(order - 1) * d_order[f] * isc; sm += e * e; }
                    if (order < kMaxOrder) { const double e = bdf_error_const(order + 1) * dnew2[f] * isc; sp += e * e; }
                }
        }
        SMC_PROF_ADD(st, 10);  // D update + order-selection norms
        if (!select) continue;
        const double inf = __longlong_as_double(0x7ff0000000000000LL);
        const double em_s = sqrt(allsum_wave(sm) / (6 * kNX)), ep_s = sqrt(allsum_wave(sp) / (6 * kNX));   // both sums: no
        const double em = (order > 1) ? em_s : inf;                                                           // branch around
        const double ep = (order < kMaxOrder) ? ep_s : inf;                                                   // a wave exchange
        const double fm = pow(em, -1.0 / order), f0 = pow(error_norm, -1.0 / (order + 1)), fp = pow(ep, -1.0 / (order + 2));
        double best = fm;
        int delta = -1;
        if (f0 > best) { best = f0; delta = 0; }
        if (fp > best) { best = fp; delta = 1; }
        order += delta;
        const double factor = fmin(10.0, safety * best);
        h_abs *= factor;
        { SMC_PROF_BEGIN(); elem_change_D(D, order, factor, node); SMC_PROF_ADD(st, 5); }
        n_equal = 0;
    }
    st.status = __builtin_amdgcn_readfirstlane(st.status);
#ifdef SMC_METH_PROFILE
    st.prof[4] = clock64() - prof_start_;
#endif
}

}  // namespace meth
}  // namespace smc
