// meth_dae_split.h -- K8 v4: one solve per WORKGROUP OF TWO WAVES; the two elimination chains get a wave each.
//
// Why.  v3 (meth_dae_elem.h) keeps the block factors of all 51 nodes in one wave's registers: 2 x 51 doubles per lane = 204 of its
// 472 VGPRs, so one wave per SIMD is all that fits, and that wave executes vector instructions in 50 % of its cycles
// (profiles/r04_k8_pmc_sq_summary.json): a dependent FP64 instruction waits for its predecessor and nothing else is there to
// issue.  The two-ended elimination of round 4 put two chains into the one instruction stream; they cost 1.5 - 1.7 x a single
// chain, i.e. the stream is then mostly issue-bound - the second chain is not free.  Here the two chains get a wave each:
//   wave 0 ("main"):   the integrator of v3 - predictor, residuals, norms, error tests over all 51 nodes (lane = node) - and the
//                      downward chain 0 -> 24 with the middle node 25;
//   wave 1 ("server"): the upward chain 50 -> 26, on wave 0's commands (FACTOR: Jacobian blocks of nodes 26 .. 50 from the
//                      differences array, transposition, elimination;  SOLVE: inward and outward scan of one right-hand side).
// Each wave holds HALF of the factors (26 + 25 doubles per lane), which brings the kernel to 256 VGPRs: two waves per SIMD, eight
// per CU, with the same four solves per CU as before (the LDS of a solve is shared by its two waves).
//   * chain position k is node k in wave 0 and node 50 - k in wave 1: same parity, same lane layout, and - with the coefficient
//     masks in the per-lane LDS ADDRESSES (SplitOffsets) - the SAME instruction stream: downwards the coupling is
//     L_i G_{i-1}, upwards U_i H_{i+1}, and both are  c1[mr] g[mr][mc] + c2[mr] g[mr < 6 ? 6 : 5][mc]  with the right slots;
//   * the waves meet at node 25: H_26 and X_25 cross through one 64-word LDS row, z_24 / w_26 through 2 x 7 words, and both
//     waves solve the middle node redundantly (no extra barrier before the solution runs outwards);
//   * synchronisation is a command protocol with a FIXED number of s_barriers per command - FACTOR: dispatch, one per staging
//     pass of the Jacobian blocks (7: one block row of all nodes per pass, two staging rows in turn), F1, F2 = 10;  SOLVE:
//     dispatch, A, C = 3;  QUIT: dispatch = 1 - executed by BOTH waves, so the two cannot disagree on the next barrier: wave 1
//     takes no decision of its own.  Every word that crosses is written before one barrier and read after it, and not rewritten
//     before the next.  (tests/test_k8_uniform_control.py: no barrier sits under a divergent branch.)
//   * a factorisation fails (-> Newton failure, halved step) when an inverse block has an entry that is not below 1e300 in
//     magnitude - ONE test per inverse since round 4; rounds 1-3 tested every pivot against a relative threshold
//     (gj_pivot_ok), which the two chains' shared basic block does not allow.  Near-singular matrices that passed neither test
//     fail either way; between the two, the Newton iteration's own divergence test catches a bad factorisation.
// (A first version ran everything that is parallel over nodes in BOTH waves on half the nodes each: correct, but 1.45 x the
// vector instructions of v3 and 6 % slower - with two waves per SIMD the instruction count is what matters.  Here wave 1 is
// idle while wave 0 evaluates residuals and norms, and the SIMD's other wave has the issue slots.)
// Arithmetic per node and per norm is that of the two-ended v3, operation by operation (the masked coefficient of the unified
// chain step multiplies by an exact zero), so v4 reproduces v3 BIT FOR BIT (tools/meth_v3_check.py N v4 v3).  PARITY UNPINNED
// against the reference's IDA like every K8 version (see meth_dae.h).
#pragma once
#include <cstdlib>

#include "meth_dae_elem.h"

namespace smc {
// which K8 the launches use: the two-wave kernels of this header (default), or with SMC_K8_SPLIT=0 the one-wave kernels of
// meth_dae_elem.h (same results bit for bit; kept for A/B runs)
inline bool meth_split_enabled() {
    const char *e = getenv("SMC_K8_SPLIT");
    return e ? atoi(e) != 0 : true;
}
inline int meth_split_role_policy() {   // SMC_K8_SPLIT_ROLES: 0 = wave 0 is the integrator (default), 1 = flipped by wave-slot parity, +4 = census
    const char *e = getenv("SMC_K8_SPLIT_ROLES");
    return e ? atoi(e) : 0;
}
namespace meth {

constexpr int kSplitThreads = 128;
constexpr int kCut = kMid + 1;                   // wave 0 eliminates nodes 0 .. 24 and the middle node 25, wave 1 nodes 50 .. 26
constexpr int kLdsXch = kLdsDoubles;             // 64: H_26 in element layout (wave 1 -> 0), then X_25 (wave 0 -> 1)
constexpr int kLdsMid = kLdsXch + 64;            // [wave][8]: z_24 / w_26 (slot 7: the lanes that hold no result)
constexpr int kLdsFac = kLdsMid + 16;            // [2]: factorisation flags
constexpr int kLdsCmd = kLdsFac + 2;             // [4]: command, -, -, hardware id (split_role)
constexpr int kLdsSplitDoubles = kLdsCmd + 4;
static_assert(kLdsSplitDoubles * 8 * 4 <= 160 * 1024, "four solves per CU");
enum : int { kCmdQuit = 0, kCmdFactor = 1, kCmdSolve = 2 };

__device__ __forceinline__ void split_barrier() { __syncthreads(); }
// The lane index as a value the compiler cannot see through: what is derived from it (row / column indices, LDS offsets, masks) is
// then recomputed where it is used - a dozen integer instructions per phase - instead of being hoisted out of the step loop
// and parked in VGPRs (or, at the 256-register limit, in scratch) for the whole solve.
__device__ __forceinline__ int opaque_lane(int lane) {
    asm volatile("" : "+v"(lane));
    return lane;
}
// Which of the workgroup's two waves is the integrator ("main", role 0) and which the chain server (role 1).  The main wave
// executes vector instructions about half of the time, the server a quarter, so every SIMD should hold one of each.  The
// dispatcher already sees to that: with four 2-wave workgroups per CU the waves (0, 1) of the groups land on SIMDs (0, 2), (2, 1),
// (1, 3), (3, 0) - one wave 0 and one wave 1 per SIMD on all 1024 SIMDs (census: a -DSMC_K8_CENSUS build with SMC_K8_SPLIT_ROLES=4, one printf line per
// wave, profiles/r04_k8_split_placement.txt) - so policy 0 (wave 0 integrates) is the default.  Policy 1 flips the roles with the
// parity of wave 0's hardware slot (HW_REG_HW_ID bits 3:0); measured: it unbalances half of the SIMDs and is 1 - 2 % slower.
__device__ __forceinline__ int split_role(double *lds, int wave_in_group, int policy) {
    const unsigned hw = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4);      // HW_REG_HW_ID, bits 3:0 = WAVE_ID
    if (wave_in_group == 0) lds[kLdsCmd + 3] = (double)hw;
    split_barrier();
    const int slot0 = __builtin_amdgcn_readfirstlane((int)lds[kLdsCmd + 3]);
    const int role = wave_in_group ^ ((policy & 1) ? (slot0 & 1) : 0);
#ifdef SMC_K8_CENSUS   // census builds only (tools/k8_split_census.sh): a device printf brings loops with divergent exits into the kernel
    if ((policy & 4) && (threadIdx.x & 63) == 0) {   // placement census (SMC_K8_SPLIT_ROLES=4 / 5): one line per wave
        const unsigned hwid = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4), xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);
        printf("[k8 placement] group %u wave %d role %d xcc %u se %u cu %u simd %u slot %u\n", blockIdx.x, wave_in_group, role, xcc,
               (hwid >> 13) & 7, (hwid >> 8) & 15, (hwid >> 4) & 3, hwid & 15);
    }
#endif
    return role;
}
// The solve's parameters as values the compiler cannot see through (they stay in their SGPRs): node_eval derives some twenty
// wave-uniform products from them (1/dz, vd Dz / dz^2, (1 - vd) nu_f, P0 / R ...).  Left alone, the compiler hoists these out of
// the step loop - they are loop-invariant - into VGPRs it does not have: they went to scratch and were reloaded in every
// residual (9 loads per Newton iteration).  Recomputed per evaluation they cost twenty multiplies and one division beside
// ~800 instructions.
__device__ __forceinline__ void opaque_params(const double *p, double (&q)[18]) {
    SMC_UNROLL
    for (int i = 0; i < 18; ++i) {
        q[i] = wave_uniform(p[i]);      // (already scalar in the kernels: they load the parameters through wave_uniform)
        asm volatile("" : "+s"(q[i]));
    }
}
// wave 0 -> wave 1: the next thing to do.  Every lane stores the same words (no lane branch in front of the barrier); the
// barrier publishes them together with whatever wave 0 wrote for the command (right-hand side, parameters, differences).
__device__ __forceinline__ void split_command(double *lds, int cmd) {
    lds[kLdsCmd] = (double)cmd;
    split_barrier();
}

// per lane and node parity: where in a node's coefficient row (meth_dae_elem.h: [0..6] ld, [7] 0, [8..14] lx, [15] 0, [16..21] ud,
// [22] 0, [23] u65) the operands of the chain steps sit - for the downward chain (L G, X U) or the upward one (U H, X L)
struct SplitOffsets {
    int f1, f2;          // coupling of the factorisation: c1[mr], c2[mr]
    int gd, ge;          // coupling factor: d[mc], e (uniform)
    int s1, s2, s3, b;   // inward scan: coefficients of prev[mc], prev[6], prev[5]; right-hand side
};
template <int Q>
__device__ __forceinline__ SplitOffsets split_offsets(const ElemLane &L, int w) {
    const int mr = L.template mr<Q>(), mc = L.template mc<Q>(), r6 = min6(mr), c6 = min6(mc);
    SplitOffsets o;
    if (w == 0) {   // downwards: L_i = diag(ld) + column 6 (rows 0..5: lx) + [6][5] (lx[6]);  G = X U, U = diag(ud) + [6][5] (u65)
        o.f1 = r6;
        o.f2 = 8 + r6;
        o.gd = 16 + c6;                       // [22] == 0: column 6 of U is empty
        o.ge = 23;
        o.s1 = c6;
        o.s2 = (mc < 6) ? 8 + mc : 7;
        o.s3 = (mc < 6) ? 7 : 8 + 6;
    } else {        // upwards: (U H)[mr] = ud[mr] H[mr] (mr < 6), u65 H[5] (mr == 6);  H = X L
        o.f1 = 16 + r6;                       // [22] == 0 for row 6
        o.f2 = (mr == 6) ? 23 : 7;
        o.gd = c6;
        o.ge = 8 + 6;
        o.s1 = 16 + c6;
        o.s2 = 7;
        o.s3 = (mc < 6) ? 22 : 23;
    }
    o.b = c6;
    return o;
}

struct SplitChain {      // what a chain step needs besides the factors
    ElemLane L;
    int w, node0, sgn;   // chain position k is node node0 + sgn k
    SplitOffsets o0, o1;
    __device__ __forceinline__ SplitChain(int w_, int lane) : L(lane), w(w_), node0(w_ ? kNX - 1 : 0), sgn(w_ ? -1 : 1),
                                                              o0(split_offsets<0>(L, w_)), o1(split_offsets<1>(L, w_)) {}
    __device__ __forceinline__ int node(int k) const { return node0 + sgn * k; }
    template <int Q> __device__ __forceinline__ const SplitOffsets &off() const { return Q ? o1 : o0; }
};

// Gauss-Jordan inversion of the 7 x 7 block spread over the lanes (meth_dae_elem.h: elem_factor_pair, one chain)
template <int Q, bool SHORT>
__device__ __forceinline__ double split_invert(const ElemLane &L, double a) {
    const double rowsign = Q ? -1.0 : 1.0;
    double akk = lane_bcast(a, 0);
    SMC_UNROLL
    for (int kk = 0; kk < 7; ++kk) {
        const double p = SHORT ? recip1_short(akk) : recip1(akk);
        const double u = __shfl(a, (L.lane & ~7) | kk), v = __shfl(a, kk * 8 + L.c);
        const double gen = fma(-(u * v), p, a);
        akk = lane_bcast(gen, kk < 6 ? 9 * kk + 9 : 0);
        const double ap = a * p * rowsign;
        const bool rk = L.r == kk, ck = L.c == kk;
        a = rk ? (ck ? p : ap) : (ck ? -ap : gen);
    }
    return a;
}

// chain position K of the elimination:  D' = D - C g_{K-1},  X = D'^{-1},  g_K = X C'   (C, C' = L, U downwards; U, L upwards)
template <int K>
__device__ __forceinline__ int split_factor_node(const SplitChain &C, const double *cf, double (&X)[kMid + 1], double (&G)[kMid]) {
    constexpr int Q = K & 1;
    const ElemLane &L = C.L;
    const int mr = L.template mr<Q>(), mc = L.template mc<Q>();
    const SplitOffsets &o = C.template off<Q>();
    const double *cK = cf + C.node(K) * kCfRow;
    double a = X[K];
    if (K > 0) {
        constexpr int KP = (K > 0) ? K - 1 : 0;
        const int kap = mr < 6 ? 6 : 5;
        const int srcT = L.c * 8 + L.r, srcK = Q ? (kap * 8 + L.r) : (L.c * 8 + kap);
        const double gT = __shfl(G[KP], srcT), gK = __shfl(G[KP], srcK);
        a = fma(-cK[o.f2], gK, fma(-cK[o.f1], gT, a));
    }
    a = split_invert<Q, true>(L, a);
    X[K] = a;
    const int src6 = Q ? (48 + L.c) : ((L.lane & ~7) | 6);          // holder of X[mr][6]
    const double x6 = __shfl(a, src6), d = cK[o.gd], e = cK[o.ge];
    const double extra = x6 * e;
    double g = fma(a, d, (mc == 5) ? extra : 0.0);
    if (C.w) {   // H = X L: column 6 collects the u column of L (a scalar branch: w is wave-uniform)
        const double lxv = cK[8 + min6(mc)];
        const double S = allsum_over_mc<Q>((mc < 6) ? a * lxv : 0.0);
        g = (mc == 6) ? fma(a, d, S) : g;
    }
    G[K] = g;
    return (int)(fabs(a) < 1e300);      // false for NaN: a vanished pivot shows in the inverse (elem_factor_pair)
}
template <int K>
struct SplitFactorLoop {
    static __device__ __forceinline__ int run(const SplitChain &C, const double *cf, double (&X)[kMid + 1], double (&G)[kMid]) {
        const int before = SplitFactorLoop<K - 1>::run(C, cf, X, G);
        return split_factor_node<K>(C, cf, X, G) & before;
    }
};
template <>
struct SplitFactorLoop<-1> {
    static __device__ __forceinline__ int run(const SplitChain &, const double *, double (&)[kMid + 1], double (&)[kMid]) { return 1; }
};
// the node where the chains meet (wave 0):  D* = D - L G_24 - U H_26 (H_26: element layout, from the exchange row),  X_25 = D*^{-1}
__device__ __forceinline__ int split_factor_middle(const ElemLane &L, const double *cf, const double *xch, double (&X)[kMid + 1],
                                                   const double (&G)[kMid]) {
    constexpr int I = kMid, Q = I & 1;
    const int mr = L.template mr<Q>();
    const double *cfi = cf + I * kCfRow;
    const int srcT = L.c * 8 + L.r;
    const int kap = mr < 6 ? 6 : 5;
    const int srcK = Q ? (kap * 8 + L.r) : (L.c * 8 + kap), src5 = Q ? (5 * 8 + L.r) : (L.c * 8 + 5);
    const double gT = __shfl(G[I - 1], srcT), gK = __shfl(G[I - 1], srcK);
    const double hT = xch[srcT], h5 = xch[src5];
    double a = X[I];
    a = fma(-cfi[8 + min6(mr)], gK, fma(-cfi[min6(mr)], gT, a));
    const double cu = cfi[(mr == 6) ? 23 : 16 + min6(mr)];
    a = fma(-cu, (mr == 6) ? h5 : hT, a);
    int ok = gj_pivot_ok(lane_bcast(a, 0));
    a = split_invert<Q, false>(L, a);
    X[I] = a;
    return ok & (int)(fabs(a) < 1e300);
}

struct SplitOperands {
    double b, c1, c2, c3;
};
// inward scan, positions 0 .. 24:  z_K = X_K (b_K - c1 z_{K-1} - c2 z_{K-1}[6] - c3 z_{K-1}[5]); operands one position ahead
template <int K>
struct SplitForward {
    static __device__ __forceinline__ double run(const SplitChain &C, const double *cf, const double *b, double *z,
                                                 const double (&X)[kMid + 1], double zprev, const SplitOperands &op) {
        constexpr int Q = K & 1;
        const ElemLane &L = C.L;
        const int mr = L.template mr<Q>(), mc = L.template mc<Q>();
        SplitOperands nx{};
        if constexpr (K + 1 < kMid) {
            const SplitOffsets &on = C.template off<1 - Q>();
            const int nn = C.node(K + 1);
            nx.b = b[nn * 7 + on.b];
            nx.c1 = cf[nn * kCfRow + on.s1];
            nx.c2 = cf[nn * kCfRow + on.s2];
            nx.c3 = cf[nn * kCfRow + on.s3];
        }
        double t = op.b;
        if (K > 0) {
            const double z6 = lane_bcast(zprev, Q ? 48 : 6), z5 = lane_bcast(zprev, Q ? 40 : 5);
            t = fma(-op.c3, z5, fma(-op.c2, z6, fma(-op.c1, zprev, t)));
        }
        const double zi = allsum_over_mc<Q>(X[K] * t);
        z[C.node(K) * kZRow + ((mc == 0 && mr < 7) ? mr : 7)] = zi;
        if constexpr (K + 1 < kMid) return SplitForward<K + 1>::run(C, cf, b, z, X, zi, nx);
        else return zi;
    }
};
// outward scan, positions 24 .. 0:  x_K = z_K - g_K x_{K+1}
template <int K>
struct SplitBackward {
    static __device__ __forceinline__ void run(const SplitChain &C, double *z, const double (&G)[kMid], double xnext, double zK) {
        constexpr int Q = K & 1, KN = (K > 0) ? K - 1 : 0;
        const ElemLane &L = C.L;
        const int mr = L.template mr<Q>(), mc = L.template mc<Q>();
        const double zN = z[C.node(KN) * kZRow + min6(L.template mr<1 - Q>())];
        const double xi = zK - allsum_over_mc<Q>(G[K] * xnext);
        z[C.node(K) * kZRow + ((mc == 0 && mr < 7) ? mr : 7)] = xi;
        if constexpr (K > 0) SplitBackward<KN>::run(C, z, G, xi, zN);
    }
};

// Iteration matrix at the predictor, transposition into the element layout, the wave's chain, the middle node.  Wave 0 (lane =
// node, all 51 of them) evaluates the Jacobian blocks, writes every node's coefficient row and stages the blocks one block ROW
// per pass; both waves pick the entries of their chain's nodes out of the staging row.  The passes alternate between two staging
// rows (the right-hand-side row and the solution rows, both idle during a factorisation), so ONE barrier per pass is enough:
// the row written in pass rho + 2 is the one everybody finished reading before the barrier of pass rho + 1.
// (Wave 1 evaluating the blocks of its own nodes - from a predictor recomputed out of the differences array - saved these seven
// barriers but put 276 VGPRs' worth of live values into a 256-register wave: eight of its factors lived in scratch, 7 GB of
// spill stores per launch, and the Jacobian of half the nodes was issued twice.)
__device__ __forceinline__ bool split_build_and_factor(int w, int lane_in, double *lds, const double *yp, const double *psi,
                                                       const double *p_in, double c, double (&X)[kMid + 1], double (&G)[kMid],
                                                       DaeStats &st) {
    SMC_PROF_BEGIN();
    const SplitChain C(w, opaque_lane(lane_in));
    const ElemLane &L = C.L;
    double *cf = lds + kLdsCf, *stage0 = lds + kLdsB, *stage1 = lds + kLdsZ, *xch = lds + kLdsXch, *fac = lds + kLdsFac;
    static_assert(kNX * kZRow >= kNX * 7, "the solution rows hold a staging row");
    SMC_UNROLL
    for (int i = 0; i <= kMid; ++i) X[i] = 0.0;
    if (w == 0) {
        const double cj = 1.0 / c;
        const int lane = L.lane;
        const bool node = lane < kNX;
        double p[18];
        opaque_params(p_in, p);
        double wm[7], wp[7], yd0[7], res[7], Lb[kNB], Db[kNB], Ub[kNB];
        neighbours(yp, wm, wp);
        SMC_UNROLL
        for (int f = 0; f < 7; ++f) yd0[f] = psi[f] * cj;
        SMC_UNROLL
        for (int q = 0; q < kNB; ++q) Lb[q] = Db[q] = Ub[q] = 0.0;
        if (node) {
            node_eval<true>(lane, wm, yp, wp, yd0, p, cj, res, Lb, Db, Ub);
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
        for (int rho = 0; rho < 7; ++rho) {
            double *stage = (rho & 1) ? stage1 : stage0;
            if (node)
                SMC_UNROLL
                for (int cc = 0; cc < 7; ++cc) stage[lane * 7 + cc] = Db[rho * 7 + cc];
            split_barrier();
            if (L.r == rho && L.c < 7)        // even positions: lane (r, c) holds [r][c]
                SMC_UNROLL
                for (int k = 0; k < kMid; k += 2) X[k] = stage[k * 7 + L.c];
            if (L.c == rho && L.r < 7) {      // odd positions and the odd middle node: lane (r, c) holds [c][r]
                SMC_UNROLL
                for (int k = 1; k <= kMid; k += 2) X[k] = stage[k * 7 + L.r];
            }
        }
    } else {
        SMC_UNROLL
        for (int rho = 0; rho < 7; ++rho) {
            const double *stage = (rho & 1) ? stage1 : stage0;
            split_barrier();
            if (L.r == rho && L.c < 7)
                SMC_UNROLL
                for (int k = 0; k < kMid; k += 2) X[k] = stage[(kNX - 1 - k) * 7 + L.c];
            if (L.c == rho && L.r < 7)
                SMC_UNROLL
                for (int k = 1; k < kMid; k += 2) X[k] = stage[(kNX - 1 - k) * 7 + L.r];
        }
    }
    SMC_PROF_ADD(st, 7);   // Jacobian blocks + transposition (wave 0)
    int ok = SplitFactorLoop<kMid - 1>::run(C, cf, X, G);
    ok = __all(ok);
    SMC_PROF_ADD(st, 0);   // the wave's chain
    if (w) {
        xch[L.lane] = G[kMid - 1];       // H_26
        fac[1] = (double)ok;             // (every lane the same word)
    }
    split_barrier();                     // F1
    if (w == 0) {
        ok &= split_factor_middle(L, cf, xch, X, G);
        ok = __all(ok) & (int)(fac[1] != 0.0);
        xch[L.lane] = X[kMid];           // (H_26 has been read: same wave, LDS in order)
        fac[0] = (double)ok;
    }
    split_barrier();                     // F2
    SMC_PROF_ADD(st, 3);   // waiting for the other chain + the middle node
    if (w) X[kMid] = xch[L.lane];
    return __builtin_amdgcn_readfirstlane((int)(fac[0] != 0.0)) != 0;
}

// The linear solve of one Newton iteration, both waves (after the command's barrier: the right-hand side is in LDS): inwards on
// the wave's chain, barrier A, the middle node in both waves, outwards, barrier C.
__device__ __forceinline__ void split_solve(int w, int lane_in, double *lds, const double (&X)[kMid + 1], const double (&G)[kMid]) {
    const SplitChain C(w, opaque_lane(lane_in));
    const ElemLane &L = C.L;
    double *b = lds + kLdsB, *z = lds + kLdsZ, *mid = lds + kLdsMid;
    const double *cf = lds + kLdsCf;
    {
        SplitOperands op{};
        op.b = b[C.node(0) * 7 + C.o0.b];
        const double zl = SplitForward<0>::run(C, cf, b, z, X, 0.0, op);
        constexpr int Q = (kMid - 1) & 1;
        const int mr = L.template mr<Q>(), mc = L.template mc<Q>();
        mid[w * 8 + ((mc == 0 && mr < 7) ? mr : 7)] = zl;
    }
    split_barrier();                     // A: z_24 and w_26 are visible
    {   // the middle node, in both waves:  x_25 = X_25 (b_25 - L_25 z_24 - U_25 w_26)
        constexpr int I = kMid, Q = I & 1;
        const int mr = L.template mr<Q>(), mc = L.template mc<Q>(), c6 = min6(mc);
        const double *cfi = cf + I * kCfRow;
        const double zp = mid[c6], z6 = mid[6], z5 = mid[5], wn = mid[8 + c6], w5 = mid[8 + 5];
        double t = b[I * 7 + c6];
        t = fma(-cfi[(mc < 6) ? 7 : 8 + 6], z5, fma(-cfi[(mc < 6) ? 8 + mc : 7], z6, fma(-cfi[c6], zp, t)));
        t = fma(-cfi[(mc < 6) ? 22 : 23], w5, fma(-cfi[16 + c6], wn, t));
        const int mrn = min6(L.template mr<1 - Q>());
        const double zK = z[C.node(kMid - 1) * kZRow + mrn];      // own z_24 / w_26 on the row index
        const double xm = allsum_over_mc<Q>(X[I] * t);
        if (w == 0) z[I * kZRow + ((mc == 0 && mr < 7) ? mr : 7)] = xm;
        SplitBackward<kMid - 1>::run(C, z, G, xm, zK);
    }
    split_barrier();                     // C: the whole solution is in LDS
}

// one modified-Newton iteration (wave 0); returns RMS(dy/scale) over all unknowns, or -1 if the residual is not finite
__device__ __forceinline__ double split_newton_iteration(int lane, double *lds, double *y, double *dd, const double *psi,
                                                         const double *p_in, double c, double corr, double rtol, double atol,
                                                         const double (&X)[kMid + 1], const double (&G)[kMid], DaeStats &st) {
    SMC_PROF_BEGIN();
    const double cj = 1.0 / c;
    const bool node = lane < kNX;
    double p[18];
    opaque_params(p_in, p);
    double *b = lds + kLdsB, *z = lds + kLdsZ;
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
    if (!__all(finite)) return -1.0;     // (no command: wave 1 stays at its barrier)
    SMC_PROF_ADD(st, 1);   // residual
    split_command(lds, kCmdSolve);
    split_solve(0, lane, lds, X, G);
    SMC_PROF_ADD(st, 2);   // command + both scans + their barriers
    double sumsq = 0.0;
    if (node)
        SMC_UNROLL
        for (int f = 0; f < 7; ++f) {
#if SMC_K8_POLICY
            const double dx = z[lane * kZRow + f] * corr;      // matrix of another cj: 2 / (1 + cjratio)
#else
            const double dx = z[lane * kZRow + f];
#endif
            const double sc = atol + rtol * fabs(y[f] - dd[f]);      // the predictor: y = yp + dd (kept as y and dd only - 14 VGPRs)
            const double q = dx * recip1(sc);
            sumsq += q * q;
            y[f] += dx;
            dd[f] += dx;
        }
    wave_lds_sync();
    const double nrm = sqrt(allsum_wave(sumsq) / kNS);
    SMC_PROF_ADD(st, 11);  // update + norm
    return nrm;
}

// Wave 1: serves the commands of wave 0 until it is told to quit (the kernel is over).  It keeps the factors of its chain
// between commands and nothing else.
__device__ __forceinline__ void dae_split_server(double *lds, int lane) {
    double X[kMid + 1], G[kMid];
    const double *slot = lds + kLdsCmd;
    for (;;) {
        split_barrier();
        const int cmd = __builtin_amdgcn_readfirstlane((int)slot[0]);
        if (cmd == kCmdQuit) break;
        if (cmd == kCmdFactor) {
            DaeStats unused;
            (void)split_build_and_factor(1, lane, lds, nullptr, nullptr, nullptr, 0.0, X, G, unused);
        } else {
            split_solve(1, lane, lds, X, G);
        }
    }
}

// Integrate one solve (wave 0; wave 1 is in dae_split_server).  lds: the workgroup's region of kLdsSplitDoubles doubles, holding
// y0 in row 0 of the differences array, zeros in rows 1..7 and the parameters in the kLdsPar row on entry; the state at tf is
// left in row 0.  Time stepping, Newton control and error tests: meth_dae_elem.h (dae_elem_integrate), line by line.
__device__ __forceinline__ void dae_split_integrate(double *lds, int lane, const double *p, double tf, double rtol, double atol,
                                                    double h0, int max_attempts, DaeStats &st) {
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
    double X[kMid + 1], G[kMid];
    bool lu_valid = false, force_rebuild = false;
    double c_lu = 0.0;
    double ss = kSsAfterSetup, c_last = 0.0;   // SMC_K8_POLICY 1 (meth_dae_elem.h): carried convergence-rate factor, c of the previous attempt
    double y[7], psi[7], dd[7];
    for (;;) {  // one iteration = one step attempt
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
        t_new = wave_uniform(t_new);
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
                y[f] = s[f];
                psi[f] = q[f] * inv_alpha;
                dd[f] = 0.0;
            }
        }
        const bool fresh = !lu_valid || force_rebuild || matrix_is_stale(c, c_lu);
#if SMC_K8_POLICY
        if (c != c_last) ss = kSsAfterCjChange;
        c_last = c;
#endif
        SMC_PROF_ADD(st, 6);   // predictor
        if (fresh) {
            ++st.nlu;
            split_command(lds, kCmdFactor);
            lu_valid = split_build_and_factor(0, lane, lds, y, psi, p, c, X, G, st);      // y is the predictor here
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
                const double dy_norm = split_newton_iteration(lane, lds, y, dd, psi, p, c, corr, rtol, atol, X, G, st);
                n_iter = kk + 1;
                ++st.newton_iters;
                if (dy_norm < 0) break;
                const int verdict = newton_verdict_ida(kk, dy_norm, dy_first, ss);
                if (verdict != 0) { converged = verdict > 0; break; }
            }
#else
            double dy_norm_old = -1.0;
#pragma unroll 1
            for (int kk = 0; kk < kNewtonMaxIter; ++kk) {
                const double dy_norm = split_newton_iteration(lane, lds, y, dd, psi, p, c, corr, rtol, atol, X, G, st);
                n_iter = kk + 1;
                ++st.newton_iters;
                if (dy_norm < 0) break;
                const double rate = (dy_norm_old >= 0) ? dy_norm / dy_norm_old : -1.0;
                const double scaled = dy_norm / (1 - rate);
                if (rate >= 0 && (rate >= 1 || ipow_small(rate, kNewtonMaxIter - kk) * scaled > newton_tol)) break;
                if (dy_norm == 0 || (rate >= 0 && rate * scaled < newton_tol)) { converged = true; break; }
                dy_norm_old = dy_norm;
            }
#endif
        }
        SMC_PROF_ADD(st, 8);   // factorisation + Newton loop incl. control
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
                const double isc = recip1(atol + rtol * fabs(y[f]));
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
        const double em_s = sqrt(allsum_wave(sm) / (6 * kNX)), ep_s = sqrt(allsum_wave(sp) / (6 * kNX));
        const double em = (order > 1) ? em_s : inf;
        const double ep = (order < kMaxOrder) ? ep_s : inf;
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
