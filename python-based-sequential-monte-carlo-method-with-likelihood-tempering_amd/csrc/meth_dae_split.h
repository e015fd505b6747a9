// meth_dae_split.h -- K8 v4: one solve per WORKGROUP OF TWO WAVES; each wave owns one half of the reactor.
//
// Why.  v3 (meth_dae_elem.h) keeps the block factors of all 51 nodes in one wave's registers: 2 x 51 doubles per lane = 204 of its
// 472 VGPRs, so one wave per SIMD is all that fits, and the vector ALUs of that SIMD are busy 43 % of the cycles
// (profiles/r04_k8_pmc_sq_summary.json): a dependent FP64 instruction waits for its predecessor and nothing else is there to
// issue.  The two-ended elimination of round 4 put two chains into the one instruction stream; they cost 1.5 - 1.7 x a single
// chain, i.e. the stream is then mostly issue-bound - the second chain is not free.  Here the two chains get a wave each:
//   wave 0: nodes 0 .. 25  (chain downwards 0 -> 24, and the middle node 25)          wave 1: nodes 26 .. 50 (chain upwards 50 -> 26)
// Each wave holds HALF of the factors (26 + 25 doubles per lane), which brings the kernel under 256 VGPRs: two waves per SIMD,
// eight per CU, with the same four solves per CU as before (the LDS of a solve is shared by its two waves).  The hardware then
// interleaves two waves per SIMD cycle by cycle, which is what the hand-interleaved pair could only approximate.
//   * everything that is parallel over nodes (predictor, residual, Jacobian, norms, difference updates) runs in BOTH waves on
//     the wave's own nodes (lane = node - 26 w); the neighbour across the cut (node 25 <-> 26) comes through 7 LDS words;
//   * chain position k is node k in wave 0 and node 50 - k in wave 1: same parity, same lane layout, and - with the coefficient
//     masks in the per-lane LDS ADDRESSES (SplitOffsets) - the SAME instruction stream: downwards the coupling is
//     L_i G_{i-1}, upwards U_i H_{i+1}, and both are  c1[mr] g[mr][mc] + c2[mr] g[mr < 6 ? 6 : 5][mc]  with the right slots;
//   * the waves meet at node 25: H_26 and X_25 cross through one 64-word LDS row, z_24 / w_26 through 2 x 7 words, and both
//     waves solve the middle node redundantly (no second barrier before the solution runs outwards);
//   * control flow is IDENTICAL in both waves by construction: every norm is the sum  s_0 + s_1  of the two waves' partial sums
//     taken from LDS in that fixed order, flags cross the same way, so both waves take the same branches and meet at the same
//     s_barrier (P after the predictor, F1 / F2 around the middle factor, A after the inward scan, C after the outward scan, E at
//     the error test).  Every slot that crosses is written before one barrier and read after it, and not rewritten before the next.
// Arithmetic per node is that of the two-ended v3 (same formulas, same order); only the norms are summed in a different order
// (per wave, then the two), so results agree with v3 to the solver's tolerance, not to the bit.  PARITY UNPINNED against the
// reference's IDA like every K8 version (see meth_dae.h); checked against v3 / v2 and the CPU checker.
#pragma once
#include <cstdlib>

#include "meth_dae_elem.h"

namespace smc {
// which K8 the launches use: SMC_K8_SPLIT=1 selects the two-wave kernels of this header, 0 the one-wave kernels of meth_dae_elem.h
inline bool meth_split_enabled() {
    const char *e = getenv("SMC_K8_SPLIT");
    return e ? atoi(e) != 0 : false;
}
namespace meth {

constexpr int kSplitThreads = 128;
constexpr int kCut = kMid + 1;                   // wave 0: nodes 0 .. 25, wave 1: nodes 26 .. 50
constexpr int kLdsXch = kLdsDoubles;             // 64: H_26 in element layout (wave 1 -> 0), then X_25 (wave 0 -> 1)
constexpr int kLdsYb = kLdsXch + 64;             // [owner wave][predictor | current][7]: unknowns of the boundary nodes 25 / 26
constexpr int kLdsMid = kLdsYb + 28;             // [wave][8]: z_24 / w_26 (slot 7: the lanes that hold no result)
constexpr int kLdsNwt = kLdsMid + 16;            // [wave][2]: Newton norm partial sum, flags
constexpr int kLdsErr = kLdsNwt + 4;             // [wave][3]: error-test partial sums (order, order - 1, order + 1)
constexpr int kLdsFac = kLdsErr + 6;             // [2]: factorisation flags
constexpr int kLdsJob = kLdsFac + 2;             // [2]: queue position, cancelled flag (kernels' work hand-out)
constexpr int kLdsSplitDoubles = kLdsJob + 2;
static_assert(kLdsSplitDoubles * 8 * 4 <= 160 * 1024, "four solves per CU");

struct SplitLane {
    int w, lane, node;      // wave of the workgroup, lane of the wave, node held in node layout
    bool active;            // the lane holds a node
    bool take_m, take_p;    // its lower / upper neighbour lives in the other wave
    __device__ __forceinline__ SplitLane(int w_, int l) : w(w_), lane(l), node(w_ ? kCut + l : l), active(l < (w_ ? kNX - kCut : kCut)),
                                                          take_m(w_ == 1 && l == 0), take_p(w_ == 0 && l == kCut - 1) {}
    __device__ __forceinline__ int chain_node(int k) const { return w ? kNX - 1 - k : k; }
};

__device__ __forceinline__ void split_barrier() { __syncthreads(); }
// The lane index as a value the compiler cannot see through: what is derived from it (row / column indices, LDS offsets, masks) is
// then recomputed where it is used - a dozen integer instructions per phase - instead of being hoisted out of the step loop
// and parked in VGPRs (or, at the 256-register limit, in scratch) for the whole solve.
__device__ __forceinline__ int opaque_lane(int lane) {
    asm volatile("" : "+v"(lane));
    return lane;
}

// unknowns of the neighbouring nodes: wave shuffles, and the other wave's boundary node from LDS (yb: its 7 words; a broadcast read)
__device__ __forceinline__ void split_neighbours(const SplitLane &S, const double *w0, const double *yb, double *wm, double *wp) {
    SMC_UNROLL
    for (int f = 0; f < 7; ++f) {
        const double lo = __shfl_up(w0[f], 1), hi = __shfl_down(w0[f], 1), far = yb[f];
        wm[f] = S.take_m ? far : lo;
        wp[f] = S.take_p ? far : hi;
    }
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

// iteration matrix at the predictor (own nodes), transposition into the element layout, the wave's chain, the middle node
__device__ __forceinline__ bool split_build_and_factor(const SplitLane &S, double *lds, const double *yp,
                                                       const double *psi, const double *p, double c, double (&X)[kMid + 1],
                                                       double (&G)[kMid]) {
    const double cj = 1.0 / c;
    const SplitChain C(S.w, opaque_lane(S.lane));
    const ElemLane &L = C.L;
    double *cf = lds + kLdsCf, *stage = lds + kLdsB, *xch = lds + kLdsXch, *fac = lds + kLdsFac;
    {
        double wm[7], wp[7], yd0[7], res[7], Lb[kNB], Db[kNB], Ub[kNB];
        split_neighbours(S, yp, lds + kLdsYb + (1 - S.w) * 14, wm, wp);
        SMC_UNROLL
        for (int f = 0; f < 7; ++f) yd0[f] = psi[f] * cj;
        SMC_UNROLL
        for (int q = 0; q < kNB; ++q) Lb[q] = Db[q] = Ub[q] = 0.0;
        if (S.active) node_eval<true>(S.node, wm, yp, wp, yd0, p, cj, res, Lb, Db, Ub);
        SMC_UNROLL
        for (int i = 0; i <= kMid; ++i) X[i] = 0.0;
        if (S.active) {
            double *o = cf + S.node * kCfRow;
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
        for (int rho = 0; rho < 7; ++rho) {   // one block row of the wave's nodes per pass through its part of the staging row
            if (S.active)
                SMC_UNROLL
                for (int cc = 0; cc < 7; ++cc) stage[S.node * 7 + cc] = Db[rho * 7 + cc];
            wave_lds_sync();
            if (L.r == rho && L.c < 7)        // even positions: lane (r, c) holds [r][c]
                SMC_UNROLL
                for (int k = 0; k < kMid; k += 2) X[k] = stage[C.node(k) * 7 + L.c];
            if (L.c == rho && L.r < 7) {      // odd positions (and the odd middle node, wave 0): lane (r, c) holds [c][r]
                SMC_UNROLL
                for (int k = 1; k < kMid; k += 2) X[k] = stage[C.node(k) * 7 + L.r];
                if (S.w == 0) X[kMid] = stage[kMid * 7 + L.r];
            }
            wave_lds_sync();
        }
    }
    int ok = SplitFactorLoop<kMid - 1>::run(C, cf, X, G);
    ok = __all(ok);
    if (S.w) {
        xch[S.lane] = G[kMid - 1];       // H_26
        if (S.lane == 0) fac[1] = (double)ok;
    }
    split_barrier();                     // F1
    if (S.w == 0) {
        ok &= split_factor_middle(L, cf, xch, X, G);
        ok = __all(ok) & (int)(fac[1] != 0.0);
        xch[S.lane] = X[kMid];           // (H_26 has been read: same wave, LDS in order)
        if (S.lane == 0) fac[0] = (double)ok;
    }
    split_barrier();                     // F2
    if (S.w) X[kMid] = xch[S.lane];
    return __builtin_amdgcn_readfirstlane((int)(fac[0] != 0.0)) != 0;
}

// one modified-Newton iteration; returns RMS(dy/scale) over all unknowns, or -1 if the residual is not finite
__device__ __forceinline__ double split_newton_iteration(const SplitLane &S, double *lds, double *y, double *dd,
                                                         const double *psi, const double *p, double c, double rtol,
                                                         double atol, const double (&X)[kMid + 1], const double (&G)[kMid]) {
    const double cj = 1.0 / c;
    const SplitChain C(S.w, opaque_lane(S.lane));
    const ElemLane &L = C.L;
    double *b = lds + kLdsB, *z = lds + kLdsZ, *mid = lds + kLdsMid, *nwt = lds + kLdsNwt, *ybo = lds + kLdsYb + S.w * 14 + 7;
    const double *cf = lds + kLdsCf;
    int finite = 1;
    {
        double wm[7], wp[7], yd0[7], res[7];
        split_neighbours(S, y, lds + kLdsYb + (1 - S.w) * 14 + 7, wm, wp);
        SMC_UNROLL
        for (int f = 0; f < 7; ++f) yd0[f] = (psi[f] + dd[f]) * cj;
        if (S.active) {
            node_eval<false>(S.node, wm, y, wp, yd0, p, cj, res, nullptr, nullptr, nullptr);
            SMC_UNROLL
            for (int r = 0; r < 7; ++r) {
                if (!(res[r] - res[r] == 0.0)) finite = 0;
                b[S.node * 7 + r] = -res[r];
            }
        }
    }
    finite = __all(finite);
    if (S.lane == 0) nwt[S.w * 2 + 1] = (double)finite;
    wave_lds_sync();
    {   // inward on the wave's chain
        SplitOperands op{};
        op.b = b[C.node(0) * 7 + C.o0.b];
        const double zl = SplitForward<0>::run(C, cf, b, z, X, 0.0, op);
        constexpr int Q = (kMid - 1) & 1;
        const int mr = L.template mr<Q>(), mc = L.template mc<Q>();
        mid[S.w * 8 + ((mc == 0 && mr < 7) ? mr : 7)] = zl;
    }
    split_barrier();                     // A: z_24, w_26, b_25 and both finite flags are visible
    if (!__builtin_amdgcn_readfirstlane((int)(nwt[1] != 0.0) & (int)(nwt[3] != 0.0))) return -1.0;      // the same words in both waves: the same branch
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
        if (S.w == 0) z[I * kZRow + ((mc == 0 && mr < 7) ? mr : 7)] = xm;
        SplitBackward<kMid - 1>::run(C, z, G, xm, zK);
    }
    wave_lds_sync();
    double sumsq = 0.0;
    if (S.active)
        SMC_UNROLL
        for (int f = 0; f < 7; ++f) {
            const double dx = z[S.node * kZRow + f];
            const double sc = atol + rtol * fabs(y[f] - dd[f]);      // the predictor: y = yp + dd (kept as y and dd only - 14 VGPRs)
            const double q = dx * recip1(sc);
            sumsq += q * q;
            y[f] += dx;
            dd[f] += dx;
        }
    if (S.take_m || S.take_p)            // the boundary node's new unknowns for the other wave's next residual
        SMC_UNROLL
        for (int f = 0; f < 7; ++f) ybo[f] = y[f];
    const double part = allsum_wave(sumsq);
    if (S.lane == 0) nwt[S.w * 2] = part;
    split_barrier();                     // C
    return sqrt(wave_uniform(nwt[0] + nwt[2]) / kNS);
}

// Integrate one solve (both waves of the workgroup call this with the same arguments).  lds: the workgroup's region of
// kLdsSplitDoubles doubles, holding y0 in row 0 of the differences array and zeros in rows 1..7 on entry (written by the
// owners of the nodes, followed by a barrier); the state at tf is left in row 0.
__device__ __forceinline__ void dae_split_integrate(double *lds, int wave, int lane, const double *p, double tf, double rtol,
                                                    double atol, double h0, int max_attempts, DaeStats &st) {
    const double newton_tol = fmax(10 * 2.220446049250313e-16 / rtol, fmin(0.03, sqrt(rtol)));
    const SplitLane S(wave, lane);
    const bool node = S.active;
    const DViewE D{lds + kLdsD, S.node};
    double *err = lds + kLdsErr, *ybo = lds + kLdsYb + S.w * 14;
    st.steps = st.rejects = st.newton_fail = st.nlu = st.newton_iters = 0;
    st.status = 0;
    double t = 0.0, h_abs = h0;
    int order = 1, n_equal = 0, attempts = 0;
    double X[kMid + 1], G[kMid];
    bool lu_valid = false, force_rebuild = false;
    double c_lu = 0.0;
    double y[7], psi[7], dd[7];
    for (;;) {  // one iteration = one step attempt
        t = wave_uniform(t);
        h_abs = wave_uniform(h_abs);
        c_lu = wave_uniform(c_lu);
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
            elem_change_D(D, order, fabs(t_new - t) / h_abs, node);
            n_equal = 0;
        }
        t_new = wave_uniform(t_new);
        n_equal = __builtin_amdgcn_readfirstlane(n_equal);
        const double h = t_new - t;
        h_abs = fabs(h);
        const double c = h / bdf_alpha(order);
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
            if (S.take_m || S.take_p)
                SMC_UNROLL
                for (int f = 0; f < 7; ++f) ybo[f] = ybo[7 + f] = s[f];
        }
        split_barrier();                 // P: the boundary nodes' predictors are visible
        const bool fresh = !lu_valid || c != c_lu || force_rebuild;
        if (fresh) {
            ++st.nlu;
            lu_valid = split_build_and_factor(S, lds, y, psi, p, c, X, G);      // y is the predictor here
            c_lu = c;
            force_rebuild = false;
        }
        bool converged = false;
        int n_iter = 0;
        if (lu_valid) {
            double dy_norm_old = -1.0;
#pragma unroll 1
            for (int kk = 0; kk < kNewtonMaxIter; ++kk) {
                const double dy_norm = split_newton_iteration(S, lds, y, dd, psi, p, c, rtol, atol, X, G);
                n_iter = kk + 1;
                ++st.newton_iters;
                if (dy_norm < 0) break;
                const double rate = (dy_norm_old >= 0) ? dy_norm / dy_norm_old : -1.0;
                const double scaled = dy_norm / (1 - rate);
                if (rate >= 0 && (rate >= 1 || ipow_small(rate, kNewtonMaxIter - kk) * scaled > newton_tol)) break;
                if (dy_norm == 0 || (rate >= 0 && rate * scaled < newton_tol)) { converged = true; break; }
                dy_norm_old = dy_norm;
            }
        }
        if (!converged && !fresh) {   // stale matrix: same step again with a fresh one
            force_rebuild = true;
            continue;
        }
        if (!converged) {
            ++st.newton_fail;
            lu_valid = false;
            h_abs *= 0.5;
            elem_change_D(D, order, 0.5, node);
            n_equal = 0;
            continue;
        }
        const double safety = 0.9 * (2 * kNewtonMaxIter + 1) / (2.0 * kNewtonMaxIter + n_iter);
        const bool select = n_equal + 1 >= order + 1;
        double se = 0.0;
        if (node)
            SMC_UNROLL
            for (int f = 0; f < 6; ++f) {
                const double isc = recip1(atol + rtol * fabs(y[f]));
                const double e = bdf_error_const(order) * dd[f] * isc;
                se += e * e;
            }
        {
            const double part = allsum_wave(se);
            if (S.lane == 0) err[S.w * 3] = part;
        }
        split_barrier();                 // E
        const double error_norm = sqrt(wave_uniform(err[0] + err[3]) / (6 * kNX));
        if (!(error_norm <= 1)) {
            ++st.rejects;
            const double factor = (error_norm == error_norm) ? fmax(0.2, safety * pow(error_norm, -1.0 / (order + 1))) : 0.2;
            h_abs *= factor;
            elem_change_D(D, order, factor, node);
            n_equal = 0;
            continue;
        }
        ++n_equal;
        t = t_new;
        ++st.steps;
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
        if (!select) continue;
        {
            const double pm = allsum_wave(sm), pp = allsum_wave(sp);
            if (S.lane == 0) { err[S.w * 3 + 1] = pm; err[S.w * 3 + 2] = pp; }
        }
        split_barrier();                 // E2 (steps that select the order: one in order + 1)
        const double inf = __longlong_as_double(0x7ff0000000000000LL);
        const double em_s = sqrt(wave_uniform(err[1] + err[4]) / (6 * kNX)), ep_s = sqrt(wave_uniform(err[2] + err[5]) / (6 * kNX));
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
        elem_change_D(D, order, factor, node);
        n_equal = 0;
    }
    st.status = __builtin_amdgcn_readfirstlane(st.status);
    split_barrier();                     // the differences array is complete for whoever reads the result
}

}  // namespace meth
}  // namespace smc
