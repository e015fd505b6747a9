// meth_dae_wave.h -- K8 v2: the DAE integrator of meth_dae.h mapped ONE SOLVE PER WAVE, lane = axial node.
//
// Why: v1 (one thread per solve, 97 KB HBM workspace per lane) is latency-bound at ~380 solves/s - every lane
// issues ~45k mostly dependent memory operations per step at one wave per SIMD.  Here everything a solve
// needs stays on chip:
//   * per lane (node i): predictor / iterate / psi / correction of its 7 unknowns, its diagonal block factor
//     LU(D'_i), G_i = D'_i^{-1} U_i and the sparse L_i, U_i  (about 150 doubles in registers);
//   * the differences array D[8][7][64] in LDS (28 KB per wave; the row index depends on the current order);
//   * residual rows and the three Jacobian blocks of all 51 nodes are evaluated IN PARALLEL (one node per
//     lane, neighbours by wave shuffles) - that is where the exp() of the rate law is;
//   * only the block elimination is a scan over the lanes.  L_i has 14 structural non-zeros (diagonal, the
//     u column, one T entry) and U_i has 7, so a scan step is ~450 FMAs:  D'_i = D_i - L_i G_{i-1},
//     LU(D'_i) without pivoting (rows 5/6 swapped, see meth_dae.h), G_i = D'_i^{-1} U_i;  right-hand sides:
//     z_i = D'_i^{-1}(b_i - L_i z_{i-1}) forward, x_i = z_i - G_i x_{i+1} backward.  Values travel between
//     neighbouring lanes with v_readlane (the source lane index is wave-uniform).
// The time-stepping logic (variable-order BDF/NDF, Newton, error control) is that of meth_dae.h, which is
// unit-tested on the CPU (tests/hostcheck) and against the CPU checker; all control decisions here derive from
// wave-reduced norms, so they are uniform across the wave.  PARITY UNPINNED against the reference's IDA.
#pragma once
#include <hip/hip_runtime.h>

#include "meth_dae.h"

namespace smc {
namespace meth {

// Wave-uniform loop control, BY CONSTRUCTION.  Everything that steers a loop of the one-wave-per-solve kernels (the index
// of the next solve, error and Newton norms, hence t, h, the order ...) has the same value in all 64 lanes - but a value
// that reaches the branch through a VGPR (a shuffle, a butterfly sum) is "divergent" to the compiler, which then manages
// the loop with exec masks and may run the two sides of an `if (lane == 0)` as separate trips through the loop.  That is
// what happened to the first dequeue loop of meth_particles_dae_kernel (`for (;;)` + lane-0 atomicAdd + __shfl + `continue`):
// lanes 1..63 reached the ds_bpermute of the __shfl with lane 0 parked on the atomic's side, read 0 from the inactive lane
// and solved item 0 for ever (ISA excerpts: profiles/r02_k8_dequeue_hang_isa.md).  v_readfirstlane puts such a value into
// an SGPR: the branches on it become scalar, no lane can leave a loop or skip a statement on its own, and the DPP / permlane
// / LDS exchanges of the integrator always run with the full wave.
__device__ __forceinline__ double wave_uniform(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
// Next index of a work counter shared by all waves, as a scalar.  EVERY lane issues the atomic (lane 0 adds `step`, the others
// add 0; the compiler's atomic optimiser turns that into one wave reduction and one memory atomic) and v_readfirstlane
// takes lane 0's return value: there is NO divergent branch between the atomic and the cross-lane read.  The obvious
// `if (lane == 0) nxt = atomicAdd(...); idx = readfirstlane(nxt);` is not safe: HIP guarantees no re-convergence after the
// `if`, and the compiler is free to run the lanes that skipped it ahead on their own - readfirstlane then returns THEIR
// nxt = 0.  That is how the round-1 K8 dequeue loop hung, and how the first version of mm_tail_kernel hung in round 2
// (profiles/r02_k8_dequeue_hang_isa.md, both listings).  `split` is set when the wave is incomplete here anyway.
__device__ __forceinline__ long long wave_dequeue(unsigned long long *counter, int lane, unsigned &split,
                                                  unsigned long long step = 1ULL) {
    const unsigned long long nxt = atomicAdd(counter, lane == 0 ? step : 0ULL);
    split |= (unsigned)(__builtin_amdgcn_read_exec() != ~0ull);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)nxt);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(nxt >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ double wave_allsum(double v) {
SMC_UNROLL
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return wave_uniform(v);
}
__device__ __forceinline__ double lane_bcast(double v, int src) {  // src is wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

struct WaveBlocks {   // per lane (node)
    double LU[kNB];   // LU(D'_i), unit lower, no pivoting
    double G[42];     // G_i = D'_i^{-1} U_i, columns 0..5 (column 6 of U_i is structurally zero), row-major 7 x 6
    double Ld[7];     // L_i diagonal
    double Lu[6];     // L_i[r][6], r = 0..5
    double L65;       // L_i[6][5]
    double Ud[6];     // U_i[r][r], r = 0..5
    double U65;       // U_i[6][5]
};

// LDS view of the differences array: D[k][f] of node `lane`
struct DView {
    double *s;
    int lane;
    __device__ __forceinline__ double &operator()(int k, int f) const { return s[(k * 7 + f) * 64 + lane]; }
};

__device__ __forceinline__ void wave_change_D(const DView &D, int order, double factor, bool node) {
    double R[6][6], Um[6][6], RU[6][6];
    
    for (int pass = 0; pass < 2; ++pass) {
        double(*M)[6] = pass ? Um : R;
        const double fac = pass ? 1.0 : factor;
        
        for (int j = 0; j <= order; ++j) M[0][j] = 1.0;
        
        for (int i = 1; i <= order; ++i) {
            M[i][0] = 0.0;
            
            for (int j = 1; j <= order; ++j) M[i][j] = M[i - 1][j] * ((i - 1 - fac * j) / i);
        }
    }
    
    for (int i = 0; i <= order; ++i)
        
        for (int j = 0; j <= order; ++j) {
            double s = 0.0;
            
            for (int q = 0; q <= order; ++q) s += R[i][q] * Um[q][j];
            RU[i][j] = s;
        }
    if (!node) return;
    SMC_UNROLL
    for (int f = 0; f < 7; ++f) {
        double dcol[6], out[6];
        
        for (int i = 0; i <= order; ++i) dcol[i] = D(i, f);
        
        for (int j = 0; j <= order; ++j) {
            double s = 0.0;
            
            for (int i = 0; i <= order; ++i) s += RU[i][j] * dcol[i];
            out[j] = s;
        }
        
        for (int j = 0; j <= order; ++j) D(j, f) = out[j];
    }
}

// neighbours' unknowns by shuffles (all lanes participate)
__device__ __forceinline__ void neighbours(const double *w0, double *wm, double *wp) {
SMC_UNROLL
    for (int f = 0; f < 7; ++f) {
        wm[f] = __shfl_up(w0[f], 1);
        wp[f] = __shfl_down(w0[f], 1);
    }
}

// iteration matrix at the predictor (parallel over nodes) + block elimination (scan). false: singular block.
__device__ __forceinline__ bool wave_build_and_factor(int lane, const double *yp, const double *psi, const double *p,
                                                      double c, WaveBlocks &B) {
    const double cj = 1.0 / c;
    const bool node = lane < kNX;
    double wm[7], wp[7], yd0[7], res[7], Lb[kNB], Db[kNB], Ub[kNB];
    neighbours(yp, wm, wp);
    SMC_UNROLL
    for (int f = 0; f < 7; ++f) yd0[f] = psi[f] * cj;
    if (node) {
        node_eval<true>(lane, wm, yp, wp, yd0, p, cj, res, Lb, Db, Ub);
        SMC_UNROLL
        for (int r = 0; r < 7; ++r) B.Ld[r] = Lb[r * 7 + r];
        SMC_UNROLL
        for (int r = 0; r < 6; ++r) { B.Lu[r] = Lb[r * 7 + 6]; B.Ud[r] = Ub[r * 7 + r]; }
        B.L65 = Lb[6 * 7 + 5];
        B.U65 = Ub[6 * 7 + 5];
    }
    int ok = 1;
    for (int i = 0; i < kNX; ++i) {
        // G_{i-1} rows needed by node i: for row r < 6: Ld[r]*G[r][:] + Lu[r]*G[6][:]; row 6: Ld[6]*G[6][:] + L65*G[5][:]
        double Gp[42];
        if (i > 0)
            SMC_UNROLL
            for (int q = 0; q < 42; ++q) Gp[q] = lane_bcast(B.G[q], i - 1);
        if (lane == i) {
            if (i > 0) {
                SMC_UNROLL
                for (int r = 0; r < 6; ++r)
                    SMC_UNROLL
                    for (int cc = 0; cc < 6; ++cc) Db[r * 7 + cc] -= B.Ld[r] * Gp[r * 6 + cc] + B.Lu[r] * Gp[6 * 6 + cc];
                SMC_UNROLL
                for (int cc = 0; cc < 6; ++cc) Db[6 * 7 + cc] -= B.Ld[6] * Gp[6 * 6 + cc] + B.L65 * Gp[5 * 6 + cc];
            }
            if (!lu7(Db)) ok = 0;
            SMC_UNROLL
            for (int q = 0; q < kNB; ++q) B.LU[q] = Db[q];
            // G = LU^{-1} U : columns 0..4 have one non-zero (diagonal), column 5 has rows 5 and 6
            SMC_UNROLL
            for (int cc = 0; cc < 6; ++cc) {
                double col[7];
                SMC_UNROLL
                for (int r = 0; r < 7; ++r) col[r] = 0.0;
                col[cc] = B.Ud[cc];
                if (cc == 5) col[6] = B.U65;
                lu7_solve(Db, col);
                SMC_UNROLL
                for (int r = 0; r < 7; ++r) B.G[r * 6 + cc] = col[r];
            }
        }
    }
    return __all(ok);
}

// one modified-Newton iteration; returns RMS(dy/scale) over all unknowns, or -1 if the residual is not finite
__device__ __forceinline__ double wave_newton_iteration(int lane, double *y, double *dd, const double *yp,
                                                        const double *psi, const double *p, double c, double rtol,
                                                        double atol, const WaveBlocks &B, DaeStats &st) {
    SMC_PROF_BEGIN();
    const double cj = 1.0 / c;
    const bool node = lane < kNX;
    double wm[7], wp[7], yd0[7], b[7];
    neighbours(y, wm, wp);
    SMC_UNROLL
    for (int f = 0; f < 7; ++f) yd0[f] = (psi[f] + dd[f]) * cj;
    int finite = 1;
    SMC_UNROLL
    for (int f = 0; f < 7; ++f) b[f] = 0.0;
    if (node) {
        double res[7];
        node_eval<false>(lane, wm, y, wp, yd0, p, cj, res, nullptr, nullptr, nullptr);
        SMC_UNROLL
        for (int r = 0; r < 7; ++r) {
            if (!(res[r] - res[r] == 0.0)) finite = 0;
            b[r] = -res[r];
        }
    }
    if (!__all(finite)) return -1.0;
    SMC_PROF_ADD(st, 1);
    // forward: z_i = LU_i^{-1} (b_i - L_i z_{i-1})
    double z[7];
    SMC_UNROLL
    for (int f = 0; f < 7; ++f) z[f] = 0.0;
    for (int i = 0; i < kNX; ++i) {
        double zp[7];
        if (i > 0)
            SMC_UNROLL
            for (int f = 0; f < 7; ++f) zp[f] = lane_bcast(z[f], i - 1);
        if (lane == i) {
            if (i > 0) {
                SMC_UNROLL
                for (int r = 0; r < 6; ++r) b[r] -= B.Ld[r] * zp[r] + B.Lu[r] * zp[6];
                b[6] -= B.Ld[6] * zp[6] + B.L65 * zp[5];
            }
            lu7_solve(B.LU, b);
            SMC_UNROLL
            for (int f = 0; f < 7; ++f) z[f] = b[f];
        }
    }
    SMC_PROF_ADD(st, 2);
    // backward: x_i = z_i - G_i x_{i+1}
    for (int i = kNX - 2; i >= 0; --i) {
        double xn[6];
        SMC_UNROLL
        for (int f = 0; f < 6; ++f) xn[f] = lane_bcast(z[f], i + 1);
        if (lane == i)
            SMC_UNROLL
            for (int r = 0; r < 7; ++r) {
                double s = z[r];
                SMC_UNROLL
                for (int cc = 0; cc < 6; ++cc) s -= B.G[r * 6 + cc] * xn[cc];
                z[r] = s;
            }
    }
    SMC_PROF_ADD(st, 3);
    double sumsq = 0.0;
    if (node)
        SMC_UNROLL
        for (int f = 0; f < 7; ++f) {
            const double sc = atol + rtol * fabs(yp[f]);
            const double q = z[f] / sc;
            sumsq += q * q;
            y[f] += z[f];
            dd[f] += z[f];
        }
    return sqrt(wave_allsum(sumsq) / kNS);
}

// Integrate one solve (the whole wave cooperates).  sD: this wave's LDS region of 8*7*64 doubles, holding y0
// in row 0 and zeros elsewhere on entry; the state at tf is left in row 0.
__device__ __forceinline__ void dae_wave_integrate(double *sD, int lane, const double *p, double tf, double rtol,
                                                   double atol, double h0, int max_attempts, DaeStats &st) {
    const double newton_tol = fmax(10 * 2.220446049250313e-16 / rtol, fmin(0.03, sqrt(rtol)));
    const bool node = lane < kNX;
    const DView D{sD, lane};
    st.steps = st.rejects = st.newton_fail = st.nlu = st.newton_iters = 0;
    st.status = 0;
#ifdef SMC_METH_PROFILE
    for (int q = 0; q < 12; ++q) st.prof[q] = 0;
    const long long prof_start_ = clock64();
#endif
    double t = 0.0, h_abs = h0;
    int order = 1, n_equal = 0, attempts = 0;
    WaveBlocks B;
    bool lu_valid = false, force_rebuild = false;
    double c_lu = 0.0;
    double yp[7], y[7], psi[7], dd[7];
    while (t < tf) {  // one iteration = one step attempt (all quantities below are wave-uniform)
        if (h_abs < 1e-14 * fmax(1.0, t) || attempts >= max_attempts) { st.status = 1; return; }
        ++attempts;
        double t_new = t + h_abs;
        if (t_new - tf > 0) {
            t_new = tf;
            { SMC_PROF_BEGIN(); wave_change_D(D, order, fabs(t_new - t) / h_abs, node); SMC_PROF_ADD(st, 5); }
            n_equal = 0;
        }
        const double h = t_new - t;
        h_abs = fabs(h);
        const double c = h / bdf_alpha(order);
        SMC_PROF_BEGIN();
        SMC_UNROLL
        for (int f = 0; f < 7; ++f) {
            double s = 0.0, q = 0.0;
            if (node) {
                
                for (int kk = 0; kk <= order; ++kk) s += D(kk, f);
                
                for (int kk = 1; kk <= order; ++kk) q += D(kk, f) * bdf_gamma(kk);
            }
            yp[f] = y[f] = s;
            psi[f] = q / bdf_alpha(order);
            dd[f] = 0.0;
        }
        // The factored iteration matrix is kept while c = h/alpha_k is unchanged (consecutive steps of equal size
        // and order - the quasi-constant-step form makes that the common case).  It is rebuilt at the current
        // predictor when c changed, or - by repeating this attempt - when Newton did not converge with the stale
        // matrix (bdf.py:343-357, `current_jac`).
        const bool fresh = !lu_valid || c != c_lu || force_rebuild;
        SMC_PROF_ADD(st, 6);
        if (fresh) {
            ++st.nlu;
            lu_valid = wave_build_and_factor(lane, yp, psi, p, c, B);
            SMC_PROF_ADD(st, 0);
            c_lu = c;
            force_rebuild = false;
        }
        bool converged = false;
        int n_iter = 0;
        if (lu_valid) {
            double dy_norm_old = -1.0;
            
            for (int kk = 0; kk < kNewtonMaxIter; ++kk) {
                const double dy_norm = wave_newton_iteration(lane, y, dd, yp, psi, p, c, rtol, atol, B, st);
                n_iter = kk + 1;
                ++st.newton_iters;
                if (dy_norm < 0) break;
                const double rate = (dy_norm_old >= 0) ? dy_norm / dy_norm_old : -1.0;
                if (rate >= 0 && (rate >= 1 || pow(rate, kNewtonMaxIter - kk) / (1 - rate) * dy_norm > newton_tol)) break;
                if (dy_norm == 0 || (rate >= 0 && rate / (1 - rate) * dy_norm < newton_tol)) { converged = true; break; }
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
            { SMC_PROF_BEGIN(); wave_change_D(D, order, 0.5, node); SMC_PROF_ADD(st, 5); }
            n_equal = 0;
            continue;
        }
        const double safety = 0.9 * (2 * kNewtonMaxIter + 1) / (2.0 * kNewtonMaxIter + n_iter);
        double se = 0.0;
        if (node)
            SMC_UNROLL
            for (int f = 0; f < 6; ++f) {
                const double sc = atol + rtol * fabs(y[f]);
                const double e = bdf_error_const(order) * dd[f] / sc;
                se += e * e;
            }
        const double error_norm = sqrt(wave_allsum(se) / (6 * kNX));
        if (!(error_norm <= 1)) {
            ++st.rejects;
            const double factor = (error_norm == error_norm) ? fmax(0.2, safety * pow(error_norm, -1.0 / (order + 1))) : 0.2;
            h_abs *= factor;
            { SMC_PROF_BEGIN(); wave_change_D(D, order, factor, node); SMC_PROF_ADD(st, 5); }
            n_equal = 0;
            continue;
        }
        ++n_equal;
        t = t_new;
        ++st.steps;
        const bool select = n_equal >= order + 1;
        SMC_PROF_ADD(st, 7);   // Newton control + error test (everything since the factorisation)
        double sm = 0.0, sp = 0.0;
        if (node)
            SMC_UNROLL
            for (int f = 0; f < 7; ++f) {
                const double dnew2 = dd[f] - D(order + 1, f);
                D(order + 2, f) = dnew2;
                double acc = dd[f];
                D(order + 1, f) = acc;
                double d_order = 0.0;
                
                for (int kk = order; kk >= 0; --kk) {
                    acc += D(kk, f);
                    D(kk, f) = acc;
                    if (kk == order) d_order = acc;
                }
                if (select && f < 6) {
                    const double sc = atol + rtol * fabs(y[f]);
                    if (order > 1) { const double e = bdf_error_const(order - 1) * d_order / sc; sm += e * e; }
                    if (order < kMaxOrder) { const double e = bdf_error_const(order + 1) * dnew2 / sc; sp += e * e; }
                }
            }
        SMC_PROF_ADD(st, 6);   // D update shares the predictor slot
        if (!select) continue;
        const double inf = __longlong_as_double(0x7ff0000000000000LL);
        const double em = (order > 1) ? sqrt(wave_allsum(sm) / (6 * kNX)) : inf;
        const double ep = (order < kMaxOrder) ? sqrt(wave_allsum(sp) / (6 * kNX)) : inf;
        const double fm = pow(em, -1.0 / order), f0 = pow(error_norm, -1.0 / (order + 1)), fp = pow(ep, -1.0 / (order + 2));
        double best = fm;
        int delta = -1;
        if (f0 > best) { best = f0; delta = 0; }
        if (fp > best) { best = fp; delta = 1; }
        order += delta;
        const double factor = fmin(10.0, safety * best);
        h_abs *= factor;
        { SMC_PROF_BEGIN(); wave_change_D(D, order, factor, node); SMC_PROF_ADD(st, 5); }
        n_equal = 0;
    }
#ifdef SMC_METH_PROFILE
    st.prof[4] = clock64() - prof_start_;
#endif
}

}  // namespace meth
}  // namespace smc
