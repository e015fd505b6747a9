// meth_dae.h -- K8: time integration of the methanation DAE F(t, X, X'; p) = 0 (one (particle, experiment)
// solve = my_model's inner body, SMC_methanation/methanation_set_likelihood.py:161-208) as host/device
// portable code: the HIP kernel instantiates it with an interleaved HBM workspace (thread per solve), the
// CPU unit check (tools/meth_dae_hostcheck.cpp) with a plain array.
//
// PARITY UNPINNED against the reference's integrator (Assimulo IDA, :167-198): no IDA in the image, no
// reference output for known inputs.  Same class of method, same equations, same tolerances:
//   * variable-order (1-5) BDF/NDF, quasi-constant step size (the algorithm of SciPy's BDF,
//     scipy/integrate/_ivp/bdf.py: differences array D, change_D on step changes, error constants with
//     Shampine-Reichelt kappa, order selection after order+1 equal steps), adapted to the implicit residual:
//     corrector G(d) = F(t_new, y_pred + d, (psi + d)/c) = 0, c = h/alpha_k;
//   * modified Newton, at most 4 iterations, SciPy's rate-based convergence test; iteration matrix
//     dF/dy + (1/c) dF/dy' ANALYTIC, rebuilt and factored at every step attempt at the predictor;
//   * rtol = atol = 1e-6 (Assimulo's IDA defaults), error test on the 306 differential variables only
//     (suppress_alg = True, :176), start from y'(0) = 0 with a small first step of order 1.
//
// Linear algebra: in node-major ordering (7 unknowns per axial node) the iteration matrix is block
// tridiagonal with 51 blocks of 7x7.  Rows 5/6 of nodes >= 1 are swapped (the reference stores the
// total-mass balance, which determines u, in the T slot and the energy balance in the u slot, :123-126):
// the diagonal blocks then have their large entries on the diagonal and block LU WITHOUT pivoting is
// accurate to ~1e-11 on these matrices (cond ~1e12; checked against a pivoted dense solve).  No pivoting
// means no data-dependent indexing, so 64 independent solves per wave stay coalesced in the interleaved
// layout.  One sweep over the nodes builds the blocks and eliminates on the fly; per node only
// W_i = L_i D'_{i-1}^{-1}, LU(D'_i) and U_i (3 x 49 doubles) are stored.
#pragma once
#include <math.h>
#include <stdint.h>

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define SMC_HD __host__ __device__ __forceinline__
#define SMC_UNROLL _Pragma("unroll")
#else
#define SMC_HD inline
#define SMC_UNROLL
#endif

namespace smc {
namespace meth {

constexpr int kNX = 51, kNS = 357, kNB = 49;
constexpr int kDaeMaxAttempts = 3000;   // step attempts per solve before it is given up (status 1, sentinel flows)
constexpr int kMaxOrder = 5, kNewtonMaxIter = 4;

// workspace layout (doubles, per solve)
constexpr int OFF_D = 0;                         // (kMaxOrder + 3) x kNS differences array, field-major
constexpr int OFF_PSI = OFF_D + 8 * kNS;
constexpr int OFF_YP = OFF_PSI + kNS;            // predictor
constexpr int OFF_Y = OFF_YP + kNS;              // Newton iterate
constexpr int OFF_DD = OFF_Y + kNS;              // accumulated correction d
constexpr int OFF_B = OFF_DD + kNS;              // right-hand side / update, node-major
constexpr int OFF_MAT = OFF_B + kNS;             // per node: W (49), LU(D') (49), U (49)
constexpr int kWsDoubles = OFF_MAT + kNX * 3 * kNB;

// element idx of this solve lives at base[idx * stride] (stride = number of solves laid side by side)
struct Ws {
    double *base;
    int64_t stride;
    SMC_HD double &operator()(int idx) const { return base[(int64_t)idx * stride]; }
};

struct DaeStats {
    int steps, rejects, newton_fail, nlu, newton_iters;
    int status;  // 0 ok, 1 step-size underflow / attempt budget exhausted, 2 singular block
#ifdef SMC_METH_PROFILE
    long long prof[12];  // shader-clock cycles: 0 build+factor, 1 residual, 2 forward, 3 backward, 4 total, 5 change_D, 6 predictor, 7 D update
#endif
};
#ifdef SMC_METH_PROFILE
#define SMC_PROF_BEGIN() long long prof_t0_ = clock64()
#define SMC_PROF_ADD(st_, slot_) do { const long long prof_t1_ = clock64(); (st_).prof[slot_] += prof_t1_ - prof_t0_; prof_t0_ = prof_t1_; } while (0)
#else
#define SMC_PROF_BEGIN() do {} while (0)
#define SMC_PROF_ADD(st_, slot_) do {} while (0)
#endif

namespace k {  // constants (methanation_set_conditon.py:74-89)
constexpr double Dz = 0.95e-5, Rhos = 5075, Hr = -164940, R = 8.3144589, Cpg = 2800, Cps = 698, Keff = 0.72,
                 Dint = 0.005, U = 68.2480;
}

// rate law r and its partial derivatives w.r.t. (Ca, Cb, Cc, Cd, T)  (methanation_set_likelihood.py:44-58).
// Round 4: an IEEE division is ~13 instructions of a wave that issues one instruction per 4 cycles, and the residual of a Newton
// iteration held 33 of them (a third of its instructions).  Quotients by the same divisor share ONE reciprocal (iT = 1/T,
// 1/(q1 q1), 1/(w1 w1) here; 1/dz, 1/T_i, 1/T_{i+-1} in node_eval): 2-3 ulp per term instead of 0.5, against terms that are
// compared at 1e-6 - the K7 kernels that are pinned to the reference's residual values (meth_model.h) are untouched.
SMC_HD double rate_and_grad(double T, double iT, double Ca, double Cb, double Cc, double Cd, const double *kin, double g[5]) {
    constexpr double iR = 1.0 / k::R;
    const double a = k::R * T * 1e-06, da = k::R * 1e-06;
    const double PH2 = Ca * a, PCO2 = Cb * a, PCH4 = Cc * a, PH2O = Cd * a;
    const double iRT = iR * iT, iRT2 = iRT * iT;
    const double kf = kin[0] * exp(-kin[1] * iRT), dkf = kf * kin[1] * iRT2;
    const double ks = kin[2] * exp(-kin[3] * iRT), dks = ks * kin[3] * iRT2;
    const double kc = kin[4] * exp(-kin[5] * iRT), dkc = kc * kin[5] * iRT2;
    const double kh = kin[6] * exp(-kin[7] * iRT), dkh = kh * kin[7] * iRT2;
    const double A = 5075e3;
    const bool clamp = !(PH2 > 0.001);
    const double PH2c = clamp ? 0.001 : PH2;
    const double s = sqrt(PH2c);
    const double q = kc * PCO2, q1 = 1 + q, iq1 = 1.0 / q1, iq2 = iq1 * iq1;
    const double rf = A * kf * q * s * iq2;
    const double drf_dq = A * kf * s * (1 - q) * (iq2 * iq1);
    const double drf_dP = clamp ? 0.0 : rf * 0.5 / PH2c;
    const double w = kh * PH2O, w1 = 1 + w, iw1 = 1.0 / w1, iw2 = iw1 * iw1;
    const double rr = A * ks * w * (PCH4 * PCH4) * iw2;
    const double drr_dw = A * ks * (PCH4 * PCH4) * (1 - w) * (iw2 * iw1);
    const double drr_dP = A * ks * w * 2 * PCH4 * iw2;
    g[0] = drf_dP * a;
    g[1] = drf_dq * kc * a;
    g[2] = -drr_dP * a;
    g[3] = -drr_dw * kh * a;
    const double drf_dT = (A * q * s * iq2) * dkf + drf_dq * (dkc * PCO2 + kc * Cb * da) + drf_dP * Ca * da;
    const double drr_dT = (A * w * (PCH4 * PCH4) * iw2) * dks + drr_dw * (dkh * PH2O + kh * Cd * da) + drr_dP * Cc * da;
    g[4] = drf_dT - drr_dT;
    return rf - rr;
}

// Residual rows of node i in SOLVER row order (rows 5/6 swapped for i >= 1) and, when JAC, the three 7x7
// blocks L (w.r.t. node i-1), D (node i), U (node i+1) of dF/dy + cj*dF/dy', row-major.
// wm, w0, wp: the 7 unknowns (Ca,Cb,Cc,Cd,Ce,T,u) of nodes i-1, i, i+1; yd0: y' of node i (only f < 6 used).
template <bool JAC>
SMC_HD void node_eval(int i, const double *wm, const double *w0, const double *wp, const double *yd0, const double *p,
                      double cj, double *res, double *Lb, double *Db, double *Ub) {
    const double T_in = p[5];
    const double P0 = p[0] * k::R * T_in + p[1] * k::R * T_in + p[2] * k::R * T_in + p[3] * k::R * T_in + p[4] * k::R * T_in;
    const double T_jacket = p[6], u_in = p[7], vd = p[8], dz = p[9];
    if (JAC)
        SMC_UNROLL
        for (int q = 0; q < kNB; ++q) Lb[q] = Db[q] = Ub[q] = 0.0;
    if (i == 0) {  // :96-102
        SMC_UNROLL
        for (int f = 0; f < 6; ++f) {
            res[f] = yd0[f];
            if (JAC) Db[f * 7 + f] = cj;
        }
        res[6] = w0[6] - u_in;
        if (JAC) Db[6 * 7 + 6] = 1.0;
        return;
    }
    if (i == kNX - 1) {  // :130-137 (u equation sits in the T slot of the reference, T equation in the u slot)
        SMC_UNROLL
        for (int f = 0; f < 5; ++f) {
            res[f] = w0[f] - wm[f];
            if (JAC) { Db[f * 7 + f] = 1.0; Lb[f * 7 + f] = -1.0; }
        }
        res[5] = w0[5] - wm[5];  // solver row 5 <- reference slot 6: T[i] - T[i-1]
        res[6] = w0[6] - wm[6];  // solver row 6 <- reference slot 5: u[i] - u[i-1]
        if (JAC) { Db[5 * 7 + 5] = 1.0; Lb[5 * 7 + 5] = -1.0; Db[6 * 7 + 6] = 1.0; Lb[6 * 7 + 6] = -1.0; }
        return;
    }
    const double idz = 1.0 / dz, idz2 = idz * idz;
    const double Ti = w0[5], Tm = wm[5], Tp = wp[5], ui = w0[6], um = wm[6];
    const double iTi = 1.0 / Ti, iTm = 1.0 / Tm, iTp = 1.0 / Tp;
    double g[5];
    const double r = rate_and_grad(Ti, iTi, w0[0], w0[1], w0[2], w0[3], p + 10, g);
    const double sc[5] = {-4, -1, 1, 2, 0};
    const double dif = vd * k::Dz * idz2;
    SMC_UNROLL
    for (int f = 0; f < 5; ++f) {  // :105-109 / :115-119
        const double ci = w0[f], cm = wm[f], cp = wp[f];
        const double diff = (i == 1) ? (cp - ci) : (cp - 2 * ci + cm);
        res[f] = -vd * yd0[f] - (ui * ci - um * cm) * idz + vd * k::Dz * diff * idz2 + (1 - vd) * sc[f] * r;
        if (JAC) {
            const double rs = (1 - vd) * sc[f];
            SMC_UNROLL
            for (int gg = 0; gg < 4; ++gg) Db[f * 7 + gg] = rs * g[gg];
            Db[f * 7 + f] += -vd * cj - ui * idz - ((i == 1) ? dif : 2 * dif);
            Db[f * 7 + 5] = rs * g[4];
            Db[f * 7 + 6] = -ci * idz;
            Lb[f * 7 + f] = um * idz + ((i == 1) ? 0.0 : dif);
            Lb[f * 7 + 6] = cm * idz;
            Ub[f * 7 + f] = dif;
        }
    }
    // gas density (:61-66) and its derivatives
    const double Ssum = w0[0] + w0[1] + w0[2] + w0[3] + w0[4];
    const double Nsum = w0[0] * 2 + w0[1] * 44 + w0[2] * 16 + w0[3] * 18 + w0[4] * 40;
    const double pref = P0 * (1.0 / k::R) * iTi;
    const double iS = 1.0 / Ssum;
    const double rg = pref * Nsum * iS * 0.001;
    const double dT = yd0[5];
    const double kap = (i == 1) ? 1.0 : 0.1;  // :111 vs :126
    const double cap = vd * rg * k::Cpg + (1 - vd) * k::Rhos * k::Cps;
    const double conv = Ti * ui - Tm * um;
    // solver row 5 = energy balance (reference slot 6)
    res[5] = -kap * cap * dT - rg * k::Cpg * conv * idz + k::Keff * (Tp - 2 * Ti + Tm) * idz2 + (1 - vd) * (-k::Hr) * r -
             2 * k::U / k::Dint * (Ti - T_jacket);
    // solver row 6 = total balance (reference slot 5)
    const double tb = -ui * P0 * (iTi - iTm) * idz - P0 * iTi * (ui - um) * idz +
                      vd * k::Dz * P0 * (iTp - 2 * iTi + iTm) * idz2 + (1 - vd) * k::R * (-2) * r;
    res[6] = (i == 1) ? (P0 * vd * (iTi * iTi) * dT + tb) : tb;
    if (JAC) {
        const double mw[5] = {2, 44, 16, 18, 40};
        const double drg_dT = -rg * iTi;
        const double he = (1 - vd) * (-k::Hr);
        SMC_UNROLL
        for (int gg = 0; gg < 5; ++gg) {
            const double drg = pref * 0.001 * (mw[gg] * Ssum - Nsum) * (iS * iS);
            Db[5 * 7 + gg] = -kap * vd * k::Cpg * drg * dT - k::Cpg * drg * conv * idz + ((gg < 4) ? he * g[gg] : 0.0);
        }
        Db[5 * 7 + 5] = -kap * (cap * cj + vd * k::Cpg * drg_dT * dT) - k::Cpg * (drg_dT * conv + rg * ui) * idz -
                        2 * k::Keff * idz2 + he * g[4] - 2 * k::U / k::Dint;
        Db[5 * 7 + 6] = -rg * k::Cpg * Ti * idz;
        Lb[5 * 7 + 5] = rg * k::Cpg * um * idz + k::Keff * idz2;
        Lb[5 * 7 + 6] = rg * k::Cpg * Tm * idz;
        Ub[5 * 7 + 5] = k::Keff * idz2;
        const double ht = (1 - vd) * k::R * (-2);
        SMC_UNROLL
        for (int gg = 0; gg < 4; ++gg) Db[6 * 7 + gg] = ht * g[gg];
        const double iTi2 = iTi * iTi;
        Db[6 * 7 + 5] = ui * P0 * iTi2 * idz + P0 * iTi2 * (ui - um) * idz + dif * P0 * (2 * iTi2) + ht * g[4];
        if (i == 1) Db[6 * 7 + 5] += P0 * vd * (-2.0 * (iTi2 * iTi) * dT + cj * iTi2);
        Db[6 * 7 + 6] = -P0 * (iTi - iTm) * idz - P0 * iTi * idz;
        Lb[6 * 7 + 5] = -ui * P0 * (iTm * iTm) * idz - dif * P0 * (iTm * iTm);
        Lb[6 * 7 + 6] = P0 * iTi * idz;
        Ub[6 * 7 + 5] = -dif * P0 * (iTp * iTp);
    }
}

// 7x7 LU without pivoting, in place (unit lower; the diagonal stores 1/pivot).  Returns false on a zero /
// non-finite pivot.
SMC_HD bool lu7(double *a) {
    SMC_UNROLL
    for (int kk = 0; kk < 7; ++kk) {
        const double piv = a[kk * 7 + kk];
        if (!(fabs(piv) > 1e-300) || !(fabs(piv) < 1e300)) return false;
        const double inv = 1.0 / piv;
        a[kk * 7 + kk] = inv;  // the diagonal holds the RECIPROCAL pivot: no divisions in the solves
        SMC_UNROLL
        for (int r = kk + 1; r < 7; ++r) {
            const double l = a[r * 7 + kk] * inv;
            a[r * 7 + kk] = l;
            SMC_UNROLL
            for (int c = kk + 1; c < 7; ++c) a[r * 7 + c] -= l * a[kk * 7 + c];
        }
    }
    return true;
}
SMC_HD void lu7_solve(const double *lu, double *b) {  // b <- (LU)^{-1} b
    SMC_UNROLL
    for (int r = 1; r < 7; ++r)
        SMC_UNROLL
        for (int c = 0; c < r; ++c) b[r] -= lu[r * 7 + c] * b[c];
    SMC_UNROLL
    for (int r = 6; r >= 0; --r) {
        SMC_UNROLL
        for (int c = r + 1; c < 7; ++c) b[r] -= lu[r * 7 + c] * b[c];
        b[r] *= lu[r * 7 + r];
    }
}
SMC_HD void lu7_rsolve(const double *lu, double *x) {  // x <- x (LU)^{-1}  (row vector)
    SMC_UNROLL
    for (int c = 0; c < 7; ++c) {
        SMC_UNROLL
        for (int q = 0; q < c; ++q) x[c] -= x[q] * lu[q * 7 + c];
        x[c] *= lu[c * 7 + c];
    }
    SMC_UNROLL
    for (int c = 5; c >= 0; --c)
        SMC_UNROLL
        for (int q = c + 1; q < 7; ++q) x[c] -= x[q] * lu[q * 7 + c];
}

// BDF/NDF constants (bdf.py:247-250): gamma_k = sum_{j<=k} 1/j, alpha_k = (1 - kappa_k) gamma_k,
// error_const_k = kappa_k gamma_k + 1/(k+1), kappa = (0, -0.1850, -1/9, -0.0823, -0.0415, 0).
// Looked up with a switch: a table held in registers and indexed with the run-time order makes hipcc emit
// a waterfall loop (s_set_gpr_idx) per access.
SMC_HD double bdf_gamma(int q) {
    switch (q) {
        case 0: return 0.0;
        case 1: return 1.0;
        case 2: return 1.0 + 1.0 / 2;
        case 3: return 1.0 + 1.0 / 2 + 1.0 / 3;
        case 4: return 1.0 + 1.0 / 2 + 1.0 / 3 + 1.0 / 4;
        default: return 1.0 + 1.0 / 2 + 1.0 / 3 + 1.0 / 4 + 1.0 / 5;
    }
}
SMC_HD double bdf_kappa(int q) {
    switch (q) {
        case 1: return -0.1850;
        case 2: return -1.0 / 9;
        case 3: return -0.0823;
        case 4: return -0.0415;
        default: return 0.0;
    }
}
SMC_HD double bdf_alpha(int q) { return (1 - bdf_kappa(q)) * bdf_gamma(q); }
SMC_HD double bdf_error_const(int q) { return (q <= kMaxOrder ? bdf_kappa(q) * bdf_gamma(q) : 0.0) + 1.0 / (q + 1); }
struct BdfConst {
    int unused;
};
SMC_HD BdfConst bdf_constants() { return BdfConst{0}; }

// bdf.py compute_R / change_D: rescale the differences array for a step-size change by `factor`
SMC_HD void change_D(const Ws &ws, int order, double factor) {
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
    for (int x = 0; x < kNS; ++x) {
        double dcol[6], out[6];
        for (int i = 0; i <= order; ++i) dcol[i] = ws(OFF_D + i * kNS + x);
        for (int j = 0; j <= order; ++j) {
            double s = 0.0;
            for (int i = 0; i <= order; ++i) s += RU[i][j] * dcol[i];
            out[j] = s;
        }
        for (int j = 0; j <= order; ++j) ws(OFF_D + j * kNS + x) = out[j];
    }
}

// load the 7 unknowns of node i from a field-major vector stored at offset off
SMC_HD void load_node(const Ws &ws, int off, int i, double *w) {
    for (int f = 0; f < 7; ++f) w[f] = ws(off + f * kNX + i);
}

// One sweep: iteration matrix at the predictor, eliminated on the fly (block LU, no pivoting).
SMC_HD bool build_and_factor(const Ws &ws, const double *p, double c) {
    const double cj = 1.0 / c;
    double wm[7], w0[7], wp[7], yd0[7], res[7];
    double Lb[kNB], Db[kNB], Ub[kNB], Uprev[kNB], LUprev[kNB];
    load_node(ws, OFF_YP, 0, w0);
    load_node(ws, OFF_YP, 1, wp);
    for (int f = 0; f < 7; ++f) wm[f] = 0.0;
    for (int i = 0; i < kNX; ++i) {
        for (int f = 0; f < 7; ++f) yd0[f] = ws(OFF_PSI + f * kNX + i) * cj;  // y' at the predictor: psi / c
        node_eval<true>(i, wm, w0, wp, yd0, p, cj, res, Lb, Db, Ub);
        const int m = OFF_MAT + i * 3 * kNB;
        if (i > 0) {
            // W = L * D'_{i-1}^{-1} (row-wise), D' = D - W * U_{i-1}
            for (int r = 0; r < 7; ++r) lu7_rsolve(LUprev, Lb + r * 7);
            for (int r = 0; r < 7; ++r)
                for (int cc = 0; cc < 7; ++cc) {
                    double s = Db[r * 7 + cc];
                    for (int q = 0; q < 7; ++q) s -= Lb[r * 7 + q] * Uprev[q * 7 + cc];
                    Db[r * 7 + cc] = s;
                }
        }
        if (!lu7(Db)) return false;
        for (int q = 0; q < kNB; ++q) {
            ws(m + q) = Lb[q];
            ws(m + kNB + q) = Db[q];
            ws(m + 2 * kNB + q) = Ub[q];
            LUprev[q] = Db[q];
            Uprev[q] = Ub[q];
        }
        for (int f = 0; f < 7; ++f) { wm[f] = w0[f]; w0[f] = wp[f]; }
        if (i + 2 < kNX) load_node(ws, OFF_YP, i + 2, wp);
    }
    return true;
}

// One modified-Newton iteration: residual at (y, (psi + d)/c), forward elimination fused with the residual
// sweep, back substitution fused with the update.  Returns the RMS norm of dy / scale over all unknowns
// (scale = atol + rtol*|y_pred|) or -1 when the residual is not finite.
SMC_HD double newton_iteration(const Ws &ws, const double *p, double c, double rtol, double atol) {
    const double cj = 1.0 / c;
    double wm[7], w0[7], wp[7], yd0[7], res[7], bprev[7];
    bool finite = true;
    load_node(ws, OFF_Y, 0, w0);
    load_node(ws, OFF_Y, 1, wp);
    for (int f = 0; f < 7; ++f) wm[f] = bprev[f] = 0.0;
    for (int i = 0; i < kNX; ++i) {
        for (int f = 0; f < 7; ++f) yd0[f] = (ws(OFF_PSI + f * kNX + i) + ws(OFF_DD + f * kNX + i)) * cj;
        node_eval<false>(i, wm, w0, wp, yd0, p, cj, res, nullptr, nullptr, nullptr);
        double b[7];
        for (int r = 0; r < 7; ++r) {
            finite = finite && (res[r] - res[r] == 0.0);
            b[r] = -res[r];
        }
        if (i > 0) {
            const int m = OFF_MAT + i * 3 * kNB;
            for (int r = 0; r < 7; ++r) {
                double s = b[r];
                for (int q = 0; q < 7; ++q) s -= ws(m + r * 7 + q) * bprev[q];
                b[r] = s;
            }
        }
        for (int r = 0; r < 7; ++r) { ws(OFF_B + i * 7 + r) = b[r]; bprev[r] = b[r]; }
        for (int f = 0; f < 7; ++f) { wm[f] = w0[f]; w0[f] = wp[f]; }
        if (i + 2 < kNX) load_node(ws, OFF_Y, i + 2, wp);
    }
    if (!finite) return -1.0;
    double xnext[7], sumsq = 0.0;
    for (int f = 0; f < 7; ++f) xnext[f] = 0.0;
    for (int i = kNX - 1; i >= 0; --i) {
        const int m = OFF_MAT + i * 3 * kNB;
        double b[7], lu[kNB];
        for (int r = 0; r < 7; ++r) b[r] = ws(OFF_B + i * 7 + r);
        if (i < kNX - 1)
            for (int r = 0; r < 7; ++r) {
                double s = b[r];
                for (int q = 0; q < 7; ++q) s -= ws(m + 2 * kNB + r * 7 + q) * xnext[q];
                b[r] = s;
            }
        for (int q = 0; q < kNB; ++q) lu[q] = ws(m + kNB + q);
        lu7_solve(lu, b);
        for (int f = 0; f < 7; ++f) {
            const int idx = f * kNX + i;
            const double sc = atol + rtol * fabs(ws(OFF_YP + idx));
            const double qv = b[f] / sc;
            sumsq += qv * qv;
            ws(OFF_Y + idx) += b[f];
            ws(OFF_DD + idx) += b[f];
            xnext[f] = b[f];
        }
    }
    return sqrt(sumsq / kNS);
}

// Integrate one solve from y0 (already stored in D[0]; the other rows of D are zero) to tf.
SMC_HD void dae_integrate(const Ws &ws, const double *p, double tf, double rtol, double atol, double h0, int max_attempts,
                          DaeStats &st) {
    const double newton_tol = fmax(10 * 2.220446049250313e-16 / rtol, fmin(0.03, sqrt(rtol)));
    st.steps = st.rejects = st.newton_fail = st.nlu = st.newton_iters = 0;
    st.status = 0;
    double t = 0.0, h_abs = h0;
    int order = 1, n_equal = 0, attempts = 0;
    while (t < tf) {  // one iteration = one step attempt
        // give up on a vanishing step or when the attempt budget (about 7x what a healthy solve needs) is spent:
        // status 1 -> the reference's -10000 sentinel (:244-249)
        if (h_abs < 1e-14 * fmax(1.0, t) || attempts >= max_attempts) { st.status = 1; return; }
        ++attempts;
        double t_new = t + h_abs;
        if (t_new - tf > 0) {
            t_new = tf;
            change_D(ws, order, fabs(t_new - t) / h_abs);
            n_equal = 0;
        }
        const double h = t_new - t;
        h_abs = fabs(h);
        const double c = h / bdf_alpha(order);
        // predictor, psi; Newton start
        for (int x = 0; x < kNS; ++x) {
            double s = 0.0, q = 0.0;
            for (int kk = 0; kk <= order; ++kk) s += ws(OFF_D + kk * kNS + x);
            for (int kk = 1; kk <= order; ++kk) q += ws(OFF_D + kk * kNS + x) * bdf_gamma(kk);
            ws(OFF_YP + x) = s;
            ws(OFF_Y + x) = s;
            ws(OFF_PSI + x) = q / bdf_alpha(order);
            ws(OFF_DD + x) = 0.0;
        }
        ++st.nlu;
        bool converged = false;
        int n_iter = 0;
        if (build_and_factor(ws, p, c)) {
            double dy_norm_old = -1.0;
            for (int kk = 0; kk < kNewtonMaxIter; ++kk) {
                // the update is applied inside newton_iteration; SciPy tests the rate BEFORE applying it, which
                // only matters for an iteration that fails - the attempt is discarded then anyway
                const double dy_norm = newton_iteration(ws, p, c, rtol, atol);
                n_iter = kk + 1;
                ++st.newton_iters;
                if (dy_norm < 0) break;
                const double rate = (dy_norm_old >= 0) ? dy_norm / dy_norm_old : -1.0;
                if (rate >= 0 && (rate >= 1 || pow(rate, kNewtonMaxIter - kk) / (1 - rate) * dy_norm > newton_tol)) break;
                if (dy_norm == 0 || (rate >= 0 && rate / (1 - rate) * dy_norm < newton_tol)) { converged = true; break; }
                dy_norm_old = dy_norm;
            }
        }
        if (!converged) {
            ++st.newton_fail;
            h_abs *= 0.5;
            change_D(ws, order, 0.5);
            n_equal = 0;
            continue;
        }
        const double safety = 0.9 * (2 * kNewtonMaxIter + 1) / (2.0 * kNewtonMaxIter + n_iter);
        // error test on the differential variables (the first 6 fields)
        double se = 0.0;
        for (int x = 0; x < 6 * kNX; ++x) {
            const double sc = atol + rtol * fabs(ws(OFF_Y + x));
            const double e = bdf_error_const(order) * ws(OFF_DD + x) / sc;
            se += e * e;
        }
        const double error_norm = sqrt(se / (6 * kNX));
        if (!(error_norm <= 1)) {
            ++st.rejects;
            const double factor = (error_norm == error_norm) ? fmax(0.2, safety * pow(error_norm, -1.0 / (order + 1))) : 0.2;
            h_abs *= factor;
            change_D(ws, order, factor);
            n_equal = 0;
            continue;
        }
        // accept: update the differences (bdf.py:396-399) and, when due, the error norms of orders k-1 / k+1
        ++n_equal;
        t = t_new;
        ++st.steps;
        const bool select = n_equal >= order + 1;
        double sm = 0.0, sp = 0.0;
        for (int x = 0; x < kNS; ++x) {
            const double dd = ws(OFF_DD + x);
            const double dnew2 = dd - ws(OFF_D + (order + 1) * kNS + x);
            ws(OFF_D + (order + 2) * kNS + x) = dnew2;
            double acc = dd;
            ws(OFF_D + (order + 1) * kNS + x) = acc;
            double d_order = 0.0;
            for (int kk = order; kk >= 0; --kk) {
                acc += ws(OFF_D + kk * kNS + x);
                ws(OFF_D + kk * kNS + x) = acc;
                if (kk == order) d_order = acc;
            }
            if (select && x < 6 * kNX) {
                const double sc = atol + rtol * fabs(ws(OFF_Y + x));
                if (order > 1) { const double e = bdf_error_const(order - 1) * d_order / sc; sm += e * e; }
                if (order < kMaxOrder) { const double e = bdf_error_const(order + 1) * dnew2 / sc; sp += e * e; }
            }
        }
        if (!select) continue;
        const double inf = 1.0 / 0.0;
        const double em = (order > 1) ? sqrt(sm / (6 * kNX)) : inf;
        const double ep = (order < kMaxOrder) ? sqrt(sp / (6 * kNX)) : inf;
        const double fm = pow(em, -1.0 / order), f0 = pow(error_norm, -1.0 / (order + 1)), fp = pow(ep, -1.0 / (order + 2));
        double best = fm;
        int delta = -1;
        if (f0 > best) { best = f0; delta = 0; }
        if (fp > best) { best = fp; delta = 1; }
        order += delta;
        const double factor = fmin(10.0, safety * best);
        h_abs *= factor;
        change_D(ws, order, factor);
        n_equal = 0;
    }
}

// my_model's outlet mapping (:204-208): state at t = 75 (D[0]) -> 5 standard-state flows
SMC_HD void outlet_flows(const Ws &ws, const double *p, double S, double P_stp, double *F) {
    const double u = ws(OFF_D + 7 * kNX - 1), T = ws(OFF_D + 6 * kNX - 1);
    const double P_total = (p[0] + p[1] + p[2] + p[3] + p[4]) * k::R * p[5];  // :165
    for (int f = 0; f < 5; ++f) {
        const double cc = ws(OFF_D + (f + 1) * kNX - 1);
        F[f] = cc * S * u * 60 * k::R * T / (P_total) * 1e6 * (P_total) / P_stp * 298 / T;
    }
}

}  // namespace meth
}  // namespace smc
