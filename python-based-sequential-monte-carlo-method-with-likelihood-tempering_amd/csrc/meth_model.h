// meth_model.h -- device functions of the CO2-methanation tubular-reactor model (configs 4-5):
// rate law, gas density and the 7 residual equations of one axial node of the 357-state index-1 DAE.
// gfx950 only.  Follows, expression by expression,
//   SMC_methanation/methanation_set_likelihood.py:44-58  func_rCH4
//   SMC_methanation/methanation_set_likelihood.py:61-66  func_rohg
//   SMC_methanation/methanation_set_likelihood.py:69-139 reaction (state field-major X[f*51+i],
//       f in {Ca,Cb,Cc,Cd,Ce,T,u}; node 0 inlet rows :96-102, node 1 special-cased :104-111,
//       interior :114-126, node 50 boundary rows with the reference's index expressions :130-137)
// Constants: methanation_set_conditon.py:74-89.
#pragma once
#include <hip/hip_runtime.h>

namespace smc {
namespace meth {

constexpr int NX = 51;
constexpr int NSTATE = 7 * NX;
constexpr int NPAR = 18;  // p0 tuple of my_model (methanation_set_likelihood.py:164)

__device__ constexpr double kDz = 0.95e-5, kRhos = 5075, kHr = -164940, kR = 8.3144589, kCpg = 2800, kCps = 698,
                            kKeff = 0.72, kDint = 0.005, kU = 68.2480;

__device__ __forceinline__ double py_max(double a, double b) { return (b > a) ? b : a; }

__device__ __forceinline__ double rCH4(double T, double Ca, double Cb, double Cc, double Cd, const double *kin) {
    const double PH2 = Ca * kR * T * 1e-06, PCO2 = Cb * kR * T * 1e-06, PCH4 = Cc * kR * T * 1e-06,
                 PH2O = Cd * kR * T * 1e-06;
    const double kf = kin[0] * exp(-kin[1] / kR / T);
    const double ks = kin[2] * exp(-kin[3] / kR / T);
    const double kCO2 = kin[4] * exp(-kin[5] / kR / T);
    const double kH2O = kin[6] * exp(-kin[7] / kR / T);
    const double dco2 = 1 + kCO2 * PCO2, dh2o = 1 + kH2O * PH2O;
    const double rf = 5075e3 * kf * kCO2 * PCO2 * sqrt(py_max(0.001, PH2)) / (dco2 * dco2);
    const double rr = 5075e3 * ks * kH2O * PH2O * (PCH4 * PCH4) / (dh2o * dh2o);
    return rf - rr;
}

__device__ __forceinline__ double rohg(double a, double b, double c, double d, double e, double T, double P0) {
    return P0 / kR / T * (a * 2 + b * 44 + c * 16 + d * 18 + e * 40) / (a + b + c + d + e) * 0.001;
}

__device__ __forceinline__ double keff_term(double Tp, double Ti, double Tm, double dz2) {
    return kKeff * (Tp - 2 * Ti + Tm) / dz2;
}

// The 7 residual rows that belong to axial node i.  X, dX: pointers to a field-major state (any
// address space), p: the 18 parameters.  out[f] receives res[f*NX + i] for f = 0..6, EXCEPT at the last
// node where the reference writes the u-equation into the T slot and vice versa (:136-137): out keeps
// the reference's slot order there as well.
template <typename XP>
__device__ __forceinline__ void node_residual(int i, XP X, XP dX, const double *p, double *out) {
    const double T_in = p[5];
    const double P0 = p[0] * kR * T_in + p[1] * kR * T_in + p[2] * kR * T_in + p[3] * kR * T_in + p[4] * kR * T_in;
    const double T_jacket = p[6], u_in = p[7], vd = p[8], dz = p[9];
    auto C = [&](int f, int k) { return X[f * NX + k]; };
    auto T = [&](int k) { return X[5 * NX + k]; };
    auto u = [&](int k) { return X[6 * NX + k]; };
    if (i == 0) {
        for (int f = 0; f < 6; ++f) out[f] = dX[f * NX];
        out[6] = u(0) - u_in;
        return;
    }
    if (i == NX - 1) {
        for (int f = 0; f < 5; ++f) out[f] = C(f, i) - C(f, i - 1);
        out[5] = u(i) - u(i - 1);   // res[6*i+5] = res[305]: the T slot receives the u equation
        out[6] = T(i) - T(i - 1);   // res[7*i+6] = res[356]: the u slot receives the T equation
        return;
    }
    const double dz2 = dz * dz;
    const double Ti = T(i), Tm = T(i - 1), Tp = T(i + 1), ui = u(i), um = u(i - 1);
    const double r = rCH4(Ti, C(0, i), C(1, i), C(2, i), C(3, i), p + 10);
    const double sc[5] = {-4, -1, 1, 2, 0};
    for (int f = 0; f < 5; ++f) {
        const double ci = C(f, i), cm = C(f, i - 1), cp = C(f, i + 1);
        const double diff = (i == 1) ? (cp - ci) : (cp - 2 * ci + cm);
        out[f] = -vd * dX[f * NX + i] - (ui * ci - um * cm) / dz + vd * kDz * diff / dz2 + (1 - vd) * sc[f] * r;
    }
    const double rg = rohg(C(0, i), C(1, i), C(2, i), C(3, i), C(4, i), Ti, P0);
    const double dT = dX[5 * NX + i];
    const double tail5 = -ui * P0 * (1 / Ti - 1 / Tm) / dz - P0 / Ti * (ui - um) / dz +
                         vd * kDz * P0 * (1 / Tp - 2 / Ti + 1 / Tm) / dz2 + (1 - vd) * kR * (-2) * r;
    const double cap = vd * rg * kCpg + (1 - vd) * kRhos * kCps;
    const double tail6 = -rg * kCpg * (Ti * ui - Tm * um) / dz + keff_term(Tp, Ti, Tm, dz2) + (1 - vd) * (-kHr) * r -
                         2 * kU / kDint * (Ti - T_jacket);
    if (i == 1) {
        out[5] = P0 * vd * (1.0 / (Ti * Ti)) * dT - ui * P0 * (1 / Ti - 1 / Tm) / dz - P0 / Ti * (ui - um) / dz +
                 vd * kDz * P0 * (1 / Tp - 2 / Ti + 1 / Tm) / dz2 + (1 - vd) * kR * (-2) * r;
        out[6] = -cap * dT + tail6;
    } else {
        out[5] = tail5;
        out[6] = -0.1 * cap * dT + tail6;
    }
}

}  // namespace meth
}  // namespace smc
