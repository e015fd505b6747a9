/*
 * oracle/methanation_oracle.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, IEEE double, no FMA contraction) of the pure-arithmetic part of the
 * reference's methanation model (paths relative to /root/reference/SMC_methanation):
 *
 *   meth_rCH4       methanation_set_likelihood.py:44-58   func_rCH4  (rate law)
 *   meth_rohg       methanation_set_likelihood.py:61-66   func_rohg  (gas density)
 *   meth_reaction   methanation_set_likelihood.py:69-139  reaction   (357-equation DAE residual)
 *   meth_loglike    methanation_set_likelihood.py:280-300 my_loglike
 *   meth_flows      methanation_set_likelihood.py:204-208 outlet node -> standard-state flows
 *
 * Parity status: PINNED at this level - tests/test_methanation_oracle.py checks every function against
 * tests/golden/methanation_golden.npz, which tests/golden/make_methanation_golden.py produced by calling
 * the reference's own functions (the reference's inlet table is missing from its repository, so the
 * conditions are a synthetic table committed as a fixture).
 * The TIME INTEGRATION (my_model -> Assimulo/SUNDIALS IDA, :144-277) is NOT restated here: Assimulo is
 * absent from the image and no output of it exists for known inputs -> "parity unpinned".
 *
 * Expression order follows Python's left-to-right evaluation of the reference's statements; x**2,
 * x**(-2), x**0.5 are pow() as Python/NumPy scalars evaluate them.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#define NX 51

/* methanation_set_conditon.py:74-89 */
static const double SC[5] = {-4, -1, 1, 2, 0};
static const double Dz = 0.95e-5, rhos = 5075, Hr = -164940, R = 8.3144589, Cpg = 2800, Cps = 698, keff = 0.72,
                    dint = 0.005, U = 68.2480;

static inline double py_max(double a, double b) { return (b > a) ? b : a; }

/* :44-58 */
double meth_rCH4(double T, double Ca, double Cb, double Cc, double Cd, const double *params) {
    double PH2 = Ca * R * T * 1e-06;
    double PCO2 = Cb * R * T * 1e-06;
    double PCH4 = Cc * R * T * 1e-06;
    double PH2O = Cd * R * T * 1e-06;
    double kf = params[0] * exp(-params[1] / R / T);
    double ks = params[2] * exp(-params[3] / R / T);
    double kCO2 = params[4] * exp(-params[5] / R / T);
    double kH2O = params[6] * exp(-params[7] / R / T);
    double rf = 5075e3 * kf * kCO2 * PCO2 * pow(py_max(0.001, PH2), 0.5) / pow(1 + kCO2 * PCO2, 2);
    double rr = 5075e3 * ks * kH2O * PH2O * pow(PCH4, 2) / pow(1 + kH2O * PH2O, 2);
    return rf - rr;
}

/* :61-66 */
double meth_rohg(double a, double b, double c, double d, double e, double T, double P0) {
    return P0 / R / T * (a * 2 + b * 44 + c * 16 + d * 18 + e * 40) / (a + b + c + d + e) * 0.001;
}

/* :69-139.  X, dX: 357 (field-major: Ca,Cb,Cc,Cd,Ce,T,u x 51 nodes); params: 18; res: 357. */
void meth_reaction(const double *X, const double *dX, const double *params, double *res) {
    const double T_in = params[5];
    const double Pa_in = params[0] * R * T_in, Pb_in = params[1] * R * T_in, Pc_in = params[2] * R * T_in,
                 Pd_in = params[3] * R * T_in, Pe_in = params[4] * R * T_in;
    const double T_jacket = params[6], u_in = params[7], vd = params[8], dz = params[9];
    const double *kin = params + 10;
    const double *Ca = X, *Cb = X + NX, *Cc = X + 2 * NX, *Cd = X + 3 * NX, *Ce = X + 4 * NX, *T = X + 5 * NX,
                 *u = X + 6 * NX;
    const double *C[5] = {Ca, Cb, Cc, Cd, Ce};
    const double P0 = Pa_in + Pb_in + Pc_in + Pd_in + Pe_in;
    for (int k = 0; k < 7 * NX; ++k) res[k] = 0.0;
    /* :96-102 */
    for (int f = 0; f < 6; ++f) res[f * NX] = dX[f * NX];
    res[6 * NX] = u[0] - u_in;
    const double dz2 = pow(dz, 2);
    /* :104-111 (node 1) and :114-126 (nodes 2..49) */
    for (int i = 1; i < NX - 1; ++i) {
        const double r = meth_rCH4(T[i], Ca[i], Cb[i], Cc[i], Cd[i], kin);
        for (int f = 0; f < 5; ++f) {
            const double *c = C[f];
            double diff = (i == 1) ? (c[i + 1] - c[i]) : (c[i + 1] - 2 * c[i] + c[i - 1]);
            res[f * NX + i] = -vd * dX[f * NX + i] - (u[i] * c[i] - u[i - 1] * c[i - 1]) / dz + vd * Dz * diff / dz2 +
                              (1 - vd) * SC[f] * r;
        }
        const double rg = meth_rohg(Ca[i], Cb[i], Cc[i], Cd[i], Ce[i], T[i], P0);
        const double common5 = -u[i] * P0 * (1 / T[i] - 1 / T[i - 1]) / dz - P0 / T[i] * (u[i] - u[i - 1]) / dz +
                               vd * Dz * P0 * (1 / T[i + 1] - 2 / T[i] + 1 / T[i - 1]) / dz2 + (1 - vd) * R * (-2) * r;
        if (i == 1) {
            /* Python evaluates  A*dX - B - C + D + E  left to right: the dX term comes first */
            res[5 * NX + i] = P0 * vd * pow(T[i], -2) * dX[5 * NX + i] - u[i] * P0 * (1 / T[i] - 1 / T[i - 1]) / dz -
                              P0 / T[i] * (u[i] - u[i - 1]) / dz +
                              vd * Dz * P0 * (1 / T[i + 1] - 2 / T[i] + 1 / T[i - 1]) / dz2 + (1 - vd) * R * (-2) * r;
            res[6 * NX + i] = -(vd * rg * Cpg + (1 - vd) * rhos * Cps) * dX[5 * NX + i] -
                              rg * Cpg * (T[i] * u[i] - T[i - 1] * u[i - 1]) / dz +
                              keff * (T[i + 1] - 2 * T[i] + T[i - 1]) / dz2 + (1 - vd) * (-Hr) * r -
                              2 * U / dint * (T[i] - T_jacket);
        } else {
            res[5 * NX + i] = common5;
            res[6 * NX + i] = -0.1 * (vd * rg * Cpg + (1 - vd) * rhos * Cps) * dX[5 * NX + i] -
                              rg * Cpg * (T[i] * u[i] - T[i - 1] * u[i - 1]) / dz +
                              keff * (T[i + 1] - 2 * T[i] + T[i - 1]) / dz2 + (1 - vd) * (-Hr) * r -
                              2 * U / dint * (T[i] - T_jacket);
        }
    }
    /* :130-137, i = NX-1 = 50: the index expressions 2*i+1 ... 7*i+6 hit 101,152,203,254,305,356 */
    {
        const int i = NX - 1;
        res[i] = Ca[i] - Ca[i - 1];
        res[2 * i + 1] = Cb[i] - Cb[i - 1];
        res[3 * i + 2] = Cc[i] - Cc[i - 1];
        res[4 * i + 3] = Cd[i] - Cd[i - 1];
        res[5 * i + 4] = Ce[i] - Ce[i - 1];
        res[6 * i + 5] = u[i] - u[i - 1];
        res[7 * i + 6] = T[i] - T[i - 1];
    }
}

void meth_reaction_batch(const double *X, const double *dX, const double *params, int64_t n, double *res) {
    for (int64_t b = 0; b < n; ++b) meth_reaction(X + b * 7 * NX, dX + b * 7 * NX, params + b * 18, res + b * 7 * NX);
}

/* NumPy pairwise sum for n <= 128 (8 accumulators), as in smc_oracle.c */
static double np_sum_small(const double *a, int n) {
    if (n < 8) {
        double r = 0.;
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    double r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

/* :280-300.  y, data: 5 x n_data row-major (n_data <= 128). */
double meth_loglike(const double *y, const double *data, double sigma, int n_data) {
    double parts[5], tmp[128];
    for (int i = 0; i < 5; ++i) {
        for (int k = 0; k < n_data; ++k) {
            double d = y[i * n_data + k] - data[i * n_data + k];
            tmp[k] = d * d;
        }
        double s = np_sum_small(tmp, n_data);
        parts[i] = -(0.5 / pow(sigma, 2)) * s - n_data * log(sigma);
    }
    return np_sum_small(parts, 5);
}

/* :204-208: outlet node of the last time point -> 5 standard-state flows.
 * y: the 357-state; P_total = (p0[0..4] summed)*R*p0[5] (:165); S = pi*Rr^2, P_stp (:81,89). */
void meth_flows(const double *y, double P_total, double S, double P_stp, double *F) {
    const double u = y[7 * NX - 1], T = y[6 * NX - 1];
    for (int f = 0; f < 5; ++f) {
        const double c = y[(f + 1) * NX - 1];
        F[f] = c * S * u * 60 * R * T / (P_total) * 1e6 * (P_total) / P_stp * 298 / T; /* 10**6 is an int in Python: exact */
    }
}

int meth_oracle_abi_version(void) { return 1; }
