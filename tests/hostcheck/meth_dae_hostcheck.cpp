// TEST TOOLING (tests/ only): compiles the product's host/device-portable DAE integrator source
// (csrc/meth_dae.h) with g++ for the CPU, so that the analytic iteration matrix and the BDF control logic can
// be unit-tested without a GPU.  It is not part of libsmc_hip.so and no product code path can reach it.
#include <cstring>
#include <vector>

#include "../../python-based-sequential-monte-carlo-method-with-likelihood-tempering_amd/csrc/meth_dae.h"

using namespace smc::meth;

extern "C" {

// residual in the REFERENCE's slot order (field-major res[f*51+i]) computed by node_eval
void hc_residual(const double *y, const double *yd, const double *p, double *res) {
    for (int i = 0; i < kNX; ++i) {
        double wm[7] = {0}, w0[7], wp[7] = {0}, yd0[7], r[7];
        for (int f = 0; f < 7; ++f) {
            w0[f] = y[f * kNX + i];
            if (i > 0) wm[f] = y[f * kNX + i - 1];
            if (i < kNX - 1) wp[f] = y[f * kNX + i + 1];
            yd0[f] = yd[f * kNX + i];
        }
        node_eval<false>(i, wm, w0, wp, yd0, p, 0.0, r, nullptr, nullptr, nullptr);
        for (int f = 0; f < 5; ++f) res[f * kNX + i] = r[f];
        if (i == 0) { res[5 * kNX] = r[5]; res[6 * kNX] = r[6]; }
        else { res[6 * kNX + i] = r[5]; res[5 * kNX + i] = r[6]; }   // undo the row swap
    }
}

// dense iteration matrix dF/dy + cj dF/dy' in solver ordering (node-major unknowns, swapped rows), 357 x 357
void hc_itermatrix(const double *y, const double *yd, const double *p, double cj, double *A) {
    std::memset(A, 0, sizeof(double) * kNS * kNS);
    for (int i = 0; i < kNX; ++i) {
        double wm[7] = {0}, w0[7], wp[7] = {0}, yd0[7], r[7], L[kNB], D[kNB], U[kNB];
        for (int f = 0; f < 7; ++f) {
            w0[f] = y[f * kNX + i];
            if (i > 0) wm[f] = y[f * kNX + i - 1];
            if (i < kNX - 1) wp[f] = y[f * kNX + i + 1];
            yd0[f] = yd[f * kNX + i];
        }
        node_eval<true>(i, wm, w0, wp, yd0, p, cj, r, L, D, U);
        for (int rr = 0; rr < 7; ++rr)
            for (int c = 0; c < 7; ++c) {
                if (i > 0) A[(7 * i + rr) * kNS + 7 * (i - 1) + c] = L[rr * 7 + c];
                A[(7 * i + rr) * kNS + 7 * i + c] = D[rr * 7 + c];
                if (i < kNX - 1) A[(7 * i + rr) * kNS + 7 * (i + 1) + c] = U[rr * 7 + c];
            }
    }
}

int hc_integrate(const double *y0, const double *p, double tf, double rtol, double atol, double h0, double *y_out,
                 int *stats /* steps, rejects, newton_fail, nlu, newton_iters, status */) {
    std::vector<double> buf(kWsDoubles, 0.0);
    Ws ws{buf.data(), 1};
    for (int x = 0; x < kNS; ++x) ws(OFF_D + x) = y0[x];
    DaeStats st;
    dae_integrate(ws, p, tf, rtol, atol, h0, 200000, st);
    for (int x = 0; x < kNS; ++x) y_out[x] = ws(OFF_D + x);
    stats[0] = st.steps; stats[1] = st.rejects; stats[2] = st.newton_fail; stats[3] = st.nlu; stats[4] = st.newton_iters;
    stats[5] = st.status;
    return st.status;
}

void hc_flows(const double *y, const double *p, double S, double P_stp, double *F) {
    std::vector<double> buf(kWsDoubles, 0.0);
    Ws ws{buf.data(), 1};
    for (int x = 0; x < kNS; ++x) ws(OFF_D + x) = y[x];
    outlet_flows(ws, p, S, P_stp, F);
}
}
