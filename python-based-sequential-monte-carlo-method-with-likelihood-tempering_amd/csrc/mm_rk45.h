// mm_rk45.h -- device functions: Michaelis-Menten log-likelihood of one (particle, experiment)
// pair, i.e. one adaptive Dormand-Prince RK45 solve with dense output at the data times and the
// sum of squared residuals.  gfx950 only.
//
// Behaviour follows, operation by operation, what the reference executes per experiment:
//   SMC_example/Micmem_likelihood.py:14-15  mm_ode               dS/dt = -Vmax*S/(Km+S)
//   SMC_example/Micmem_likelihood.py:17-33  simulate_mm_on_grid  solve_ivp(RK45, t_eval=t), P = S0 - S
//   SMC_example/Micmem_likelihood.py:65-71  residual, sum(residual**2)
// and, inside solve_ivp (SciPy, third-party to the reference; behaviour of 1.15.x):
//   rk.py:14-71 rk_step, rk.py:111-176 step controller (SAFETY .9, MIN_FACTOR .2, MAX_FACTOR 10,
//   exponent -1/5, min_step = 10 ulp(t), factor <= 1 after a rejection, clip to t_bound),
//   common.py:68-134 select_initial_step, rk.py:178-180/552-574 quartic dense output,
//   ivp.py:700-720 t_eval dispatch (searchsorted side='right').
//
// Design for wave64: the step loop is FLATTENED - one loop iteration is one step attempt; a lane
// whose attempt is rejected simply retries while its neighbours commit and go on, so no lane waits
// for another lane's rejection.  The semantics are those of SciPy's nested loops.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace smc {

// Dormand-Prince coefficients (rk.py:377-404), the same double literals Python evaluates.
#define A21 (1.0 / 5)
#define A31 (3.0 / 40)
#define A32 (9.0 / 40)
#define A41 (44.0 / 45)
#define A42 (-56.0 / 15)
#define A43 (32.0 / 9)
#define A51 (19372.0 / 6561)
#define A52 (-25360.0 / 2187)
#define A53 (64448.0 / 6561)
#define A54 (-212.0 / 729)
#define A61 (9017.0 / 3168)
#define A62 (-355.0 / 33)
#define A63 (46732.0 / 5247)
#define A64 (49.0 / 176)
#define A65 (-5103.0 / 18656)
#define B1 (35.0 / 384)
#define B3 (500.0 / 1113)
#define B4 (125.0 / 192)
#define B5 (-2187.0 / 6784)
#define B6 (11.0 / 84)
#define E1 (-71.0 / 57600)
#define E3 (71.0 / 16695)
#define E4 (-71.0 / 1920)
#define E5 (17253.0 / 339200)
#define E6 (-22.0 / 525)
#define E7 (1.0 / 40)

#define RK_MAX_ATTEMPTS (1 << 20)  // hard bound so that every wave drains; SciPy has none

struct MMSolveResult {
    double sum_r2;   // sum over the n_t data times of (P_obs - P_model)^2
    int failed;      // 1 if the solve did not reach t_bound (SciPy: status -1, the reference raises)
    int attempts;    // RK45 step attempts (accepted + rejected)
};

__device__ __forceinline__ double mm_rhs(double S, double negVmax, double Km) {
    return (negVmax * S) / (Km + S);  // ((-Vmax)*S)/(Km+S), Micmem_likelihood.py:15
}
// Python's min(a,b)/max(a,b): keep a unless b is strictly better (NaN never is)
__device__ __forceinline__ double py_min(double a, double b) { return (b < a) ? b : a; }
__device__ __forceinline__ double py_max(double a, double b) { return (b > a) ? b : a; }
// 10*|nextafter(t,inf)-t| for t >= 0 (rk.py:120)
__device__ __forceinline__ double min_step_of(double t) {
    double up = __longlong_as_double(__double_as_longlong(t) + 1);
    return 10.0 * fabs(up - t);
}

// t_e / P_e: the experiment's data times and observations (LDS).  pred (optional, global) receives
// P_model at the n_t data times.
template <bool WRITE_PRED>
__device__ __forceinline__ MMSolveResult mm_solve_experiment(double Vmax, double Km, double S0, const double *t_e,
                                                             const double *P_e, int n_t, double rtol, double atol,
                                                             double *pred) {
    MMSolveResult res;
    res.failed = 0;
    res.attempts = 0;
    const double negVmax = -Vmax;
    const double t0 = t_e[0], t_bound = t_e[n_t - 1];
    double t = t0, y = S0;
    double f = mm_rhs(y, negVmax, Km);
    double sum_r2 = 0.0;
    int i_out = 0;

    // ---- select_initial_step (common.py:68-134): direction +1, order 4, max_step inf ----
    double h_abs;
    {
        const double interval = fabs(t_bound - t0);
        if (interval == 0.0) {
            h_abs = 0.0;
        } else {
            const double scale = atol + fabs(y) * rtol;
            const double q0 = y / scale, q1 = f / scale;
            const double d0 = sqrt(q0 * q0), d1 = sqrt(q1 * q1);
            double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : (0.01 * d0) / d1;
            h0 = py_min(h0, interval);
            const double y1 = y + h0 * f;
            const double f1 = mm_rhs(y1, negVmax, Km);
            const double q2 = (f1 - f) / scale;
            const double d2 = sqrt(q2 * q2) / h0;
            double h1;
            if (d1 <= 1e-15 && d2 <= 1e-15)
                h1 = py_max(1e-6, h0 * 1e-3);
            else
                h1 = pow(0.01 / py_max(d1, d2), 0.2);
            h_abs = py_min(py_min(100.0 * h0, h1), interval);
        }
    }

    bool active = true;
    if (t == t_bound) {  // base.py:181-187: nothing to integrate; every t_eval <= t gets y
        while (i_out < n_t && t_e[i_out] <= t) {
            const double P_model = S0 - y;
            if (WRITE_PRED) pred[i_out] = P_model;
            const double r = P_e[i_out] - P_model;
            sum_r2 += r * r;
            ++i_out;
        }
        active = false;
    }

    double min_step = min_step_of(t);
    if (h_abs < min_step) h_abs = min_step;  // rk.py:122-127 (max_step = inf)
    bool rejected = false;

    while (active) {
        if (h_abs < min_step || res.attempts >= RK_MAX_ATTEMPTS) {  // rk.py:133-134 TOO_SMALL_STEP
            res.failed = 1;
            break;
        }
        double t_new = t + h_abs;
        if (t_new - t_bound > 0) t_new = t_bound;
        const double h = t_new - t;
        h_abs = fabs(h);

        // ---- rk_step (rk.py:64-71), stages summed left to right ----
        const double k0 = f;
        const double k1 = mm_rhs(y + (k0 * A21) * h, negVmax, Km);
        const double k2 = mm_rhs(y + (k0 * A31 + k1 * A32) * h, negVmax, Km);
        const double k3 = mm_rhs(y + (k0 * A41 + k1 * A42 + k2 * A43) * h, negVmax, Km);
        const double k4 = mm_rhs(y + (k0 * A51 + k1 * A52 + k2 * A53 + k3 * A54) * h, negVmax, Km);
        const double k5 = mm_rhs(y + (k0 * A61 + k1 * A62 + k2 * A63 + k3 * A64 + k4 * A65) * h, negVmax, Km);
        const double y_new = y + h * (k0 * B1 + k2 * B3 + k3 * B4 + k4 * B5 + k5 * B6);
        const double k6 = mm_rhs(y_new, negVmax, Km);
        ++res.attempts;

        // ---- error norm (rk.py:106-110,146-147) ----
        const double ay = fabs(y), ayn = fabs(y_new);
        double scale = atol + ((ay > ayn || ay != ay) ? ay : ayn) * rtol;  // np.maximum (NaN-propagating)
        if (ayn != ayn) scale = ayn;
        const double err = (k0 * E1 + k2 * E3 + k3 * E4 + k4 * E5 + k5 * E6 + k6 * E7) * h;
        const double q = err / scale;
        const double error_norm = sqrt(q * q);

        if (error_norm < 1.0) {
            double factor;
            if (error_norm == 0.0)
                factor = 10.0;
            else
                factor = py_min(10.0, 0.9 * pow(error_norm, -0.2));
            if (rejected) factor = py_min(1.0, factor);
            h_abs *= factor;

            const double t_old = t, y_old = y;
            t = t_new;
            y = y_new;
            f = k6;

            // ---- outputs with t_eval in (t_old, t] (ivp.py:700-720) by the quartic interpolant ----
            if (i_out < n_t && t_e[i_out] <= t) {
                // Q = K.T.dot(P) (rk.py:179); P[1][:] = 0 and P[j][0] = 0 for j > 0
                const double Q0 = k0;
                const double Q1 = k0 * (-8048581381.0 / 2820520608) + k2 * (131558114200.0 / 32700410799) +
                                  k3 * (-1754552775.0 / 470086768) + k4 * (127303824393.0 / 49829197408) +
                                  k5 * (-282668133.0 / 205662961) + k6 * (40617522.0 / 29380423);
                const double Q2 = k0 * (8663915743.0 / 2820520608) + k2 * (-68118460800.0 / 10900136933) +
                                  k3 * (14199869525.0 / 1410260304) + k4 * (-318862633887.0 / 49829197408) +
                                  k5 * (2019193451.0 / 616988883) + k6 * (-110615467.0 / 29380423);
                const double Q3 = k0 * (-12715105075.0 / 11282082432) + k2 * (87487479700.0 / 32700410799) +
                                  k3 * (-10690763975.0 / 1880347072) + k4 * (701980252875.0 / 199316789632) +
                                  k5 * (-1453857185.0 / 822651844) + k6 * (69997945.0 / 29380423);
                const double hd = t - t_old;  // RkDenseOutput.__init__ (rk.py:555)
                do {
                    const double x = (t_e[i_out] - t_old) / hd;
                    const double p2 = x * x, p3 = p2 * x, p4 = p3 * x;  // cumprod
                    const double S = hd * (Q0 * x + Q1 * p2 + Q2 * p3 + Q3 * p4) + y_old;
                    const double P_model = S0 - S;
                    if (WRITE_PRED) pred[i_out] = P_model;
                    const double r = P_e[i_out] - P_model;
                    sum_r2 += r * r;
                    ++i_out;
                } while (i_out < n_t && t_e[i_out] <= t);
            }
            if (t - t_bound >= 0) {  // base.py:196
                active = false;
            } else {  // head of the next _step_impl
                rejected = false;
                min_step = min_step_of(t);
                if (h_abs < min_step) h_abs = min_step;
            }
        } else {
            h_abs *= py_max(0.2, 0.9 * pow(error_norm, -0.2));
            rejected = true;
        }
    }
    if (res.failed || i_out != n_t) {
        res.failed = 1;
        sum_r2 = __longlong_as_double(0x7ff8000000000000LL);
        if (WRITE_PRED)
            for (int i = i_out; i < n_t; ++i) pred[i] = sum_r2;
    }
    res.sum_r2 = sum_r2;
    return res;
}

}  // namespace smc
