// mm_rk45.h -- device functions: the Michaelis-Menten solve of one (particle, experiment) work
// item as a per-lane state machine: mm_item_begin() starts an item, mm_item_attempt() performs ONE
// adaptive Dormand-Prince RK45 step attempt (plus, on acceptance, the dense output at the data times
// that the step covers and their squared residuals).  gfx950 only.
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
// Why a state machine: SciPy's loops are nested (steps > attempts > outputs) and their trip counts
// differ wildly between particles (3..3000 attempts per solve over the prior).  On a 64-lane wave a
// nested formulation makes every lane wait for the slowest one.  Here one loop iteration of the wave
// is one ATTEMPT of every live lane: a lane whose attempt is rejected retries, a lane that finishes
// its item is handed the next item from the work queue (mm_kernels.hip), independent of its
// neighbours.  The arithmetic per item is exactly that of the nested loops.
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

__device__ __forceinline__ double mm_rhs(double S, double negVmax, double Km) {
    return (negVmax * S) / (Km + S);  // ((-Vmax)*S)/(Km+S), Micmem_likelihood.py:15
}
// Python's min(a,b)/max(a,b): keep a unless b is strictly better (NaN never is)
__device__ __forceinline__ double py_min(double a, double b) { return (b < a) ? b : a; }
__device__ __forceinline__ double py_max(double a, double b) { return (b > a) ? b : a; }
// 10*|nextafter(t,inf)-t| for t >= 0 (rk.py:120)
__device__ __forceinline__ double min_step_of(double t) {
    const double up = __longlong_as_double(__double_as_longlong(t) + 1);
    return 10.0 * fabs(up - t);
}
__device__ __forceinline__ double quiet_nan() { return __longlong_as_double(0x7ff8000000000000LL); }

// x ** -0.2 for x >= 0 (the step-size controller's error_norm ** error_exponent, rk.py:104,155,169).
// The generic pow() is ~200 dependent FP64 instructions - half of a whole RK45 attempt and therefore
// half of the serial critical path of a stiff solve (10^5 dependent attempts).  This is a dedicated
// inverse fifth root: x = m * 2^e with m in [0.5,1), e = 5q + r;  x^-0.2 = m^-0.2 * 2^(-r/5) * 2^-q.
// m^-0.2 starts from the hardware f32 log2/exp2 (about 22 good bits) and takes two Newton steps
// y <- y + y*(1 - m*y^5)/5 in FP64 (quadratic convergence: 22 -> 42 -> >53 bits).  Error <= 2 ulp,
// the same class as the libm-vs-device pow difference it replaces (the CPU checker under tests keeps libm pow).
__device__ __forceinline__ double pow_minus_fifth(double x) {
    if (!(x > 0.0)) return (x == 0.0) ? __longlong_as_double(0x7ff0000000000000LL) : quiet_nan();  // 0 -> inf
    if (x == __longlong_as_double(0x7ff0000000000000LL)) return 0.0;
    const double m = __builtin_amdgcn_frexp_mant(x);   // [0.5, 1)
    const int e = __builtin_amdgcn_frexp_exp(x);
    // floor division of e by 5 (e in [-1073, 1024]):  (e + 1075) / 5 - 215
    const int q = (e + 1075) / 5 - 215;
    const int r = e - 5 * q;                            // 0..4
    const float lf = __builtin_amdgcn_logf((float)m);   // log2(m), v_log_f32
    double y = (double)__builtin_amdgcn_exp2f(-0.2f * lf);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double y2 = y * y;
        const double y5 = (y2 * y2) * y;
        const double rho = fma(-m, y5, 1.0);
        y = fma(y * 0.2, rho, y);
    }
    // 2^(-r/5)
    double c = 1.0;
    c = (r == 1) ? 0.87055056329612413913627001747975 : c;
    c = (r == 2) ? 0.75785828325519900315210303617030 : c;
    c = (r == 3) ? 0.65975395538644709660394940471556 : c;
    c = (r == 4) ? 0.57434917749851750271711478780370 : c;
    return ldexp(y * c, -q);
}

// Live state of one lane's current item.  common.py:63-65: the RMS norm of a size-1 vector is
// sqrt(x*x)/1, which is |x| exactly in IEEE arithmetic (and where x*x over/underflows the step
// controller takes the same branch with the same factor), so norms are written as fabs().
struct MMItem {
    double negVmax, Km, S0;   // parameters of the item
    double t, y, f;           // solver state (rk.py: self.t, self.y, self.f)
    double h_abs, min_step;   // step size carried between attempts, min_step of the current step
    double t_bound;
    double t_next;            // s_t[t_off + i_out], or +inf when every data time has been served
    double sum_r2;            // running sum of squared residuals
    int t_off;                // offset of the experiment's row in the LDS tables
    int i_out;                // next data time to be served (ivp.py: t_eval_i)
    int attempts;
    bool rejected;            // a rejection happened in the current step (rk.py:131,171)
};

// Start an item: RungeKutta.__init__ (rk.py:96-104) incl. select_initial_step (common.py:68-134,
// direction +1, order 4, max_step inf) and the head of the first _step_impl (rk.py:120-127).
// Returns false when there is nothing to integrate (t0 == t_bound, base.py:181-187): all outputs
// are then already accumulated.
template <bool WRITE_PRED>
__device__ __forceinline__ bool mm_item_begin(MMItem &it, double Vmax, double Km, double S0, const double *s_t,
                                              const double *s_P, int t_off, int n_t, double rtol, double atol,
                                              double *pred) {
    it.negVmax = -Vmax;
    it.Km = Km;
    it.S0 = S0;
    it.t_off = t_off;
    const double t0 = s_t[t_off];
    it.t_bound = s_t[t_off + n_t - 1];
    it.t = t0;
    it.y = S0;
    it.f = mm_rhs(S0, it.negVmax, Km);
    it.sum_r2 = 0.0;
    it.i_out = 0;
    it.attempts = 0;
    it.rejected = false;
    const double interval = fabs(it.t_bound - t0);
    if (interval == 0.0) {
        it.h_abs = 0.0;
    } else {
        const double scale = atol + fabs(it.y) * rtol;
        const double d0 = fabs(it.y / scale), d1 = fabs(it.f / scale);
        double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : (0.01 * d0) / d1;
        h0 = py_min(h0, interval);
        const double y1 = it.y + h0 * it.f;
        const double f1 = mm_rhs(y1, it.negVmax, Km);
        const double d2 = fabs((f1 - it.f) / scale) / h0;
        double h1;
        if (d1 <= 1e-15 && d2 <= 1e-15)
            h1 = py_max(1e-6, h0 * 1e-3);
        else
            h1 = 1.0 / pow_minus_fifth(0.01 / py_max(d1, d2));  // x ** (1/5), common.py:130
        it.h_abs = py_min(py_min(100.0 * h0, h1), interval);
    }
    if (it.t == it.t_bound) {  // every t_eval <= t gets y
        while (it.i_out < n_t && s_t[t_off + it.i_out] <= it.t) {
            const double P_model = S0 - it.y;
            if (WRITE_PRED) pred[it.i_out] = P_model;
            const double r = s_P[t_off + it.i_out] - P_model;
            it.sum_r2 += r * r;
            ++it.i_out;
        }
        return false;
    }
    it.t_next = s_t[t_off];
    it.min_step = min_step_of(it.t);
    if (it.h_abs < it.min_step) it.h_abs = it.min_step;  // rk.py:122-127 (max_step = inf)
    return true;
}

// One step attempt.  Returns 0 while the item is still running, 1 when it finished (t reached
// t_bound), 2 when it failed (step size underflow, rk.py:133-134; SciPy status -1).
template <bool WRITE_PRED>
__device__ __forceinline__ int mm_item_attempt(MMItem &it, const double *s_t, const double *s_P, int n_t, double rtol,
                                               double atol, double *pred) {
    if (it.h_abs < it.min_step || it.attempts >= RK_MAX_ATTEMPTS) return 2;
    const double t = it.t, y = it.y, negVmax = it.negVmax, Km = it.Km;
    double t_new = t + it.h_abs;
    if (t_new - it.t_bound > 0) t_new = it.t_bound;
    const double h = t_new - t;
    it.h_abs = fabs(h);

    // ---- rk_step (rk.py:64-71), stages summed left to right ----
    const double k0 = it.f;
    const double k1 = mm_rhs(y + (k0 * A21) * h, negVmax, Km);
    const double k2 = mm_rhs(y + (k0 * A31 + k1 * A32) * h, negVmax, Km);
    const double k3 = mm_rhs(y + (k0 * A41 + k1 * A42 + k2 * A43) * h, negVmax, Km);
    const double k4 = mm_rhs(y + (k0 * A51 + k1 * A52 + k2 * A53 + k3 * A54) * h, negVmax, Km);
    const double k5 = mm_rhs(y + (k0 * A61 + k1 * A62 + k2 * A63 + k3 * A64 + k4 * A65) * h, negVmax, Km);
    const double y_new = y + h * (k0 * B1 + k2 * B3 + k3 * B4 + k4 * B5 + k5 * B6);
    const double k6 = mm_rhs(y_new, negVmax, Km);
    ++it.attempts;

    // ---- error norm (rk.py:106-110,146-147); np.maximum propagates NaN ----
    const double ay = fabs(y), ayn = fabs(y_new);
    const double scale = atol + ((ay > ayn || ay != ay) ? ay : ayn) * rtol;
    const double err = (k0 * E1 + k2 * E3 + k3 * E4 + k4 * E5 + k5 * E6 + k6 * E7) * h;
    const double error_norm = fabs(err / scale);

    // 0.9 * error_norm ** -0.2, needed by both branches of rk.py:149-171 (error_norm == 0 gives inf,
    // which min(MAX_FACTOR, .) turns into MAX_FACTOR exactly as the reference's special case does)
    const double pw = 0.9 * pow_minus_fifth(error_norm);

    if (!(error_norm < 1.0)) {
        it.h_abs *= py_max(0.2, pw);
        it.rejected = true;
        return 0;
    }
    double factor = py_min(10.0, pw);
    if (it.rejected) factor = py_min(1.0, factor);
    it.h_abs *= factor;

    const double t_old = t, y_old = y;
    it.t = t_new;
    it.y = y_new;
    it.f = k6;

    // ---- outputs with t_eval in (t_old, t] (ivp.py:700-720) by the quartic interpolant ----
    if (it.t_next <= t_new) {
        int i_out = it.i_out;
        const int base = it.t_off;
        double t_next = it.t_next;
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
        const double hd = t_new - t_old;  // RkDenseOutput.__init__ (rk.py:555)
        double sum_r2 = it.sum_r2;
        do {
            const double x = (t_next - t_old) / hd;
            const double p2 = x * x, p3 = p2 * x, p4 = p3 * x;  // cumprod
            const double S = hd * (Q0 * x + Q1 * p2 + Q2 * p3 + Q3 * p4) + y_old;
            const double P_model = it.S0 - S;
            if (WRITE_PRED) pred[i_out] = P_model;
            const double r = s_P[base + i_out] - P_model;
            sum_r2 += r * r;
            ++i_out;
            t_next = (i_out < n_t) ? s_t[base + i_out] : __longlong_as_double(0x7ff0000000000000LL);
        } while (t_next <= t_new);
        it.sum_r2 = sum_r2;
        it.i_out = i_out;
        it.t_next = t_next;
    }
    if (t_new - it.t_bound >= 0) return 1;  // base.py:196
    it.rejected = false;                    // head of the next _step_impl
    it.min_step = min_step_of(t_new);
    if (it.h_abs < it.min_step) it.h_abs = it.min_step;
    return 0;
}

}  // namespace smc
