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

#include "pow_fifth_exact.h"
#define SMC_HAVE_POW_FIFTH_EXACT 1
#include "rk45_math.h"

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

template <int DIV>
__device__ __forceinline__ double mm_div(double a, double b) {
    return DIV == kDivLean6 ? lean_div6(a, b) : DIV == kDivLean5 ? lean_div5(a, b) : a / b;
}
template <int DIV>
__device__ __forceinline__ double mm_rhs_t(double S, double negVmax, double Km) {
    return mm_div<DIV>(negVmax * S, Km + S);  // ((-Vmax)*S)/(Km+S), Micmem_likelihood.py:15
}
__device__ __forceinline__ double mm_rhs(double S, double negVmax, double Km) { return mm_rhs_t<kDivIeee>(S, negVmax, Km); }

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
    int t_off;                // offset of the experiment's row in the LDS table (mm_table_row)
    int i_out;                // next data time to be served (ivp.py: t_eval_i)
    int attempts;
    bool rejected;            // a rejection happened in the current step (rk.py:131,171)
};

// The data of the experiments in LDS: one row of n_t + 1 (time, P_obs) pairs per experiment, the last pair being the
// sentinel (+inf, 0).  The dense-output loop fetches the next time and its observation with ONE ds_read_b128 and needs
// no test for the end of the row: in the posterior phase a third of the solve kernel's vector instructions are that loop
// (PMC counts), executed under a partial exec mask.
__device__ __forceinline__ int mm_table_row(int e, int n_t) { return e * (n_t + 1); }
__device__ __forceinline__ void mm_table_fill(double2 *s_tp, const double *t, const double *P_obs, int n_ex, int n_t, int tid,
                                              int n_threads) {
    for (int i = tid; i < n_ex * (n_t + 1); i += n_threads) {
        const int e = i / (n_t + 1), k = i - e * (n_t + 1);
        s_tp[i] = (k < n_t) ? make_double2(t[e * n_t + k], P_obs[e * n_t + k])
                            : make_double2(__longlong_as_double(0x7ff0000000000000LL), 0.0);
    }
}

// Start an item: RungeKutta.__init__ (rk.py:96-104) incl. select_initial_step (common.py:68-134,
// direction +1, order 4, max_step inf) and the head of the first _step_impl (rk.py:120-127).
// Returns false when there is nothing to integrate (t0 == t_bound, base.py:181-187): all outputs
// are then already accumulated.
template <bool WRITE_PRED, bool EXACT = false>
__device__ __forceinline__ bool mm_item_begin(MMItem &it, double Vmax, double Km, double S0, const double2 *s_tp,
                                              int t_off, int n_t, double rtol, double atol, double *pred) {
    it.negVmax = -Vmax;
    it.Km = Km;
    it.S0 = S0;
    it.t_off = t_off;
    const double t0 = s_tp[t_off].x;
    it.t_bound = s_tp[t_off + n_t - 1].x;
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
        const double scale = EXACT ? __dadd_rn(atol, __dmul_rn(fabs(it.y), rtol)) : atol + fabs(it.y) * rtol;
        const double d0 = fabs(it.y / scale), d1 = fabs(it.f / scale);
        double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : (0.01 * d0) / d1;
        h0 = py_min(h0, interval);
        const double y1 = EXACT ? __dadd_rn(it.y, __dmul_rn(h0, it.f)) : it.y + h0 * it.f;
        const double f1 = mm_rhs(y1, it.negVmax, Km);
        const double d2 = fabs((f1 - it.f) / scale) / h0;
        double h1;
        if (d1 <= 1e-15 && d2 <= 1e-15)
            h1 = py_max(1e-6, h0 * 1e-3);
        else
            h1 = pow_plus_fifth<EXACT>(0.01 / py_max(d1, d2));  // x ** (1/5), common.py:130
        it.h_abs = py_min(py_min(100.0 * h0, h1), interval);
    }
    if (it.t == it.t_bound) {  // every t_eval <= t gets y
        while (it.i_out < n_t && s_tp[t_off + it.i_out].x <= it.t) {
            const double P_model = S0 - it.y;
            if (WRITE_PRED) pred[it.i_out] = P_model;
            const double r = s_tp[t_off + it.i_out].y - P_model;
            it.sum_r2 += r * r;
            ++it.i_out;
        }
        return false;
    }
    it.t_next = t0;
#ifdef SMC_PROBE_NO_OUTPUTS   // timing probe only (wrong results): no data time is ever due, the dense-output path never runs
    it.t_next = __longlong_as_double(0x7ff0000000000000LL);
#endif
    it.min_step = min_step_of(it.t);
    if (it.h_abs < it.min_step) it.h_abs = it.min_step;  // rk.py:122-127 (max_step = inf)
    return true;
}

// rk_step (rk.py:64-71, stages summed left to right) + error norm (rk.py:106-110,146-147; np.maximum
// propagates NaN) of one attempt: a pure function of the lane's state.
struct RkStages {
    double k1, k2, k3, k4, k5, k6, y_new, error_norm;
};
template <int LEAN /* kDiv* */>
__device__ __forceinline__ RkStages rk_attempt_core(double y, double k0, double h, double negVmax, double Km,
                                                    double rtol, double atol) {
    RkStages s;
    s.k1 = mm_rhs_t<LEAN>(y + (k0 * A21) * h, negVmax, Km);
    s.k2 = mm_rhs_t<LEAN>(y + (k0 * A31 + s.k1 * A32) * h, negVmax, Km);
    s.k3 = mm_rhs_t<LEAN>(y + (k0 * A41 + s.k1 * A42 + s.k2 * A43) * h, negVmax, Km);
    s.k4 = mm_rhs_t<LEAN>(y + (k0 * A51 + s.k1 * A52 + s.k2 * A53 + s.k3 * A54) * h, negVmax, Km);
    s.k5 = mm_rhs_t<LEAN>(y + (k0 * A61 + s.k1 * A62 + s.k2 * A63 + s.k3 * A64 + s.k4 * A65) * h, negVmax, Km);
    s.y_new = y + h * (k0 * B1 + s.k2 * B3 + s.k3 * B4 + s.k4 * B5 + s.k5 * B6);
    s.k6 = mm_rhs_t<LEAN>(s.y_new, negVmax, Km);
    // np.maximum(|y|, |y_new|) propagates NaN, v_max_f64 returns the other operand - and it makes no difference to the one
    // value that is used, error_norm: a NaN y makes k0, a NaN y_new makes k6 and with it err NaN whatever the scale is.  One
    // instruction instead of two compares, an or and two selects, in every attempt.
    const double scale = atol + fmax(fabs(y), fabs(s.y_new)) * rtol;
    const double err = (k0 * E1 + s.k2 * E3 + s.k3 * E4 + s.k4 * E5 + s.k5 * E6 + s.k6 * E7) * h;
    s.error_norm = fabs(mm_div<LEAN>(err, scale));
    return s;
}
// The same attempt with EVERY product and sum rounded separately (no fused multiply-add), in the order NumPy executes
// rk_step's dot products (rk.py:64-71: dy = K[:s].T @ a[s, :s] * h, left to right) - parity mode (smc_set_exact_pow).  With
// FMA contraction the stage values differ from the CPU checker's (compiled -ffp-contract=off) in the last bit of nearly
// every attempt, which is harmless except on RK45's stability limit, where one such bit eventually flips one accept / reject
// decision of the controller.  Separately rounded, the device walks the checker's step sequence attempt for attempt.
// (b3 = e2 = 0: the zero products of the dot product change nothing and are left out.)
#define SMC_M(a, b) __dmul_rn((a), (b))
#define SMC_A(a, b) __dadd_rn((a), (b))
__device__ __forceinline__ RkStages rk_attempt_core_exact(double y, double k0, double h, double negVmax, double Km,
                                                          double rtol, double atol) {
    RkStages s;
    s.k1 = mm_rhs(SMC_A(y, SMC_M(SMC_M(k0, A21), h)), negVmax, Km);
    s.k2 = mm_rhs(SMC_A(y, SMC_M(SMC_A(SMC_M(k0, A31), SMC_M(s.k1, A32)), h)), negVmax, Km);
    s.k3 = mm_rhs(SMC_A(y, SMC_M(SMC_A(SMC_A(SMC_M(k0, A41), SMC_M(s.k1, A42)), SMC_M(s.k2, A43)), h)), negVmax, Km);
    s.k4 = mm_rhs(SMC_A(y, SMC_M(SMC_A(SMC_A(SMC_A(SMC_M(k0, A51), SMC_M(s.k1, A52)), SMC_M(s.k2, A53)), SMC_M(s.k3, A54)), h)),
                  negVmax, Km);
    s.k5 = mm_rhs(SMC_A(y, SMC_M(SMC_A(SMC_A(SMC_A(SMC_A(SMC_M(k0, A61), SMC_M(s.k1, A62)), SMC_M(s.k2, A63)), SMC_M(s.k3, A64)),
                                       SMC_M(s.k4, A65)), h)), negVmax, Km);
    s.y_new = SMC_A(y, SMC_M(h, SMC_A(SMC_A(SMC_A(SMC_A(SMC_M(k0, B1), SMC_M(s.k2, B3)), SMC_M(s.k3, B4)), SMC_M(s.k4, B5)),
                                      SMC_M(s.k5, B6))));
    s.k6 = mm_rhs(s.y_new, negVmax, Km);
    const double scale = SMC_A(atol, SMC_M(fmax(fabs(y), fabs(s.y_new)), rtol));   // see rk_attempt_core on the NaN case
    const double err = SMC_M(SMC_A(SMC_A(SMC_A(SMC_A(SMC_A(SMC_M(k0, E1), SMC_M(s.k2, E3)), SMC_M(s.k3, E4)), SMC_M(s.k4, E5)),
                                         SMC_M(s.k5, E6)), SMC_M(s.k6, E7)), h);
    s.error_norm = fabs(err / scale);
    return s;
}

// The outputs with t_eval in (t_old, t_new] of an accepted step (ivp.py:700-720, rk.py:561-574): quartic interpolant,
// residual against the observation, running sum.  LEAN_X: see the caller.
template <bool WRITE_PRED, bool LEAN_X>
__device__ __forceinline__ void mm_dense_outputs(MMItem &it, const double2 *s_tp, double t_old, double t_new, double hd,
                                                 double y_old, double Q0, double Q1, double Q2, double Q3, double *pred) {
    int i_out = it.i_out;
    const int base = it.t_off;
    double t_next = it.t_next;
    double sum_r2 = it.sum_r2;
    double P_obs = s_tp[base + i_out].y;
    do {
        const double num = t_next - t_old;
        const double x = LEAN_X ? lean_div6(num, hd) : num / hd;
        const double p2 = x * x, p3 = p2 * x, p4 = p3 * x;  // cumprod
        const double S = hd * (Q0 * x + Q1 * p2 + Q2 * p3 + Q3 * p4) + y_old;
        const double P_model = it.S0 - S;
        if (WRITE_PRED) pred[i_out] = P_model;
        const double r = P_obs - P_model;
        sum_r2 += r * r;
        ++i_out;
        const double2 nx = s_tp[base + i_out];   // i_out == n_t reads the sentinel (+inf, 0)
        t_next = nx.x;
        P_obs = nx.y;
    } while (t_next <= t_new);
    it.sum_r2 = sum_r2;
    it.i_out = i_out;
    it.t_next = t_next;
}

// The dense output of an accepted step that covers data times: Q = K.T.dot(P) (rk.py:179), then mm_dense_outputs.
template <bool WRITE_PRED, int DIV>
__device__ __forceinline__ void mm_attempt_outputs(MMItem &it, const double2 *s_tp, const RkStages &st, double k0, double t_old,
                                                   double t_new, double y_old, double *pred) {
    const double k2 = st.k2, k3 = st.k3, k4 = st.k4, k5 = st.k5, k6 = st.k6;
    // P[1][:] = 0 and P[j][0] = 0 for j > 0
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
    // x = (t_eval - t_old) / h (rk.py:566) by the six-operation division when its operands are safely inside the
    // normal range: h itself, and the numerator, which is either 0 (t_eval == t_old, first step only), or t_eval
    // (t_old == 0: smc_set_model_mm admits only data times that are 0 or >= 2^-400 in magnitude), or at least one
    // ulp of |t_old| >= 2^-400.  Two copies of the loop rather than a select per output.
    const bool lean_x = (DIV != kDivIeee) && hd >= 0x1p-400 && hd <= 0x1p400 && (t_old == 0.0 || fabs(t_old) >= 0x1p-400);
    if (lean_x)
        mm_dense_outputs<WRITE_PRED, true>(it, s_tp, t_old, t_new, hd, y_old, Q0, Q1, Q2, Q3, pred);
    else
        mm_dense_outputs<WRITE_PRED, false>(it, s_tp, t_old, t_new, hd, y_old, Q0, Q1, Q2, Q3, pred);
}

// One step attempt.  Returns 0 while the item is still running, 1 when it finished (t reached
// t_bound), 2 when it failed (step size underflow, rk.py:133-134; SciPy status -1).
template <bool WRITE_PRED, int DIV = kDivLean6, bool EXACT = false>
__device__ __forceinline__ int mm_item_attempt(MMItem &it, const double2 *s_tp, int n_t, double rtol, double atol,
                                               double *pred) {
    // rk.py:133-134 TOO_SMALL_STEP (plus the hard attempt bound): tested together with the other rare
    // conditions in the single branch below; the stages computed meanwhile are simply discarded
    const bool fail = it.h_abs < it.min_step || it.attempts >= RK_MAX_ATTEMPTS;
    const double t = it.t, y = it.y, negVmax = it.negVmax, Km = it.Km;
    // rk.py:137-141: t_new = t + h_abs, clipped to t_bound - a minimum (v_min_f64 instead of a compare and two selects; h_abs is
    // never NaN for finite parameters: the controller's factors come out of Python's min / max, which drop a NaN)
    const double t_new = fmin(t + it.h_abs, it.t_bound);
    const double h = t_new - t;
    it.h_abs = fabs(h);

    const double k0 = it.f;
    RkStages st = EXACT ? rk_attempt_core_exact(y, k0, h, negVmax, Km, rtol, atol) : rk_attempt_core<DIV>(y, k0, h, negVmax, Km, rtol, atol);
    ++it.attempts;

    // 0.9 * error_norm ** -0.2 is needed by both branches of rk.py:149-171 (error_norm == 0 gives inf,
    // which min(MAX_FACTOR, .) turns into MAX_FACTOR exactly as the reference's special case does).
    // Accept / reject is written with selects, not branches: with per-lane operands every vector-compare ->
    // exec-mask round trip costs as much as several FP64 operations.  The one branch below covers both the rare IEEE re-run
    // and the dense output.
    double pw = 0.9 * pow_minus_fifth<EXACT>(st.error_norm);
    bool accept = st.error_norm < 1.0;
    const bool redo = !EXACT && !__builtin_isfinite(st.error_norm);   // NaN / inf from the lean division (v_cmp_class: no constant)
    if (fail || redo || (accept && it.t_next <= t_new)) {
        if (fail) return 2;
        if (redo) {
            st = rk_attempt_core<kDivIeee>(y, k0, h, negVmax, Km, rtol, atol);
            pw = 0.9 * pow_minus_fifth<EXACT>(st.error_norm);
            accept = st.error_norm < 1.0;
        }
        // ---- outputs with t_eval in (t_old, t] (ivp.py:700-720) by the quartic interpolant ----
        if (accept && it.t_next <= t_new) mm_attempt_outputs<WRITE_PRED, DIV>(it, s_tp, st, k0, t, t_new, y, pred);
    }
    double fac_acc = py_min(10.0, pw);
    fac_acc = it.rejected ? py_min(1.0, fac_acc) : fac_acc;
    const double fac_rej = py_max(0.2, pw);
    it.h_abs *= accept ? fac_acc : fac_rej;
    it.rejected = !accept;
    it.t = accept ? t_new : t;
    it.y = accept ? st.y_new : y;
    it.f = accept ? st.k6 : k0;
    // head of the next _step_impl (rk.py:120-127) after an accepted step; after a rejection t, hence
    // min_step, is unchanged and the reference only re-tests h_abs < min_step (top of this function)
    it.min_step = min_step_of(it.t);
    it.h_abs = (accept && it.h_abs < it.min_step) ? it.min_step : it.h_abs;
    return (accept && (t_new - it.t_bound >= 0)) ? 1 : 0;  // base.py:196
}

// ---------------------------------------------------------------------------------------------------------------------
// The lone chain, by hand.  A stiff solve that runs alone in its wave (solve_sched.h: solo phase, uniform tail) is the serial
// critical path of every early sweep: ~10^5 dependent attempts, each a chain of ~190 compiler-scheduled instructions of which
// a lone wave issues one per >= 4 clocks whatever they are (0.43 us per attempt).  ~100 of them are the Runge-Kutta stages,
// which the compiler does well; the other ~90 are step control written as selects for the per-lane case (two v_cndmask per
// double), scalar bookkeeping and Dormand-Prince constants re-materialised with s_mov pairs because the tableau does not
// fit beside the scheduler's state.  mm_fast_uniform_attempts() is that loop as ONE inline-asm block for the common case
// of a stiff chain - an attempt that is rejected, or accepted without a data time in (t, t_new] - with the tableau in
// scalar registers for the whole loop, branches instead of selects (every operand is wave-uniform) and no bookkeeping:
// ~140 instructions per attempt.  Anything else - an output is due, the step size underflows, the error norm leaves
// 2^-64 .. 2^64 (zero, inf and NaN included: the lean division's re-run, the controller's special cases) - leaves the
// block BEFORE the attempt is committed, and mm_item_attempt() performs that attempt from the same state.
// Bit-identical to mm_item_attempt<., kDivLean6, false> by construction: the same operations in the same order with the
// same fused multiply-adds the compiler forms for rk_attempt_core (first two terms of a stage sum: fma(k0, a_i1, rn(k1 *
// a_i2)), the others fused in order; (sum) * h + y fused) - tests/test_gpu_parity.py runs stiff populations with the block
// on and off (smc_set_fast_tail) and demands equal bits, so a compiler that changes its mind is noticed.
// Hazards the assembler does not see inside an asm block (gfx940 rules of LLVM's hazard recogniser): one independent
// instruction or s_nop between a transcendental (v_rcp_f64, v_log_f32, v_exp_f32) and the first use of its result.
// Returns 1 when the next attempt needs mm_item_attempt() (state untouched by it), 0 when `budget` ran out.
// the first lane's value in a scalar register pair: for operands that are wave-uniform by construction
__device__ __forceinline__ double mm_uniform(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ int mm_fast_uniform_attempts(MMItem &it, double rtol, double atol, int &budget) {
    double t = it.t, y = it.y, k0 = it.f, habs = it.h_abs, minstep = it.min_step;
    // scalar-register operands of the block: wave-uniform by construction, the v_readfirstlane says so to the compiler
    const double nvm = mm_uniform(it.negVmax), km = mm_uniform(it.Km), tb = mm_uniform(it.t_bound);
    const double tstop = mm_uniform(fmin(it.t_next, it.t_bound));   // the first time the block must not step over
    atol = mm_uniform(atol);
    // wave-uniform by construction; the readfirstlane makes them so for the compiler (scalar-register operands of the block)
    int attempts = __builtin_amdgcn_readfirstlane(it.attempts), rejected = __builtin_amdgcn_readfirstlane(it.rejected ? 1 : 0), code;
    budget = __builtin_amdgcn_readfirstlane(budget);
    double tnew, h, k1, k2, k3, k4, k5, k6, acc, den, num, r, e, q;   // k1 doubles as y_new (b2 = e2 = 0), acc as the factor
    float w;
    unsigned long long accm;
    asm volatile(
        "s_mov_b32 %[code], 1\n"
        ".Lfast_loop_%=:\n"
        "v_cmp_lt_f64 vcc, %[habs], %[minstep]\n"             // rk.py:133-134: left to mm_item_attempt
        "s_cbranch_vccnz .Lfast_end_%=\n"
        "v_add_f64 %[tnew], %[t], %[habs]\n"
        "v_min_f64 %[tnew], %[tnew], %[tb]\n"
        "v_add_f64 %[h], %[tnew], -%[t]\n"
        // stage 1
        "v_mul_f64 %[acc], %[k0], %[a21]\n"
        "v_fma_f64 %[acc], %[acc], %[h], %[y]\n"
        "v_add_f64 %[den], %[km], %[acc]\n"
        "v_rcp_f64 %[r], %[den]\n"
        "v_mul_f64 %[num], %[nvm], %[acc]\n"
        "v_fma_f64 %[e], -%[den], %[r], 1.0\n"
        "v_fma_f64 %[r], %[r], %[e], %[r]\n"
        "v_mul_f64 %[q], %[num], %[r]\n"
        "v_fma_f64 %[e], -%[den], %[q], %[num]\n"
        "v_fma_f64 %[k1], %[e], %[r], %[q]\n"
        // stage 2
        "v_mul_f64 %[acc], %[k1], %[a32]\n"
        "v_fma_f64 %[acc], %[k0], %[a31], %[acc]\n"
        "v_fma_f64 %[acc], %[acc], %[h], %[y]\n"
        "v_add_f64 %[den], %[km], %[acc]\n"
        "v_rcp_f64 %[r], %[den]\n"
        "v_mul_f64 %[num], %[nvm], %[acc]\n"
        "v_fma_f64 %[e], -%[den], %[r], 1.0\n"
        "v_fma_f64 %[r], %[r], %[e], %[r]\n"
        "v_mul_f64 %[q], %[num], %[r]\n"
        "v_fma_f64 %[e], -%[den], %[q], %[num]\n"
        "v_fma_f64 %[k2], %[e], %[r], %[q]\n"
        // stage 3
        "v_mul_f64 %[acc], %[k1], %[a42]\n"
        "v_fma_f64 %[acc], %[k0], %[a41], %[acc]\n"
        "v_fma_f64 %[acc], %[k2], %[a43], %[acc]\n"
        "v_fma_f64 %[acc], %[acc], %[h], %[y]\n"
        "v_add_f64 %[den], %[km], %[acc]\n"
        "v_rcp_f64 %[r], %[den]\n"
        "v_mul_f64 %[num], %[nvm], %[acc]\n"
        "v_fma_f64 %[e], -%[den], %[r], 1.0\n"
        "v_fma_f64 %[r], %[r], %[e], %[r]\n"
        "v_mul_f64 %[q], %[num], %[r]\n"
        "v_fma_f64 %[e], -%[den], %[q], %[num]\n"
        "v_fma_f64 %[k3], %[e], %[r], %[q]\n"
        // stage 4
        "v_mul_f64 %[acc], %[k1], %[a52]\n"
        "v_fma_f64 %[acc], %[k0], %[a51], %[acc]\n"
        "v_fma_f64 %[acc], %[k2], %[a53], %[acc]\n"
        "v_fma_f64 %[acc], %[k3], %[a54], %[acc]\n"
        "v_fma_f64 %[acc], %[acc], %[h], %[y]\n"
        "v_add_f64 %[den], %[km], %[acc]\n"
        "v_rcp_f64 %[r], %[den]\n"
        "v_mul_f64 %[num], %[nvm], %[acc]\n"
        "v_fma_f64 %[e], -%[den], %[r], 1.0\n"
        "v_fma_f64 %[r], %[r], %[e], %[r]\n"
        "v_mul_f64 %[q], %[num], %[r]\n"
        "v_fma_f64 %[e], -%[den], %[q], %[num]\n"
        "v_fma_f64 %[k4], %[e], %[r], %[q]\n"
        // stage 5
        "v_mul_f64 %[acc], %[k1], %[a62]\n"
        "v_fma_f64 %[acc], %[k0], %[a61], %[acc]\n"
        "v_fma_f64 %[acc], %[k2], %[a63], %[acc]\n"
        "v_fma_f64 %[acc], %[k3], %[a64], %[acc]\n"
        "v_fma_f64 %[acc], %[k4], %[a65], %[acc]\n"
        "v_fma_f64 %[acc], %[acc], %[h], %[y]\n"
        "v_add_f64 %[den], %[km], %[acc]\n"
        "v_rcp_f64 %[r], %[den]\n"
        "v_mul_f64 %[num], %[nvm], %[acc]\n"
        "v_fma_f64 %[e], -%[den], %[r], 1.0\n"
        "v_fma_f64 %[r], %[r], %[e], %[r]\n"
        "v_mul_f64 %[q], %[num], %[r]\n"
        "v_fma_f64 %[e], -%[den], %[q], %[num]\n"
        "v_fma_f64 %[k5], %[e], %[r], %[q]\n"
        // y_new and k6 = f(y_new)
        "v_mul_f64 %[acc], %[k2], %[b3]\n"
        "v_fma_f64 %[acc], %[k0], %[b1], %[acc]\n"
        "v_fma_f64 %[acc], %[k3], %[b4], %[acc]\n"
        "v_fma_f64 %[acc], %[k4], %[b5], %[acc]\n"
        "v_fma_f64 %[acc], %[k5], %[b6], %[acc]\n"
        "v_fma_f64 %[k1], %[h], %[acc], %[y]\n"
        "v_add_f64 %[den], %[km], %[k1]\n"
        "v_rcp_f64 %[r], %[den]\n"
        "v_mul_f64 %[num], %[nvm], %[k1]\n"
        "v_fma_f64 %[e], -%[den], %[r], 1.0\n"
        "v_fma_f64 %[r], %[r], %[e], %[r]\n"
        "v_mul_f64 %[q], %[num], %[r]\n"
        "v_fma_f64 %[e], -%[den], %[q], %[num]\n"
        "v_fma_f64 %[k6], %[e], %[r], %[q]\n"
        // error estimate over the scale: den = atol + max(|y|, |y_new|) * rtol, num = (sum e_j k_j) * h, q = num / den
        "v_max_f64 %[den], |%[y]|, |%[k1]|\n"
        "v_fma_f64 %[den], %[rtol], %[den], %[atol]\n"
        "v_rcp_f64 %[r], %[den]\n"
        "v_mul_f64 %[acc], %[k2], %[e3]\n"
        "v_fma_f64 %[acc], %[k0], %[e1], %[acc]\n"
        "v_fma_f64 %[acc], %[k3], %[e4], %[acc]\n"
        "v_fma_f64 %[acc], %[k4], %[e5], %[acc]\n"
        "v_fma_f64 %[acc], %[k5], %[e6], %[acc]\n"
        "v_fma_f64 %[acc], %[k6], %[e7], %[acc]\n"
        "v_mul_f64 %[num], %[h], %[acc]\n"
        "v_fma_f64 %[e], -%[den], %[r], 1.0\n"
        "v_fma_f64 %[r], %[r], %[e], %[r]\n"
        "v_mul_f64 %[q], %[r], %[num]\n"
        "v_fma_f64 %[e], -%[den], %[q], %[num]\n"
        "v_fma_f64 %[q], %[e], %[r], %[q]\n"                  // error_norm = |q|
        // 0.9 * error_norm ** -0.2: pow_minus_fifth_core<false> (rk45_math.h)
        "v_cvt_f32_f64 %[w], |%[q]|\n"
        "v_cmp_lt_f64 %[accm], |%[q]|, 1.0\n"                 // accept (rk.py:149)
        "v_log_f32 %[w], %[w]\n"
        "s_nop 0\n"
        "v_mul_f32 %[w], 0xbe4ccccd, %[w]\n"
        "v_exp_f32 %[w], %[w]\n"
        "s_nop 0\n"
        "v_cvt_f64_f32 %[r], %[w]\n"                          // y
        "v_mul_f64 %[e], %[r], %[r]\n"
        "v_mul_f64 %[e], %[e], %[e]\n"
        "v_mul_f64 %[e], %[e], %[r]\n"                        // y^5
        "v_fma_f64 %[e], -|%[q]|, %[e], 1.0\n"                // rho
        "v_mul_f64 %[acc], %[e], %[r]\n"                      // y * rho
        "v_fma_f64 %[e], %[c012], %[e], %[c02]\n"             // 0.2 + 0.12 rho
        "v_fma_f64 %[r], %[acc], %[e], %[r]\n"
        "v_mul_f64 %[acc], %[r], %[c09]\n"
        "s_and_b64 vcc, exec, %[accm]\n"
        "s_cbranch_vccz .Lfast_reject_%=\n"
        // ---- accepted: no output may be due (tstop = min(t_next, t_bound)) and error_norm >= 2^-64
        "v_cmp_le_f64 vcc, %[tstop], %[tnew]\n"
        "s_cbranch_vccnz .Lfast_end_%=\n"
        "v_cmp_nge_f64 vcc, |%[q]|, %[clo]\n"
        "s_cbranch_vccnz .Lfast_end_%=\n"
        "v_min_f64 %[acc], %[acc], %[c10]\n"                    // min(MAX_FACTOR, .)
        "s_cmp_eq_u32 %[rejected], 0\n"
        "s_cbranch_scc1 .Lfast_first_%=\n"
        "v_min_f64 %[acc], %[acc], 1.0\n"                       // rk.py:158-159
        ".Lfast_first_%=:\n"
        "v_mul_f64 %[habs], |%[h]|, %[acc]\n"
        "v_mov_b64 %[t], %[tnew]\n"
        "v_mov_b64 %[y], %[k1]\n"
        "v_mov_b64 %[k0], %[k6]\n"
        "s_mov_b32 %[rejected], 0\n"
        "v_lshl_add_u64 %[e], %[tnew], 0, 1\n"                // min_step = 10 |nextafter(t, inf) - t|, t >= 0 (rk.py:120)
        "v_add_f64 %[e], %[e], -%[tnew]\n"
        "v_mul_f64 %[minstep], |%[e]|, %[c10]\n"
        "v_max_f64 %[habs], %[habs], %[minstep]\n"            // rk.py:122-127
        "s_branch .Lfast_next_%=\n"
        ".Lfast_reject_%=:\n"
        "v_cmp_nlt_f64 vcc, |%[q]|, %[chi]\n"                 // error_norm >= 2^64, inf or NaN: left to mm_item_attempt
        "s_cbranch_vccnz .Lfast_end_%=\n"
        "v_max_f64 %[acc], %[acc], %[c02]\n"                    // max(MIN_FACTOR, .)
        "v_mul_f64 %[habs], |%[h]|, %[acc]\n"
        "s_mov_b32 %[rejected], 1\n"
        ".Lfast_next_%=:\n"
        "s_add_i32 %[attempts], %[attempts], 1\n"
        "s_sub_i32 %[budget], %[budget], 1\n"
        "s_cmp_gt_i32 %[budget], 0\n"
        "s_cbranch_scc1 .Lfast_loop_%=\n"
        "s_mov_b32 %[code], 0\n"
        ".Lfast_end_%=:\n"
        : [t] "+v"(t), [y] "+v"(y), [k0] "+v"(k0), [habs] "+v"(habs), [minstep] "+v"(minstep), [attempts] "+s"(attempts),
          [rejected] "+s"(rejected), [budget] "+s"(budget), [code] "=&s"(code), [tnew] "=&v"(tnew), [h] "=&v"(h), [k1] "=&v"(k1),
          [k2] "=&v"(k2), [k3] "=&v"(k3), [k4] "=&v"(k4), [k5] "=&v"(k5), [k6] "=&v"(k6), [acc] "=&v"(acc),
          [den] "=&v"(den), [num] "=&v"(num), [r] "=&v"(r), [e] "=&v"(e), [q] "=&v"(q), [w] "=&v"(w),
          [accm] "=&s"(accm)
        : [nvm] "s"(nvm), [km] "s"(km), [tb] "s"(tb), [tstop] "s"(tstop), [rtol] "v"(rtol), [atol] "s"(atol), [c02] "v"(0.2),
          [a21] "s"(A21), [a31] "s"(A31), [a32] "s"(A32), [a41] "s"(A41), [a42] "s"(A42), [a43] "s"(A43), [a51] "s"(A51),
          [a52] "s"(A52), [a53] "s"(A53), [a54] "s"(A54), [a61] "s"(A61), [a62] "s"(A62), [a63] "s"(A63), [a64] "s"(A64),
          [a65] "s"(A65), [b1] "s"(B1), [b3] "s"(B3), [b4] "s"(B4), [b5] "s"(B5), [b6] "s"(B6), [e1] "s"(E1), [e3] "s"(E3),
          [e4] "s"(E4), [e5] "s"(E5), [e6] "s"(E6), [e7] "s"(E7), [c09] "s"(0.9), [c10] "s"(10.0), [c012] "s"(0.12),
          [clo] "s"(0x1p-64), [chi] "s"(0x1p64)
        : "vcc", "scc");
    it.t = t;
    it.y = y;
    it.f = k0;
    it.h_abs = habs;
    it.min_step = minstep;
    it.attempts = attempts;
    it.rejected = rejected != 0;
    return code;
}

// Tried for the lone chain of the uniform tail and NOT adopted (round 3, tools/isa_blocks.py on the listings): (a) a twin
// of this function with rk.py's own branches instead of the selects - bit-identical, 14 selects fewer, but inside the
// scheduler the compiler then re-materialises 25 instead of 5 FP64 constants per attempt (51 s_mov_b32: 223 instructions per
// attempt instead of 190); (b) the same as a real function call (own register allocation): the call's register needs drive
// the whole kernel to 149 VGPRs = 3 waves per SIMD (and hipcc 7.2 fails on an LDS pointer passed through the call); (c) the
// Dormand-Prince tableau pinned in vector registers by opaque v_mov (the lone wave has registers to spare): 177 VGPRs
// kernel-wide, or 55 spills under a 128-register cap, and the IEEE re-run gets if-converted into the loop body.

}  // namespace smc
