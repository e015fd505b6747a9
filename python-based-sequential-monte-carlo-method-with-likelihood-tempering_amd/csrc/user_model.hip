// user_model.hip -- SURVEY.md 8(f) N1: the reference is meant to be "modified for your problem" (README.md:4): a user
// writes Micmem_likelihood.py's three ingredients - the ODE right-hand side (:14-15), the initial state and what is
// compared with the data (:17-33) - and keeps the drivers.  Here the same three ingredients are given as HIP device
// functions; they are compiled at run time (hiprtc, gfx950) into the likelihood kernel below and plugged into the
// on-device loop (propose -> solve -> accept) exactly like the two built-in models.
//
// The kernel restates solve_ivp(method="RK45", t_eval=t, rtol, atol) for an NS-dimensional state as SciPy does it
// (scipy/integrate/_ivp: rk.py rk_step / _step_impl, common.py select_initial_step and RMS norm, ivp.py t_eval
// dispatch through the quartic dense output) and the Gaussian log-likelihood of Micmem_likelihood.py:62-73.
// For NS = 1 every operation is in the order of the built-in Michaelis-Menten kernel, so the MM model written as a user
// model reproduces it (tests/test_gpu_user_model.py); for NS > 1 the stage sums are plain left-to-right sums where
// NumPy calls BLAS, so agreement with SciPy is at rounding level per step, not bitwise.
// Scheduling as in the built-in kernel: (experiment, particle) items, persistent lanes that take the next item from a
// global counter when their solve ends; a small built-in kernel then sums the items of a particle into its logL.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/smc_hip.h"
#include "smc_internal.h"
#include "solve_sched.h"   // kChunk: the grid is sized in chunks of the shared scheduler

namespace smc {

// solve_sched.h, sweep_args.h and philox.h as strings (csrc/Makefile generates the file from the headers themselves): hiprtc
// gets them as in-memory headers, so the run-time compiled kernel is scheduled by the very code the built-in kernel uses
#include "embedded_headers.inc"

// what the user's functions may call (include/smc_hip.h)
// The user's functions are compiled TWICE, in two namespaces that differ in what smc_div(a, b) is (include/smc_hip.h): the
// six-operation division of the built-in kernel, and a / b.  An attempt runs on the first and is repeated on the second
// whenever its error norm is not finite (smc_user::item_attempt) - the built-in kernel's scheme, for a model it does not know.
static const char *kUserPreludeLean = R"SRC(
#include "rk45_math.h"
namespace smc_user_lean {
__device__ __forceinline__ double smc_div(double a, double b) { return smc::lean_div6(a, b); }
)SRC";
static const char *kUserPreludeIeee = R"SRC(
}  // namespace smc_user_lean
namespace smc_user_ieee {
__device__ __forceinline__ double smc_div(double a, double b) { return a / b; }
)SRC";

static const char *kUserKernelSource = R"SRC(
// ---- appended by libsmc_hip.so after the user's source -------------------------------------------------------
#include "sweep_args.h"     // in-memory headers handed to hiprtc by the library: the argument blocks,
#include "philox.h"         // the counter-based generator (the early-rejection bound re-derives the acceptance uniform),
#include "solve_sched.h"    // the scheduler the built-in Michaelis-Menten kernel uses,
#include "rk45_math.h"      // and its step-controller arithmetic (fast inverse fifth root, min_step)
#define NS SMC_USER_NS
namespace smc_user {
__device__ const double RK_A[6][5] = {
    {0, 0, 0, 0, 0},
    {1.0 / 5, 0, 0, 0, 0},
    {3.0 / 40, 9.0 / 40, 0, 0, 0},
    {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0},
    {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0},
    {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
__device__ const double RK_C[6] = {0.0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1.0};
__device__ const double RK_B[6] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
__device__ const double RK_E[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920, 17253.0 / 339200, -22.0 / 525, 1.0 / 40};
__device__ const double RK_P[7][4] = {
    {1, -8048581381.0 / 2820520608, 8663915743.0 / 2820520608, -12715105075.0 / 11282082432},
    {0, 0, 0, 0},
    {0, 131558114200.0 / 32700410799, -68118460800.0 / 10900136933, 87487479700.0 / 32700410799},
    {0, -1754552775.0 / 470086768, 14199869525.0 / 1410260304, -10690763975.0 / 1880347072},
    {0, 127303824393.0 / 49829197408, -318862633887.0 / 49829197408, 701980252875.0 / 199316789632},
    {0, -282668133.0 / 205662961, 2019193451.0 / 616988883, -1453857185.0 / 822651844},
    {0, 40617522.0 / 29380423, -110615467.0 / 29380423, 69997945.0 / 29380423}};

__device__ __forceinline__ double py_min(double a, double b) { return (b < a) ? b : a; }
__device__ __forceinline__ double py_max(double a, double b) { return (b > a) ? b : a; }
// the two compilations of the user's functions (see the prelude): with the six-operation division, and with a / b
struct Lean {
    static __device__ __forceinline__ void rhs(double t, const double *y, const double *th, const double *c, double *d) { smc_user_lean::smc_user_rhs(t, y, th, c, d); }
    static __device__ __forceinline__ double div(double a, double b) { return SMC_USER_USES_DIV ? smc::lean_div6(a, b) : a / b; }
};
struct Ieee {
    static __device__ __forceinline__ void y0(const double *th, const double *c, double *y) { smc_user_ieee::smc_user_y0(th, c, y); }
    static __device__ __forceinline__ void rhs(double t, const double *y, const double *th, const double *c, double *d) { smc_user_ieee::smc_user_rhs(t, y, th, c, d); }
    static __device__ __forceinline__ double obs(double t, const double *y, const double *th, const double *c) { return smc_user_ieee::smc_user_obs(t, y, th, c); }
    static __device__ __forceinline__ double div(double a, double b) { return a / b; }
};
// common.py:63-65  np.linalg.norm(x) / x.size ** 0.5
__device__ __forceinline__ double rms(const double *x) {
    if (NS == 1) return fabs(x[0]);   // sqrt(x*x) / sqrt(1) is |x| exactly in IEEE arithmetic (as the built-in kernel writes it)
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NS; ++i) s += x[i] * x[i];
    return sqrt(s) / sqrt((double)NS);
}

// One solve_ivp(RK45, t_eval = t[0..n_t)) call as a resumable state: item_begin sets it up (initial state, first
// derivative, select_initial_step), every item_attempt is one pass of rk.py's `while not step_accepted` body plus,
// when the step is accepted, the t_eval outputs it covers.  sr2 accumulates (obs - smc_user_obs)^2.
struct Item {
    double t, h_abs, y[NS], f[NS], sr2;
    double t_bound, t_next;   // t_eval[n_t - 1]; t_eval[i_out] (+inf when every data time has been served) - copies of LDS
                              // values, so that an attempt without outputs (nearly all of a stiff solve) reads no memory
    int i_out, status;   // status: 0 running, 1 finished, -1 step size underflow (the reference's solver raises)
    bool rejected;
};
// The data of one experiment in LDS: n_t + 1 (time, observation) pairs, the last one the sentinel (+inf, 0) - as in the
// built-in kernel (mm_rk45.h: mm_table_fill), so that an output costs ONE ds_read_b128 and no test for the end of the row.
__device__ __forceinline__ void item_cache_times(Item &it, const double2 *tp, int n_t) {
    it.t_bound = tp[n_t - 1].x;
    it.t_next = tp[it.i_out].x;
}

__device__ __forceinline__ void emit(Item &it, const double *yy, const double *theta, const double *cond, double t_out, double obs) {
    const double r = obs - Ieee::obs(t_out, yy, theta, cond);
    it.sr2 += r * r;
}

__device__ void item_begin(Item &it, const double *theta, const double *cond, const double2 *tp, int n_t,
                           double rtol, double atol) {
    const double t0 = tp[0].x, t_bound = tp[n_t - 1].x;
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    double tmp[NS];
    it.t = t0;
    it.sr2 = 0.0;
    it.i_out = 0;
    it.status = 0;
    it.rejected = false;
    Ieee::y0(theta, cond, it.y);
    Ieee::rhs(t0, it.y, theta, cond, it.f);
    // common.py select_initial_step, direction +1, order 4, max_step inf
    const double interval_length = fabs(t_bound - t0);
    if (interval_length == 0.0) {
        it.h_abs = 0.0;
    } else {
        double scale[NS], y1[NS], f1[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) { scale[i] = atol + fabs(it.y[i]) * rtol; tmp[i] = it.y[i] / scale[i]; }
        const double d0 = rms(tmp);
#pragma unroll
        for (int i = 0; i < NS; ++i) tmp[i] = it.f[i] / scale[i];
        const double d1 = rms(tmp);
        double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
        h0 = py_min(h0, interval_length);
#pragma unroll
        for (int i = 0; i < NS; ++i) y1[i] = it.y[i] + h0 * 1.0 * it.f[i];
        Ieee::rhs(t0 + h0 * 1.0, y1, theta, cond, f1);
#pragma unroll
        for (int i = 0; i < NS; ++i) tmp[i] = (f1[i] - it.f[i]) / scale[i];
        const double d2 = rms(tmp) / h0;
        // x ** (1 / 5) as the built-in kernel evaluates it: the reciprocal of the fast inverse fifth root (rk45_math.h)
        const double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? py_max(1e-6, h0 * 1e-3) : 1.0 / smc::pow_minus_fifth<false>(0.01 / py_max(d1, d2));
        it.h_abs = py_min(py_min(py_min(100 * h0, h1), interval_length), inf);
    }
    if (it.t == t_bound) {   // base.py:181-187: nothing to integrate
        while (tp[it.i_out].x <= it.t) { emit(it, it.y, theta, cond, tp[it.i_out].x, tp[it.i_out].y); ++it.i_out; }   // stops at the sentinel
        it.status = 1;
    }
    item_cache_times(it, tp, n_t);
}

// rk_step (rk.py:64-71) and the error norm (rk.py:106-110,146-147) of one attempt: a pure function of the item's state.
// M = the model with smc_div as the six-operation division (Lean) or as the IEEE sequence (Ieee): see item_attempt.
struct Stages {
    double K[7][NS], y_new[NS], error_norm;
};
template <class M>
__device__ __forceinline__ void rk_stages(Stages &st, const Item &it, double t, double h, const double *theta, const double *cond,
                                          double rtol, double atol) {
    double tmp[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) st.K[0][i] = it.f[i];
#pragma unroll
    for (int s = 1; s < 6; ++s) {   // rk_step: dy = K[:s].T @ a[:s] * h
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < s; ++j) acc += st.K[j][i] * RK_A[s][j];
            tmp[i] = it.y[i] + acc * h;
        }
        M::rhs(t + RK_C[s] * h, tmp, theta, cond, st.K[s]);
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) acc += st.K[j][i] * RK_B[j];      // b2 = 0 included: 0 * NaN is NaN in NumPy's dot as well
        st.y_new[i] = it.y[i] + h * acc;
    }
    M::rhs(t + h, st.y_new, theta, cond, st.K[6]);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        double scale = atol + fmax(fabs(it.y[i]), fabs(st.y_new[i])) * rtol;
        if (it.y[i] != it.y[i] || st.y_new[i] != st.y_new[i]) scale = it.y[i] + st.y_new[i];   // np.maximum propagates NaN
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < 7; ++j) acc += st.K[j][i] * RK_E[j];
        tmp[i] = M::div(acc * h, scale);
    }
    st.error_norm = rms(tmp);
}

// One pass of rk.py's `while not step_accepted` body.  Written like the built-in kernel's mm_item_attempt (mm_rk45.h): the
// accept / reject bookkeeping as selects, ONE rarely taken branch for everything else (failure, the IEEE re-run, the dense
// output) - with per-lane operands every compare -> exec-mask round trip costs several FP64 operations, and on the
// wave-uniform operands of a lone chain every compare -> scalar-branch round trip drains the pipeline.
// smc_div: the attempt runs on the model compiled with the six-operation division (no scaling, no fix-up: NaN where a / b
// needs a subnormal or infinite divisor or a * (1 / b) overflows); whenever the error norm does not come out finite the
// whole attempt is repeated on the model compiled with IEEE division, exactly as the built-in kernel does.
__device__ __forceinline__ void item_attempt(Item &it, const double *theta, const double *cond, const double2 *tp, double rtol,
                                             double atol) {
    const double t_bound = it.t_bound;
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    const double t = it.t;
    const double min_step = (t >= 0.0) ? smc::min_step_of(t) : 10 * fabs(nextafter(t, inf) - t);
    // rk.py:111-121: clip at the start of a step (a value raised here stays raised for the re-tries of the step)
    double h_abs = (!it.rejected && it.h_abs < min_step) ? min_step : it.h_abs;
    const bool fail = h_abs < min_step;                  // rk.py:133-134 TOO_SMALL_STEP: handled in the rare branch below
    const double t_new = fmin(t + h_abs, t_bound);       // rk.py:137-141 (h_abs is never NaN: Python's min / max drop a NaN factor)
    const double h = t_new - t;
    h_abs = fabs(h);
    Stages st;
    rk_stages<Lean>(st, it, t, h, theta, cond, rtol, atol);
    // error_norm ** -0.2 (rk.py:155,169) by the dedicated inverse fifth root of the built-in kernel (<= 1.5 ulp; the generic
    // pow costs 350 ns on the dependent chain of an attempt): 0 -> inf, which min(10, .) turns into MAX_FACTOR
    double pw = 0.9 * smc::pow_minus_fifth<false>(st.error_norm);
    bool accept = st.error_norm < 1.0;                   // false for a NaN norm: Python's max(0.2, nan) is 0.2
    const bool redo = SMC_USER_USES_DIV && !__builtin_isfinite(st.error_norm);
    if (fail || redo || (accept && it.t_next <= t_new)) {
        if (fail) {
            it.status = -1;
            return;
        }
        if (redo) {
            rk_stages<Ieee>(st, it, t, h, theta, cond, rtol, atol);
            pw = 0.9 * smc::pow_minus_fifth<false>(st.error_norm);
            accept = st.error_norm < 1.0;
        }
        // outputs in (t, t_new] and t_eval[0] == t0 on the first step (ivp.py:700-720): quartic dense output
        if (accept && it.t_next <= t_new) {
            double Q[NS][4];
#pragma unroll
            for (int i = 0; i < NS; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j < 7; ++j)
                        if (RK_P[j][k] != 0.0) acc += st.K[j][i] * RK_P[j][k];   // an accepted step has finite K: the zeros change nothing
                    Q[i][k] = acc;
                }
            const double hd = t_new - t;
            int i_out = it.i_out;
            double2 nx = tp[i_out];                      // (t_next, its observation)
            do {
                const double x = smc::checked_lean_div(nx.x - t, hd);
                const double p1 = x, p2 = p1 * x, p3 = p2 * x, p4 = p3 * x;
                double yy[NS];
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    double acc = 0.0;
                    acc += Q[i][0] * p1;
                    acc += Q[i][1] * p2;
                    acc += Q[i][2] * p3;
                    acc += Q[i][3] * p4;
                    yy[i] = hd * acc + it.y[i];
                }
                emit(it, yy, theta, cond, nx.x, nx.y);
                ++i_out;
                nx = tp[i_out];                          // i_out == n_t reads the sentinel (+inf, 0)
            } while (nx.x <= t_new);
            it.i_out = i_out;
            it.t_next = nx.x;
        }
    }
    double fac_acc = py_min(10.0, pw);
    fac_acc = it.rejected ? py_min(1.0, fac_acc) : fac_acc;
    const double fac_rej = py_max(0.2, pw);
    it.h_abs = h_abs * (accept ? fac_acc : fac_rej);
    it.rejected = !accept;
    it.t = accept ? t_new : t;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        it.y[i] = accept ? st.y_new[i] : it.y[i];
        it.f[i] = accept ? st.K[6][i] : it.f[i];
    }
    it.status = (accept && (t_new - t_bound >= 0)) ? 1 : 0;   // base.py:196
}
}  // namespace smc_user

// What solve_sched.h needs to know about an item of the user model (see the list at the top of that file): the built-in
// Michaelis-Menten kernel's scheduler - chunked dequeue, pool of started items in LDS, tight attempt loop, uniform tail,
// exact early rejection - runs the user's model unchanged (VERDICT r2 item 6).
struct UserOps {
    struct Item {
        smc_user::Item s;
        double th[SMC_USER_DIM];
        long long out_idx;      // e * n + p
        int e;
        unsigned attempts;
    };
    static constexpr int kPoolWords = 2 * NS + 6;
    const smc::UserSolveArgs &a;
    long long n;
    int n_ex;
    const int *list;            // the predictably long items of the sweep, handed out first - only for a model that comes with
    unsigned n_list;            // a cost hint (smc_user_cost; smc_user_cost_scan_kernel below builds the lists), else nullptr / 0
    const int *solo;            // ... and its longest solves, one per wave on uniform operands
    unsigned n_solo;
    int patience;               // in-phase waves (solve_sched.h): with the cost order of a model that comes with a cost hint
    long long n_pos;            // positions of the index-ordered pass

    __device__ __forceinline__ const double *cond(int e) const { return a.cond + (long long)e * a.n_cond; }
    const double2 *s_tp;        // the data of all experiments in LDS: rows of n_t + 1 (time, observation) pairs (item_cache_times)
    __device__ __forceinline__ const double2 *row(int e) const { return s_tp + e * (a.n_t + 1); }
    __device__ __forceinline__ void publish(long long idx, double sum, int info) const {
        // visible to the waves of other XCDs while the kernel runs (the early-rejection bound reads the siblings' sums)
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(a.sum_r2) + idx, (unsigned long long)__double_as_longlong(sum),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.info[idx] = info;
    }
    __device__ __forceinline__ void load_theta(Item &it, long long p) const {
#pragma unroll
        for (int c = 0; c < SMC_USER_DIM; ++c) it.th[c] = a.theta[c * a.stride + p];
    }
    __device__ __forceinline__ int start(long long p, int e, bool from_list, Item &nb) const {
        nb.out_idx = (long long)e * a.n + p;
        nb.e = e;
        nb.attempts = 0;
        // index-ordered pass: a particle of the lists has been handed out already.  The flag was written by the scan kernel
        // that built the lists - the hint is NOT evaluated a second time here (a user's expression need not round the same
        // way in two kernels, and an item skipped here but missing from the list would never be solved)
        if (!from_list && list && a.listed[p] != 0) return smc::kStartSkipped;
        if (a.p0 && a.p0[p] == 0) {          // masked proposal: not solved, the accept kernel keeps lk1
            publish(nb.out_idx, 0.0, 0);
            return smc::kStartDone;
        }
        load_theta(nb, p);
        smc_user::item_begin(nb.s, nb.th, cond(e), row(e), a.n_t, a.rtol, a.atol);
        if (nb.s.status == 0) return smc::kStartStarted;
        publish(nb.out_idx, nb.s.sr2, nb.s.status < 0 ? (1 << 30) : 0);
        return smc::kStartDone;
    }
    __device__ __forceinline__ void pack(const Item &nb, double *slot) const {
        slot[0 * 64] = nb.s.t;
        slot[1 * 64] = nb.s.h_abs;
        slot[2 * 64] = nb.s.sr2;
        slot[3 * 64] = __hiloint2double(nb.s.i_out, (int)nb.s.rejected);
        slot[4 * 64] = __hiloint2double((int)nb.attempts, nb.e);
        slot[5 * 64] = __longlong_as_double(nb.out_idx);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            slot[(6 + i) * 64] = nb.s.y[i];
            slot[(6 + NS + i) * 64] = nb.s.f[i];
        }
    }
    __device__ __forceinline__ void unpack(Item &it, const double *slot) const {
        it.s.t = slot[0 * 64];
        it.s.h_abs = slot[1 * 64];
        it.s.sr2 = slot[2 * 64];
        const double w3 = slot[3 * 64], w4 = slot[4 * 64];
        it.s.rejected = __double2loint(w3) != 0;
        it.s.i_out = __double2hiint(w3);
        it.e = __double2loint(w4);
        it.attempts = (unsigned)__double2hiint(w4);
        it.out_idx = __double_as_longlong(slot[5 * 64]);
        it.s.status = 0;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            it.s.y[i] = slot[(6 + i) * 64];
            it.s.f[i] = slot[(6 + NS + i) * 64];
        }
        smc_user::item_cache_times(it.s, row(it.e), a.n_t);
        load_theta(it, it.out_idx - (long long)it.e * a.n);      // the parameters come from HBM / L2 again, not through the pool
    }
    __device__ __forceinline__ int attempt(Item &it) const {
        smc_user::item_attempt(it.s, it.th, cond(it.e), row(it.e), a.rtol, a.atol);
        ++it.attempts;
        if (it.attempts >= 0x1fffffffu) it.s.status = -1;        // hard bound so that every wave drains
        return it.s.status;
    }
    __device__ __forceinline__ int uniform_attempts(Item &it, int budget) const { return smc::uniform_attempts_plain(*this, it, budget); }
    __device__ __forceinline__ bool long_running(const Item &it) const { return it.attempts > 64u; }
    __device__ __forceinline__ long long positions() const { return n_pos; }
    __device__ __forceinline__ int start_at(long long pos, int e, Item &nb) const {
        return start(a.order ? (long long)a.order[pos] : pos, e, false, nb);
    }
    __device__ __forceinline__ void finish(Item &it, int st) const {
        publish(it.out_idx, it.s.sr2, (int)(it.attempts & 0x1fffffffu) | ((st < 0) ? (1 << 30) : 0));
    }
    __device__ __forceinline__ Item broadcast(const Item &it, int src) const {
        Item u;
        u.s.t = smc::lane_value(it.s.t, src);
        u.s.h_abs = smc::lane_value(it.s.h_abs, src);
        u.s.sr2 = smc::lane_value(it.s.sr2, src);
        u.s.t_bound = smc::lane_value(it.s.t_bound, src);
        u.s.t_next = smc::lane_value(it.s.t_next, src);
        u.s.i_out = __builtin_amdgcn_readlane(it.s.i_out, src);
        u.s.status = __builtin_amdgcn_readlane(it.s.status, src);
        u.s.rejected = __builtin_amdgcn_readlane((int)it.s.rejected, src) != 0;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            u.s.y[i] = smc::lane_value(it.s.y[i], src);
            u.s.f[i] = smc::lane_value(it.s.f[i], src);
        }
#pragma unroll
        for (int c = 0; c < SMC_USER_DIM; ++c) u.th[c] = smc::lane_value(it.th[c], src);
        u.out_idx = smc::lane_value_ll(it.out_idx, src);
        u.e = __builtin_amdgcn_readlane(it.e, src);
        u.attempts = (unsigned)__builtin_amdgcn_readlane((int)it.attempts, src);
        return u;
    }
    __device__ __forceinline__ bool reject_enabled() const { return a.rej != nullptr; }
    // EXACT early rejection, as for the built-in model (mm_kernels.hip: mm_certainly_rejected): the Gaussian likelihood
    // lk2 = sum_e [c0 - sum_r2_e / (2 sigma^2)] (Micmem_likelihood.py:70-73) only falls while a solve accumulates squared
    // residuals, lk1 and rr are fixed before the sweep, so a proposal that fails exp((lk2 - lk1) gamma) >= rr with the sums
    // accumulated SO FAR (0 for a sibling still running) is rejected whatever the rest would add.  The expression is the
    // one user_finish_kernel evaluates, in the same order.
    __device__ __forceinline__ bool certainly_rejected(const Item &it) const {
        const long long p = it.out_idx - (long long)it.e * a.n;
        const double sigma = a.est_sigma ? it.th[SMC_USER_DIM - 1] : a.sigma_fixed;
        if (!(sigma > 0.0)) return false;
        const double s2 = sigma * sigma;
        const double c0 = (-0.5 * a.n_t) * log(2.0 * 3.141592653589793 * s2);
        double lk2_bound = 0.0;
        for (int k = 0; k < a.n_ex; ++k) {
            double S = 0.0;
            if (k == it.e) {
                S = it.s.sr2;
            } else {
                const double v = __longlong_as_double((long long)__hip_atomic_load(
                    reinterpret_cast<unsigned long long *>(a.sum_r2) + (long long)k * a.n + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (v < 0.0) return true;      // a sibling has already established the rejection
                if (v == v) S = v;             // finished; NaN = still running: counts as 0
            }
            lk2_bound += c0 - S / (2.0 * s2);
        }
        const smc::RejectArgs &r = *a.rej;
        double rr;
        if (r.device_rng) {
            const smc::u32x4 ru = smc::philox_block(r.seed, (unsigned long long)(r.global_offset + p), r.stream, SMC_PHILOX_BLOCK_UNIFORM);
            rr = smc::u01_from(ru.x, ru.y);
        } else {
            rr = r.rr[p];
        }
        double pp = exp((lk2_bound - r.lk1[p]) * r.gamma);
        if (r.prior_mode != 0) pp = pp * r.pratio[p];      // SMC_PRIOR_MODE_MASK == 0
        return pp < rr * (1.0 - 1e-12);
    }
    __device__ __forceinline__ void cancel(Item &it) const {
        publish(it.out_idx, -1.0, (int)(it.attempts & 0x1fffffffu) | (1 << 29));
    }
};

// Outputs per item: the sum of squared residuals and attempts | cancelled << 29 | failed << 30.
// small models: hold the register allocation at four waves per SIMD (the built-in kernel's occupancy); the re-run with IEEE
// division would otherwise cost the bulk loop its fourth wave
#if NS <= 2
#define SMC_USER_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#define SMC_USER_WAVES_ATTR
#endif
extern "C" __global__ void __launch_bounds__(256) SMC_USER_WAVES_ATTR smc_user_solve_kernel(smc::UserSolveArgs a) {
    extern __shared__ double s_pool_all[];       // per wave: a ring of 64 started items of UserOps::kPoolWords words, then
    double *s_pool = s_pool_all + (threadIdx.x >> 6) * (UserOps::kPoolWords * 64);
    double2 *s_tp = reinterpret_cast<double2 *>(s_pool_all + 4 * (UserOps::kPoolWords * 64));   // the data, read by every output
    for (int i = threadIdx.x; i < a.n_ex * (a.n_t + 1); i += blockDim.x) {
        const int e = i / (a.n_t + 1), k = i - e * (a.n_t + 1);
        s_tp[i] = (k < a.n_t) ? make_double2(a.t[e * a.n_t + k], a.obs[e * a.n_t + k])
                              : make_double2(__longlong_as_double(0x7ff0000000000000LL), 0.0);
    }
    __syncthreads();
    const unsigned n_list = a.stiff_list ? (unsigned)__builtin_amdgcn_readfirstlane((int)a.stiff_count[0]) : 0u;
    unsigned n_solo = a.stiff_list ? (unsigned)__builtin_amdgcn_readfirstlane((int)a.stiff_count[1]) : 0u;
    if (n_solo > a.solo_cap) n_solo = a.solo_cap;   // the overflow went onto the ordinary list
    UserOps ops{a, a.n, a.n_ex, a.stiff_list, n_list, a.stiff_list ? a.stiff_list + (a.stiff_cap - 1) : nullptr, n_solo, a.patience,
                a.n_ordered ? (long long)__builtin_amdgcn_readfirstlane((int)a.n_ordered[0]) : a.n, s_tp};
    smc::solve_persistent(ops, a.queue, s_pool);
}

#ifdef SMC_USER_HAS_COST
// The cost hint of the model (include/smc_hip.h): smc_user_cost(theta) ~ RK45 step attempts of one solve.  Above
// SMC_USER_LIST_COST attempts a particle's solves are handed out before the index-ordered items, above SMC_USER_SOLO_COST they
// run one per wave (solve_sched.h) - the thresholds of the built-in Michaelis-Menten kernel (Vmax > 60 Km, > 1000 Km) in
// attempts (3.7 Vmax / Km).  One atomic per listed lane, no cross-lane read after it; every particle at most once.
extern "C" __global__ void __launch_bounds__(256) smc_user_cost_scan_kernel(smc::UserScanArgs a) {
    if (blockIdx.x == 0 && threadIdx.x < 2) a.count_next[threadIdx.x] = 0u;
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.n) return;
    double th[SMC_USER_DIM];
#pragma unroll
    for (int c = 0; c < SMC_USER_DIM; ++c) th[c] = a.theta[c * a.stride + p];
    const bool masked = a.p0 && a.p0[p] == 0;
    const double cost = masked ? 0.0 : smc_user_ieee::smc_user_cost(th);
    const bool on_list = cost > SMC_USER_LIST_COST;     // false for NaN
    a.listed[p] = on_list ? 1 : 0;
    if (a.bucket) {      // cost class: four per factor of two in the hint, the longest first (NaN and < 1: the last real class)
        unsigned b = 127u;
        if (masked) {
            for (int e = 0; e < a.n_ex; ++e) {
                a.done_sums[(long long)e * a.n + p] = 0.0;
                a.done_info[(long long)e * a.n + p] = 0;
            }
        } else {
            const int u = (cost >= 1.0) ? (int)(__float_as_uint((float)cost) >> 21) - 127 * 4 : 0;
            b = (unsigned)(123 - (u < 0 ? 0 : (u > 123 ? 123 : u)));
        }
        a.bucket[p] = (unsigned char)b;
    }
    if (!on_list) return;
    if (cost > SMC_USER_SOLO_COST) {
        const unsigned k = atomicAdd(a.count + 1, 1u);
        if (k < a.solo_cap) {
            a.stiff_list[a.stiff_cap - 1 - (long long)k] = (int)p;
            return;
        }
    }
    a.stiff_list[atomicAdd(a.count, 1u)] = (int)p;
}
#endif
)SRC";

// log-likelihood of Micmem_likelihood.py:62-73 per particle from the per-item sums; counters as in the built-in path
__global__ void __launch_bounds__(256)
user_finish_kernel(const double *__restrict__ theta, int64_t stride, int64_t n, int dim, uint8_t *__restrict__ p0mask,
                   const double *__restrict__ sum_r2, const int *__restrict__ info, int n_ex, int n_t, int est_sigma,
                   double sigma_fixed, double *__restrict__ lk_out, SweepCounters *__restrict__ counters) {
    unsigned long long attempts = 0, failed = 0;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const bool masked = p0mask && p0mask[p] == 0;   // proposal reset to the current point: the stored likelihood is used
        const double sigma = est_sigma ? theta[(int64_t)(dim - 1) * stride + p] : sigma_fixed;
        if (!masked && sigma <= 0.0) lk_out[p] = -__longlong_as_double(0x7ff0000000000000LL);   // Micmem_likelihood.py:53-54
        if (!masked && sigma > 0.0) {
            const double s2 = sigma * sigma;
            const double c0 = (-0.5 * n_t) * log(2.0 * 3.141592653589793 * s2);
            double lk = 0.0;
            unsigned pf = 0, cancelled = 0;
            for (int e = 0; e < n_ex; ++e) {
                lk += c0 - sum_r2[(int64_t)e * n + p] / (2.0 * s2);
                const int fl = info[(int64_t)e * n + p];
                attempts += (unsigned)(fl & 0x1fffffff);
                pf |= (unsigned)(fl >> 30) & 1u;
                cancelled |= (unsigned)(fl >> 29) & 1u;
            }
            failed += pf;
            if (cancelled) {      // a solve stopped because the rejection was certain: logL was never completed - the accept
                lk = __longlong_as_double(0x7ff8000000000000LL);   // kernel sees the flag and keeps p_filt, lk1
                if (p0mask) p0mask[p] = 2;
            }
            lk_out[p] = lk;
        }
    }
    if (failed) atomicAdd(&counters->n_failed, failed);
    if (attempts) atomicAdd(&counters->rk_attempts, attempts);
}

struct UserModel {
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;
    hipFunction_t fn_scan = nullptr;      // smc_user_cost_scan_kernel: only for a source that defines smc_user_cost
    int32_t *d_list = nullptr;            // the two lists of a sweep (n_local entries), their flags and two alternating counter pairs
    uint8_t *d_listed = nullptr;
    unsigned *d_count = nullptr;
    int parity = 0;
    double *d_t = nullptr, *d_obs = nullptr, *d_cond = nullptr, *d_sum = nullptr;
    int *d_info = nullptr;
    int n_ex = 0, n_t = 0, n_cond = 0, n_states = 0, est_sigma = 1;
    int blocks_per_cu = 4;   // persistent blocks (4 waves each) per CU: what the compiled kernel's registers and LDS allow
    double sigma_fixed = 0, rtol = 1e-3, atol = 1e-6;
};

// the optional fourth ingredient: a source that mentions smc_user_cost must define it (include/smc_hip.h)
static bool has_cost_hint(const char *user_source) { return strstr(user_source, "smc_user_cost") != nullptr; }

static std::string build_source(const char *user_source, int n_states, int dim) {
    char head[256];
    snprintf(head, sizeof head, "#define SMC_USER_NS %d\n#define SMC_USER_DIM %d\n%s", n_states, dim,
             has_cost_hint(user_source) ? "#define SMC_USER_HAS_COST 1\n#define SMC_USER_LIST_COST 220.0\n#define SMC_USER_SOLO_COST 3700.0\n" : "");
    return std::string(head) + (strstr(user_source, "smc_div") ? "#define SMC_USER_USES_DIV 1\n" : "#define SMC_USER_USES_DIV 0\n") +
           kUserPreludeLean + "#line 1 \"user_model\"\n" + user_source + "\n" + kUserPreludeIeee + "#line 1 \"user_model\"\n" + user_source +
           "\n}  // namespace smc_user_ieee\n" + kUserKernelSource;
}

// compile for gfx950; on failure `log` holds hiprtc's diagnostics
static bool compile_user(const std::string &src, std::vector<char> &code, std::string &log) {
    hiprtcProgram prog;
    const char *headers[] = {k_sweep_args_h, k_philox_h, k_solve_sched_h, k_rk45_math_h};
    const char *names[] = {"sweep_args.h", "philox.h", "solve_sched.h", "rk45_math.h"};
    if (hiprtcCreateProgram(&prog, src.c_str(), "smc_user_model.hip", 4, headers, names) != HIPRTC_SUCCESS) {
        log = "hiprtcCreateProgram failed";
        return false;
    }
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=on", "-fno-fast-math"};
    const hiprtcResult r = hiprtcCompileProgram(prog, 4, opts);
    size_t ls = 0;
    if (hiprtcGetProgramLogSize(prog, &ls) == HIPRTC_SUCCESS && ls > 1) {
        log.resize(ls);
        (void)hiprtcGetProgramLog(prog, &log[0]);
    }
    bool ok = (r == HIPRTC_SUCCESS);
    if (ok) {
        size_t cs = 0;
        ok = hiprtcGetCodeSize(prog, &cs) == HIPRTC_SUCCESS && cs > 0;
        if (ok) {
            code.resize(cs);
            ok = hiprtcGetCode(prog, code.data()) == HIPRTC_SUCCESS;
        }
    }
    (void)hiprtcDestroyProgram(&prog);
    return ok;
}

void user_model_release(smc_ctx *c) {
    UserModel *u = (UserModel *)c->user;
    if (!u) return;
    (void)hipFree(u->d_t);
    (void)hipFree(u->d_obs);
    (void)hipFree(u->d_cond);
    (void)hipFree(u->d_sum);
    (void)hipFree(u->d_info);
    (void)hipFree(u->d_list);
    (void)hipFree(u->d_listed);
    (void)hipFree(u->d_count);
    if (u->module) (void)hipModuleUnload(u->module);
    delete u;
    c->user = nullptr;
}

// dynamic LDS of the compiled kernel: four waves' pools of started items, then the data times and observations
static size_t user_lds_bytes(int n_states, int n_ex, int n_t) {
    return ((size_t)4 * (2 * n_states + 6) * 64 + (size_t)2 * n_ex * (n_t + 1)) * sizeof(double);
}

static void launch_user_kernel(smc_ctx *c, const double *theta, int64_t stride, int64_t n, uint8_t *p0mask, double *lk,
                               bool reject) {
    UserModel *u = (UserModel *)c->user;
    UserSolveArgs a{};
    a.theta = theta;
    a.stride = stride;
    a.n = n;
    a.p0 = p0mask;
    a.t = u->d_t;
    a.obs = u->d_obs;
    a.cond = u->d_cond;
    a.n_ex = u->n_ex;
    a.n_t = u->n_t;
    a.n_cond = u->n_cond;
    a.dim = c->dim;
    a.est_sigma = u->est_sigma;
    a.sigma_fixed = u->sigma_fixed;
    a.rtol = u->rtol;
    a.atol = u->atol;
    a.sum_r2 = u->d_sum;
    a.info = u->d_info;
    a.queue = c->d_queue;
    a.rej = reject ? c->d_reject : nullptr;
    // persistent grid: the waves take chunks of kChunk items until the queue is empty (solve_sched.h); with a cost hint a
    // small sweep gets a wave per item so that solo solves do not queue behind each other (as mm_kernels.hip: solve_grid_blocks)
    const bool lists = u->fn_scan && c->stiff_first != 0;
    const int64_t chunks = (((n + 63) / 64) * 64 * u->n_ex + kChunk - 1) / kChunk;
    const int64_t need = (chunks + 3) / 4 + (lists ? (n * u->n_ex + 3) / 4 : 0);
    int64_t blocks = (int64_t)c->cu_count * u->blocks_per_cu;
    if (blocks > need) blocks = need;
    if (blocks < 1) blocks = 1;
    const unsigned lds = (unsigned)user_lds_bytes(u->n_states, u->n_ex, u->n_t);
    if (lists) {
        u->parity ^= 1;
        UserScanArgs sa{};
        sa.theta = theta;
        sa.stride = stride;
        sa.n = n;
        sa.p0 = p0mask;
        sa.listed = u->d_listed;
        sa.stiff_list = u->d_list;
        sa.count = u->d_count + 2 * u->parity;
        sa.count_next = u->d_count + 2 * (u->parity ^ 1);
        sa.stiff_cap = c->n_local;
        sa.solo_cap = (unsigned)(blocks * 4 / u->n_ex);      // one solo solve per wave of the grid
        // a heterogeneous Metropolis sweep large enough to have something to sort: cost order + in-phase waves, as for the
        // built-in model (mm_kernels.hip); the classes come from the model's hint
        const bool cost_order = p0mask != nullptr && c->cost_order != 0 && c->in_phase != 0 && n >= 16384 && c->d_bucket;
        if (cost_order) {
            sa.bucket = c->d_bucket;
            sa.done_sums = u->d_sum;
            sa.done_info = u->d_info;
            sa.n_ex = u->n_ex;
        }
        void *sargs[] = {&sa};
        const hipError_t e = hipModuleLaunchKernel(u->fn_scan, (unsigned)((n + 255) / 256), 1, 1, 256, 1, 1, 0, c->stream, sargs, nullptr);
        if (e != hipSuccess) {
            smc_fail(c, (std::string("launch of the user-model cost scan failed: ") + hipGetErrorString(e)).c_str());
            c->launch_failed = true;
            return;
        }
        a.listed = u->d_listed;
        a.stiff_list = u->d_list;
        a.stiff_count = sa.count;
        a.stiff_cap = sa.stiff_cap;
        a.solo_cap = sa.solo_cap;
        if (cost_order) {
            a.n_ordered = launch_cost_sort_order(c, n);
            if (a.n_ordered) {
                a.order = c->d_order;
                a.patience = kInPhasePatience;
            }
        }
    }
    void *args[] = {&a};
    {
        ScopedTimer tm(c, SMC_T_SOLVE);
        (void)hipMemsetAsync(c->d_queue, 0, sizeof(unsigned long long), c->stream);
        const hipError_t e = hipModuleLaunchKernel(u->fn, (unsigned)blocks, 1, 1, 256, 1, 1, lds, c->stream, args, nullptr);
        if (e != hipSuccess) {
            smc_fail(c, (std::string("launch of the user-model kernel failed: ") + hipGetErrorString(e)).c_str());
            c->launch_failed = true;
            return;
        }
    }
    const int64_t g = (n + 255) / 256;
    hipLaunchKernelGGL(user_finish_kernel, dim3((unsigned)(g < 1024 ? g : 1024)), dim3(256), 0, c->stream, theta, stride, n,
                       c->dim, p0mask, u->d_sum, u->d_info, u->n_ex, u->n_t, u->est_sigma, u->sigma_fixed, lk, c->d_counters);
}

void launch_user_loglik(smc_ctx *c, const double *theta, int64_t stride, int64_t n, double *lk) {
    if (n > 0) launch_user_kernel(c, theta, stride, n, nullptr, lk, false);
}

void launch_user_mh(smc_ctx *c, int64_t n, const MHParams &mh_in) {
    if (n <= 0) return;
    ParticleSet &P = c->set[SMC_SET_PRED];
    UserModel *u = (UserModel *)c->user;
    // exact early rejection (UserOps::certainly_rejected): off while the proposals' likelihoods are captured for inspection
    const bool reject = c->early_reject != 0 && c->debug_capture == 0 && mh_in.gamma > 0.0 && c->d_reject;
    MHParams mh = mh_in;
    if (reject) {
        mh.pending_sums = u->d_sum;          // the propose kernel marks every item of the sweep "not finished yet"
        mh.pending_n_ex = u->n_ex;
        mh.reject_out = c->d_reject;
        mh.reject_lk1 = c->set[SMC_SET_FILT].lk;
    }
    launch_generic_propose(c, n, mh);
    launch_user_kernel(c, P.theta, P.stride, n, c->d_p0, c->d_mlk2, reject);
    launch_generic_accept(c, n, mh, c->d_mlk2);
}

}  // namespace smc

using namespace smc;

extern "C" {

int smc_user_model_check(const char *source, int n_states, int dim, char *log, int log_cap) {
    if (!source || n_states < 1 || n_states > SMC_USER_MAX_STATES || dim < 1 || dim > SMC_MAX_DIM) return 2;
    std::vector<char> code;
    std::string lg;
    const bool ok = compile_user(build_source(source, n_states, dim), code, lg);
    if (log && log_cap > 0) {
        strncpy(log, lg.c_str(), (size_t)log_cap - 1);
        log[log_cap - 1] = 0;
    }
    return ok ? 0 : 1;
}

int smc_user_model_dump_source(const char *source, int n_states, int dim, const char *dir) {
    if (!source || !dir || n_states < 1 || n_states > SMC_USER_MAX_STATES || dim < 1 || dim > SMC_MAX_DIM) return 2;
    const std::string src = build_source(source, n_states, dim);
    const char *names[] = {"smc_user_model.hip", "sweep_args.h", "philox.h", "solve_sched.h", "rk45_math.h"};
    const char *texts[] = {src.c_str(), k_sweep_args_h, k_philox_h, k_solve_sched_h, k_rk45_math_h};
    for (int i = 0; i < 5; ++i) {
        FILE *f = fopen((std::string(dir) + "/" + names[i]).c_str(), "w");
        if (!f) return 1;
        const bool ok = fputs(texts[i], f) >= 0;
        if (fclose(f) != 0 || !ok) return 1;
    }
    return 0;
}

int smc_set_model_user(smc_ctx *c, const char *source, int n_states, const double *t, const double *obs, const double *cond,
                       int n_ex, int n_t, int n_cond, int est_sigma, double sigma_fixed, double rtol, double atol) {
    if (!c) return smc_fail(nullptr, "NULL context");
    if (!source) return smc_fail(c, "smc_set_model_user: NULL source");
    if (n_states < 1 || n_states > SMC_USER_MAX_STATES) return smc_fail(c, "smc_set_model_user: n_states out of range");
    if (n_ex < 1 || n_t < 1 || n_cond < 0) return smc_fail(c, "smc_set_model_user: bad data shape");
    if (hipSetDevice(c->device) != hipSuccess) return smc_fail(c, "hipSetDevice failed");
    std::vector<char> code;
    std::string lg;
    if (!compile_user(build_source(source, n_states, c->dim), code, lg)) {
        std::string msg = "user model does not compile:\n" + lg;
        if (msg.size() > 3500) msg.resize(3500);
        return smc_fail(c, msg.c_str());
    }
    (void)hipStreamSynchronize(c->stream);
    user_model_release(c);
    UserModel *u = new UserModel();
    c->user = u;
    if (hipModuleLoadData(&u->module, code.data()) != hipSuccess ||
        hipModuleGetFunction(&u->fn, u->module, "smc_user_solve_kernel") != hipSuccess) {
        user_model_release(c);
        return smc_fail(c, "smc_set_model_user: loading the compiled module failed");
    }
    if (has_cost_hint(source) && hipModuleGetFunction(&u->fn_scan, u->module, "smc_user_cost_scan_kernel") != hipSuccess) {
        user_model_release(c);
        return smc_fail(c, "smc_set_model_user: the source mentions smc_user_cost but the scan kernel is missing from the module");
    }
    const size_t nt = (size_t)n_ex * n_t * sizeof(double), nc = (size_t)n_ex * (n_cond > 0 ? n_cond : 1) * sizeof(double);
    bool ok = hipMalloc(&u->d_t, nt) == hipSuccess && hipMalloc(&u->d_obs, nt) == hipSuccess && hipMalloc(&u->d_cond, nc) == hipSuccess;
    ok = ok && hipMemcpy(u->d_t, t, nt, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(u->d_obs, obs, nt, hipMemcpyHostToDevice) == hipSuccess;
    if (ok && n_cond > 0) ok = hipMemcpy(u->d_cond, cond, nc, hipMemcpyHostToDevice) == hipSuccess;
    if (ok && !c->d_mlk2) ok = hipMalloc(&c->d_mlk2, (size_t)c->n_local * sizeof(double)) == hipSuccess;
    ok = ok && hipMalloc(&u->d_sum, (size_t)c->n_local * n_ex * sizeof(double)) == hipSuccess &&
         hipMalloc(&u->d_info, (size_t)c->n_local * n_ex * sizeof(int)) == hipSuccess;
    if (ok && u->fn_scan)
        ok = hipMalloc(&u->d_list, (size_t)c->n_local * sizeof(int32_t)) == hipSuccess &&
             hipMalloc(&u->d_listed, (size_t)c->n_local) == hipSuccess && hipMalloc(&u->d_count, 4 * sizeof(unsigned)) == hipSuccess &&
             hipMemset(u->d_count, 0, 4 * sizeof(unsigned)) == hipSuccess;
    if (!ok) {
        user_model_release(c);
        return smc_fail(c, "smc_set_model_user: device allocation / upload failed");
    }
    {   // occupancy of the compiled kernel with its pool in LDS
        int nb = 0;
        const size_t lds = user_lds_bytes(n_states, n_ex, n_t);
        if (lds > 150 * 1024) {
            user_model_release(c);
            return smc_fail(c, "smc_set_model_user: data set too large for the kernel's LDS table (n_ex * n_t * 16 B + pools > 150 KB)");
        }
        if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&nb, u->fn, 256, lds) == hipSuccess && nb >= 1) u->blocks_per_cu = nb;
        if (lds > 48 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void *>(u->fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            user_model_release(c);
            return smc_fail(c, "smc_set_model_user: raising the dynamic LDS limit of the compiled kernel failed");
        }
    }
    u->n_ex = n_ex;
    u->n_t = n_t;
    u->n_cond = n_cond;
    u->n_states = n_states;
    u->est_sigma = est_sigma;
    u->sigma_fixed = sigma_fixed;
    u->rtol = rtol;
    u->atol = atol;
    c->model_kind = 3;
    c->have_model = true;
    return 0;
}

}  // extern "C"
