/*
 * oracle/smc_oracle.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, IEEE double, no FMA contraction, no fast-math) of the
 * scalar arithmetic on the reference's particle hot path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / the reported CPU baseline.  The product path
 * (python-based-..._amd/) never links, imports or calls it.
 *
 * What is restated, and from where (paths relative to /root/reference unless they
 * start with scipy/, which is SciPy 1.15.3's scipy/integrate/_ivp/ - a third-party
 * dependency of the reference that is not vendored in it; version unpinned by the
 * reference, 1.15.3 is what this image holds):
 *
 *   mm_rhs                SMC_example/Micmem_likelihood.py:14-15   (mm_ode)
 *   rk45 solve            SMC_example/Micmem_likelihood.py:17-33   (simulate_mm_on_grid ->
 *                         scipy solve_ivp(method="RK45", rtol=1e-3, atol=1e-6, t_eval=t))
 *                           scipy/rk.py:14-71      rk_step (FSAL)
 *                           scipy/rk.py:8-11,111-176  step controller
 *                           scipy/rk.py:377-404    Dormand-Prince tableau (C, A, B, E, P)
 *                           scipy/rk.py:178-180,552-574  quartic dense output
 *                           scipy/common.py:63-65  RMS norm
 *                           scipy/common.py:68-134 select_initial_step
 *                           scipy/base.py:169-199  step(): finished test
 *                           scipy/ivp.py:653-723   t_eval dispatch (searchsorted side='right')
 *   mm_loglik             SMC_example/Micmem_likelihood.py:35-77   (log_likelihood_mm_multi)
 *   resample_residual_systematic
 *                         SMC_example/Micmem_SMC_main.py:147-184
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against the
 * golden vectors tests/golden/ (the .npz files) which tests/golden/make_golden.py produced by
 * running the reference itself (unmodified, seed 20250205, N=1000) in the build
 * container, and against scipy.integrate.solve_ivp called live.
 *
 * Summation orders: SciPy's np.dot calls go through BLAS, whose accumulation order
 * for these 1xs products is implementation defined; they are restated here as
 * plain left-to-right sums.  np.sum over the 40 residuals follows NumPy's pairwise
 * kernel (8 accumulators for n <= 128).  Remaining differences against the
 * reference are O(1 ulp) per operation and are covered by the tolerances written in
 * the tests.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define RK_SAFETY 0.9
#define RK_MIN_FACTOR 0.2
#define RK_MAX_FACTOR 10.0

/* scipy/rk.py:377-404 - written as the same Python float expressions (a/b in double) */
/* C (stage times) is listed for completeness: the Michaelis-Menten RHS is autonomous, so t + c*h is never used */
static const double RK_C[6] __attribute__((unused)) = {0.0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1.0};
static const double RK_A[6][5] = {
    {0, 0, 0, 0, 0},
    {1.0 / 5, 0, 0, 0, 0},
    {3.0 / 40, 9.0 / 40, 0, 0, 0},
    {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0},
    {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0},
    {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
static const double RK_B[6] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
static const double RK_E[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920, 17253.0 / 339200, -22.0 / 525,
                               1.0 / 40};
static const double RK_P[7][4] = {
    {1, -8048581381.0 / 2820520608, 8663915743.0 / 2820520608, -12715105075.0 / 11282082432},
    {0, 0, 0, 0},
    {0, 131558114200.0 / 32700410799, -68118460800.0 / 10900136933, 87487479700.0 / 32700410799},
    {0, -1754552775.0 / 470086768, 14199869525.0 / 1410260304, -10690763975.0 / 1880347072},
    {0, 127303824393.0 / 49829197408, -318862633887.0 / 49829197408, 701980252875.0 / 199316789632},
    {0, -282668133.0 / 205662961, 2019193451.0 / 616988883, -1453857185.0 / 822651844},
    {0, 40617522.0 / 29380423, -110615467.0 / 29380423, 69997945.0 / 29380423}};

typedef struct {
    int64_t n_attempts; /* rk_step calls inside _step_impl (accepted + rejected) */
    int64_t n_accepted;
    int64_t nfev;       /* RHS evaluations, counted as SciPy's solver.nfev does */
    int32_t status;     /* 0 finished, -1 failed (TOO_SMALL_STEP) */
    int32_t n_out;      /* number of t_eval points written */
} rk45_stats;

/* Micmem_likelihood.py:14-15:  return - Vmax * S / (Km + S)   ==  ((-Vmax)*S)/(Km+S) */
static inline double mm_rhs(double S, double Vmax, double Km) { return ((-Vmax) * S) / (Km + S); }

/* Python's builtin min(a, b) / max(a, b) on floats: return a unless b is strictly better */
static inline double py_min(double a, double b) { return (b < a) ? b : a; }
static inline double py_max(double a, double b) { return (b > a) ? b : a; }

/* scipy/common.py:63-65 for a size-1 vector: np.linalg.norm(x) / 1 ** 0.5 */
static inline double rms_norm1(double x) { return sqrt(x * x) / 1.0; }

/*
 * One solve_ivp(RK45) call for the scalar Michaelis-Menten ODE with t_eval.
 * y_out receives S(t_eval[i]); returns the stats.  t_eval must be increasing with
 * t_eval[0] >= t0 and t_eval[n-1] <= t_bound (ivp.py:598-609); the reference passes
 * t_span = (t[0], t[-1]) (Micmem_likelihood.py:22).
 */
rk45_stats oracle_mm_rk45(double Vmax, double Km, double S0, const double *t_eval, int n_t, double rtol,
                          double atol, double *y_out) {
    rk45_stats st;
    memset(&st, 0, sizeof st);
    const double t0 = t_eval[0], t_bound = t_eval[n_t - 1];
    double t = t0, y = S0;
    double K[7];

    /* RungeKutta.__init__ (rk.py:96-104): f = fun(t, y); h_abs = select_initial_step(...) */
    double f = mm_rhs(y, Vmax, Km);
    st.nfev++;
    double h_abs;
    {
        /* scipy/common.py:68-134, direction = +1, order = error_estimator_order = 4, max_step = inf */
        double interval_length = fabs(t_bound - t0);
        if (interval_length == 0.0) {
            h_abs = 0.0;
        } else {
            double scale = atol + fabs(y) * rtol;
            double d0 = rms_norm1(y / scale);
            double d1 = rms_norm1(f / scale);
            double h0;
            if (d0 < 1e-5 || d1 < 1e-5)
                h0 = 1e-6;
            else
                h0 = 0.01 * d0 / d1;
            h0 = py_min(h0, interval_length);
            double y1 = y + h0 * 1.0 * f;
            double f1 = mm_rhs(y1, Vmax, Km);
            st.nfev++;
            double d2 = rms_norm1((f1 - f) / scale) / h0;
            double h1;
            if (d1 <= 1e-15 && d2 <= 1e-15)
                h1 = py_max(1e-6, h0 * 1e-3);
            else
                h1 = pow(0.01 / py_max(d1, d2), 1.0 / (4 + 1));
            h_abs = py_min(py_min(py_min(100 * h0, h1), interval_length), INFINITY);
        }
    }
    const double error_exponent = -1.0 / (4 + 1);

    int i_out = 0; /* t_eval_i (ivp.py:611) */

    /* ivp.py:653: while status is None: solver.step() */
    for (;;) {
        /* base.py:181-187: t == t_bound before stepping -> finished without a step */
        if (t == t_bound) {
            /* t_old = t; t = t_bound; dense output would be ConstantDenseOutput */
            double t_old = t;
            (void)t_old;
            while (i_out < n_t && t_eval[i_out] <= t) y_out[i_out++] = y;
            st.status = 0;
            break;
        }
        /* ---- _step_impl (rk.py:111-176) ---- */
        double min_step = 10 * fabs(nextafter(t, INFINITY) - t);
        if (h_abs > INFINITY)
            h_abs = INFINITY;
        else if (h_abs < min_step)
            h_abs = min_step;
        int step_accepted = 0, step_rejected = 0, failed = 0;
        double h = 0, t_new = t, y_new = y, f_new = f;
        while (!step_accepted) {
            if (h_abs < min_step) {
                failed = 1;
                break;
            }
            h = h_abs * 1.0;
            t_new = t + h;
            if (1.0 * (t_new - t_bound) > 0) t_new = t_bound;
            h = t_new - t;
            h_abs = fabs(h);

            /* rk_step (rk.py:14-71) */
            K[0] = f;
            for (int s = 1; s < 6; s++) {
                double acc = 0.0;
                for (int j = 0; j < s; j++) acc += K[j] * RK_A[s][j];
                double dy = acc * h;
                K[s] = mm_rhs(y + dy, Vmax, Km);
            }
            {
                double acc = 0.0;
                for (int j = 0; j < 6; j++) acc += K[j] * RK_B[j];
                y_new = y + h * acc;
            }
            f_new = mm_rhs(y_new, Vmax, Km);
            K[6] = f_new;
            st.nfev += 6;
            st.n_attempts++;

            double scale = atol + fmax(fabs(y), fabs(y_new)) * rtol; /* np.maximum propagates NaN like fmax does not; see note */
            if (isnan(y) || isnan(y_new)) scale = NAN;
            double err;
            {
                double acc = 0.0;
                for (int j = 0; j < 7; j++) acc += K[j] * RK_E[j];
                err = acc * h;
            }
            double error_norm = rms_norm1(err / scale);

            if (error_norm < 1) {
                double factor;
                if (error_norm == 0)
                    factor = RK_MAX_FACTOR;
                else
                    factor = py_min(RK_MAX_FACTOR, RK_SAFETY * pow(error_norm, error_exponent));
                if (step_rejected) factor = py_min(1.0, factor);
                h_abs *= factor;
                step_accepted = 1;
            } else {
                h_abs *= py_max(RK_MIN_FACTOR, RK_SAFETY * pow(error_norm, error_exponent));
                step_rejected = 1;
            }
        }
        if (failed) {
            st.status = -1;
            break;
        }
        st.n_accepted++;
        double t_old = t, y_old = y;
        t = t_new;
        y = y_new;
        f = f_new;
        int finished = (1.0 * (t - t_bound) >= 0); /* base.py:196 */

        /* ---- ivp.py:700-720: outputs that fall in (t_old, t] (and t_eval == t0 on the first step) ---- */
        int i_new = i_out;
        while (i_new < n_t && t_eval[i_new] <= t) i_new++; /* searchsorted(t_eval, t, side='right') */
        if (i_new > i_out) {
            if (t == t_old) { /* base.py:212-214 ConstantDenseOutput */
                for (int i = i_out; i < i_new; i++) y_out[i] = y;
            } else {
                /* rk.py:178-180: Q = K.T.dot(P) */
                double Q[4];
                for (int k = 0; k < 4; k++) {
                    double acc = 0.0;
                    for (int j = 0; j < 7; j++) acc += K[j] * RK_P[j][k];
                    Q[k] = acc;
                }
                double hd = t - t_old; /* RkDenseOutput.__init__ */
                for (int i = i_out; i < i_new; i++) {
                    double x = (t_eval[i] - t_old) / hd;
                    double p1 = x, p2 = p1 * x, p3 = p2 * x, p4 = p3 * x; /* cumprod */
                    double acc = 0.0;
                    acc += Q[0] * p1;
                    acc += Q[1] * p2;
                    acc += Q[2] * p3;
                    acc += Q[3] * p4;
                    double yy = hd * acc;
                    yy += y_old;
                    y_out[i] = yy;
                }
            }
            i_out = i_new;
        }
        if (finished) {
            st.status = 0;
            break;
        }
    }
    st.n_out = i_out;
    return st;
}

/* NumPy's pairwise summation kernel (numpy/_core/src/umath/loops_utils.h.src, pairwise_sum) */
static double np_pairwise_sum(const double *a, ptrdiff_t n) {
    if (n < 8) {
        double res = 0.;
        for (ptrdiff_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        ptrdiff_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        ptrdiff_t n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
    }
}

double oracle_np_sum(const double *a, int64_t n) { return np_pairwise_sum(a, (ptrdiff_t)n); }

/*
 * Micmem_likelihood.py:35-77.  theta = [Vmax, Km, sigma]; t is n_ex x n_t (row per
 * experiment), P_obs likewise, S0[n_ex].  pred (optional, n_ex x n_t) receives P_model.
 * Returns logL; *n_failed counts experiments whose solve did not reach t_bound (the
 * reference raises there - ragged sol.y - which aborts the whole run; the oracle
 * returns NaN for such a particle instead).  stats_sum (optional) accumulates RK work.
 */
double oracle_mm_loglik(const double *theta, const double *t, const double *P_obs, const double *S0, int n_ex,
                        int n_t, int est_sigma, double sigma_fixed, double rtol, double atol, double *pred,
                        int *n_failed, int64_t *stats_sum /* [3]: attempts, accepted, nfev */) {
    double Vmax = theta[0], Km = theta[1];
    double sigma = est_sigma ? theta[2] : sigma_fixed;
    if (n_failed) *n_failed = 0;
    if (sigma <= 0) return -INFINITY; /* :53-54 */
    double logL_total = 0.0;
    double S_model[512];
    double r2[512];
    for (int i = 0; i < n_ex; i++) {
        const double *ti = t + (size_t)i * n_t;
        const double *Pi = P_obs + (size_t)i * n_t;
        rk45_stats st = oracle_mm_rk45(Vmax, Km, S0[i], ti, n_t, rtol, atol, S_model);
        if (stats_sum) {
            stats_sum[0] += st.n_attempts;
            stats_sum[1] += st.n_accepted;
            stats_sum[2] += st.nfev;
        }
        if (st.status != 0 || st.n_out != n_t) {
            if (n_failed) (*n_failed)++;
            for (int k = st.n_out; k < n_t; k++) S_model[k] = NAN;
        }
        for (int k = 0; k < n_t; k++) {
            double P_model = S0[i] - S_model[k];   /* :32 */
            if (pred) pred[(size_t)i * n_t + k] = P_model;
            double residual = Pi[k] - P_model;      /* :68 */
            r2[k] = residual * residual;            /* residual**2 */
        }
        double s2 = pow(sigma, 2.0);                /* sigma**2 on a NumPy float64 scalar -> pow() */
        double logL_i = -0.5 * n_t * log(2 * M_PI * s2) - np_pairwise_sum(r2, n_t) / (2 * s2); /* :70-71 */
        logL_total += logL_i;
    }
    return logL_total;
}

/* sim_particle (Micmem_likelihood.py:79-92): particle is (N,3) C-order; one evaluation per row. */
void oracle_mm_loglik_batch(const double *particle, int64_t N, const double *t, const double *P_obs,
                            const double *S0, int n_ex, int n_t, int est_sigma, double sigma_fixed, double rtol,
                            double atol, double *lk_out, double *pred_out /* optional N x n_ex x n_t */,
                            int64_t *n_failed_total, int64_t *stats_sum /* optional [3] */, int n_threads) {
    int64_t nf = 0, s0 = 0, s1 = 0, s2 = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : nf, s0, s1, s2)
#endif
    for (int64_t p = 0; p < N; p++) {
        int f = 0;
        int64_t ss[3] = {0, 0, 0};
        lk_out[p] = oracle_mm_loglik(particle + 3 * p, t, P_obs, S0, n_ex, n_t, est_sigma, sigma_fixed, rtol, atol,
                                     pred_out ? pred_out + (size_t)p * n_ex * n_t : NULL, &f, ss);
        nf += f;
        s0 += ss[0];
        s1 += ss[1];
        s2 += ss[2];
    }
    if (n_failed_total) *n_failed_total = nf;
    if (stats_sum) {
        stats_sum[0] = s0;
        stats_sum[1] = s1;
        stats_sum[2] = s2;
    }
}

/*
 * Micmem_SMC_main.py:147-184, the pure-Python residual-systematic resampling loop.
 *   p_weight  : normalised weights on entry (N), overwritten with the residuals (:150)
 *   p_is      : out, offspring counts (N)
 *   p_pred    : (N,d) ancestors;  lk: (N)
 *   p_filt/lk1: (N,d)/(N) persistent output buffers - rows >= total offspring keep
 *               their previous contents, exactly as in the reference
 * Returns the number of rows written (n at :184); if it would exceed N the reference
 * raises IndexError - here the copy stops at N and the return value is > N.
 */
int64_t oracle_resample_residual_systematic(double *p_weight, int64_t N, double wrand_u /* np.random.rand() */,
                                            int64_t *p_is, const double *p_pred, const double *lk, int d,
                                            double *p_filt, double *lk1, int64_t *n_tmp_out) {
    const double inv_Np = 1.0 / (double)N; /* Micmem_settings.py:17 */
    int64_t sum_is = 0;
    for (int64_t j = 0; j < N; j++) {
        p_is[j] = (int64_t)trunc(p_weight[j] * (double)N); /* :147 */
        sum_is += p_is[j];
    }
    for (int64_t j = 0; j < N; j++) p_weight[j] = p_weight[j] - (double)p_is[j] * inv_Np; /* :150 */
    int64_t n_tmp = N - sum_is;                                                            /* :153 */
    double wrand = wrand_u * inv_Np;                                                       /* :156 */
    double sum = 0.0;
    int64_t n = 0;
    for (int64_t j = 0; j < N; j++) {
        sum += p_weight[j];  /* :167 */
        if (sum >= wrand) {  /* :168 */
            p_is[j] += 1;
            wrand += inv_Np;
            n_tmp -= 1;
        }
        for (int64_t k = 0; k < p_is[j]; k++) {
            if (n < N) {
                for (int c = 0; c < d; c++) p_filt[n * d + c] = p_pred[j * d + c];
                lk1[n] = lk[j];
            }
            n += 1;
        }
    }
    if (n_tmp_out) *n_tmp_out = n_tmp;
    return n;
}

int oracle_abi_version(void) { return 1; }
