/*
 * oracle/meth_dae_oracle.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU implementation of the time integration of the methanation DAE F(t, X, X'; p) = 0
 * (reaction(), SMC_methanation/methanation_set_likelihood.py:69-139) from the driver's initial guess
 * (SMC_methanation_main.py:47-58) to t = 75 s, and of my_model's outlet mapping (:204-208).
 *
 * PARITY UNPINNED.  The reference integrates with Assimulo's IDA (SUNDIALS; variable-order BDF, :167-198,
 * rtol = atol = 1e-6, suppress_alg, IDA_YA_YDP_INIT).  Assimulo/SUNDIALS are absent from this image, their
 * source is not under /root/reference, the reference's inlet table is missing and no output of my_model for
 * known inputs exists - IDA's step/order heuristics cannot be restated or checked.  What is implemented here is
 * an integrator of the SAME CLASS solving the SAME equations to the SAME tolerances:
 *     variable-order (1-5) BDF/NDF in the quasi-constant-step form of SciPy's BDF (scipy/integrate/_ivp/
 *     bdf.py, Shampine & Reichelt), adapted to the fully implicit residual: the corrector solves
 *     G(d) = F(t_new, y_pred + d, (psi + d)/c) = 0 by modified Newton with the iteration matrix
 *     dF/dy + (1/c) dF/dy', evaluated at the predictor at every step; error test on the differential
 *     variables only (suppress_alg); first step of order 1 from y'(0) = 0 (the reference's yd0).
 * It is checked by (tests/test_methanation_dae.py): agreement with SciPy's own BDF on an ODE written as
 * F = y' - f; tolerance refinement (1e-6 vs 1e-9 runs); vanishing residual of the steady state it reaches.
 * The HIP kernel implements the same algorithm and is compared with this file.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NX 51
#define NS 357
#define KB 13             /* half bandwidth in node-major ordering */
#define LDAB (3 * KB + 1) /* banded storage with room for pivoting fill-in */
#define MAX_ORDER 5
#define NEWTON_MAXITER 4

void meth_reaction(const double *X, const double *dX, const double *params, double *res);
void meth_flows(const double *y, double P_total, double S, double P_stp, double *F);

typedef struct {
    int64_t steps, rejects, nlu, nres, newton_fail;
    int32_t status; /* 0 ok, -1 step underflow / too many steps */
    int32_t order_hist[6];
} dae_stats;

typedef void (*resid_fn)(const double *y, const double *yd, const double *p, double *res, void *user);

static void meth_resid(const double *y, const double *yd, const double *p, double *res, void *user) {
    (void)user;
    meth_reaction(y, yd, p, res);
}

/* node-major index k = 7*i + f  <->  field-major f*51 + i */
static inline int fm_of_nm(int k) { return (k % 7) * NX + k / 7; }

/* banded LU with partial pivoting (LAPACK dgbtf2 layout: AB(kl+ku+1+i-j, j) = A(i,j), kl = ku = KB) */
static int band_factor(double *ab, int *piv, int n) {
    const int kl = KB, ku = KB, kv = kl + ku;
    for (int j = 0; j < n; ++j) {
        int km = (kl < n - 1 - j) ? kl : n - 1 - j;
        int jp = 0;
        double mx = fabs(ab[kv + j * LDAB]);
        for (int i = 1; i <= km; ++i) {
            double v = fabs(ab[kv + i + j * LDAB]);
            if (v > mx) { mx = v; jp = i; }
        }
        piv[j] = j + jp;
        if (mx == 0.0) return -1;
        int ju = (j + ku + jp < n - 1) ? j + ku + jp : n - 1; /* last column touched */
        if (jp != 0)
            for (int c = j; c <= ju; ++c) {
                double t = ab[kv + jp + j - c + c * LDAB];
                ab[kv + jp + j - c + c * LDAB] = ab[kv + j - c + c * LDAB];
                ab[kv + j - c + c * LDAB] = t;
            }
        double r = 1.0 / ab[kv + j * LDAB];
        for (int i = 1; i <= km; ++i) ab[kv + i + j * LDAB] *= r;
        for (int c = j + 1; c <= ju; ++c) {
            double t = ab[kv + j - c + c * LDAB];
            if (t != 0.0)
                for (int i = 1; i <= km; ++i) ab[kv + i + j - c + c * LDAB] -= ab[kv + i + j * LDAB] * t;
        }
    }
    return 0;
}
static void band_solve(const double *ab, const int *piv, int n, double *b) {
    const int kl = KB, ku = KB, kv = kl + ku;
    for (int j = 0; j < n; ++j) {
        int km = (kl < n - 1 - j) ? kl : n - 1 - j;
        int l = piv[j];
        if (l != j) { double t = b[l]; b[l] = b[j]; b[j] = t; }
        for (int i = 1; i <= km; ++i) b[j + i] -= ab[kv + i + j * LDAB] * b[j];
    }
    for (int j = n - 1; j >= 0; --j) {
        b[j] /= ab[kv + j * LDAB];
        int lo = (j - kv > 0) ? j - kv : 0;
        for (int i = lo; i < j; ++i) b[i] -= ab[kv + i - j + j * LDAB] * b[j];
    }
}

/* iteration matrix dG/dd = dF/dy + (1/c) dF/dy' at (y, yd) by coloured finite differences of the residual:
 * 21 evaluations (7 fields x 3 node classes), columns perturbed together never share a row. */
static void iteration_matrix(resid_fn F, void *user, const double *y, const double *yd, const double *p, double c,
                             double *ab, int n_fields, int64_t *nres) {
    static const double SQRT_EPS = 1.4901161193847656e-08;
    double f0[NS], f1[NS], yp[NS], ydp[NS], hcol[NS];
    memset(ab, 0, sizeof(double) * LDAB * NS);
    F(y, yd, p, f0, user);
    (*nres)++;
    const int kv = 2 * KB;
    for (int f = 0; f < n_fields; ++f)
        for (int cls = 0; cls < 3; ++cls) {
            memcpy(yp, y, sizeof yp);
            memcpy(ydp, yd, sizeof ydp);
            for (int i = cls; i < NX; i += 3) {
                int col = f * NX + i;
                double h = SQRT_EPS * fmax(fabs(y[col]), 1e-3);
                double tmp = y[col] + h;
                h = tmp - y[col];
                hcol[col] = h;
                yp[col] = tmp;
                ydp[col] = yd[col] + h / c;
            }
            F(yp, ydp, p, f1, user);
            (*nres)++;
            for (int i = cls; i < NX; i += 3) {
                int col = f * NX + i, jc = 7 * i + f;
                for (int ii = i - 1; ii <= i + 1; ++ii) {
                    if (ii < 0 || ii >= NX) continue;
                    for (int g = 0; g < 7; ++g) {
                        int row = g * NX + ii, ir = 7 * ii + g;
                        ab[kv + ir - jc + jc * LDAB] = (f1[row] - f0[row]) / hcol[col];
                    }
                }
            }
        }
}

static double rms_masked(const double *v, const double *scale, const unsigned char *mask, int n) {
    double s = 0.0;
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (!mask || mask[i]) {
            double q = v[i] / scale[i];
            s += q * q;
            m++;
        }
    return sqrt(s / (m ? m : 1));
}

/* scipy bdf.py: compute_R / change_D */
static void change_D(double D[][NS], int order, double factor, int n) {
    double R[6][6], U[6][6], RU[6][6], tmp[6];
    for (int pass = 0; pass < 2; ++pass) {
        double (*M)[6] = pass ? U : R;
        double fac = pass ? 1.0 : factor;
        for (int j = 0; j <= order; ++j) M[0][j] = 1.0;
        for (int i = 1; i <= order; ++i) {
            M[i][0] = 0.0;
            for (int j = 1; j <= order; ++j) M[i][j] = M[i - 1][j] * ((i - 1 - fac * j) / i);
        }
        /* column 0 of cumprod: M[0][0] = 1, M[i][0] = 0 */
    }
    for (int i = 0; i <= order; ++i)
        for (int j = 0; j <= order; ++j) {
            double s = 0.0;
            for (int k = 0; k <= order; ++k) s += R[i][k] * U[k][j];
            RU[i][j] = s;
        }
    for (int x = 0; x < n; ++x) {
        for (int j = 0; j <= order; ++j) {
            double s = 0.0;
            for (int i = 0; i <= order; ++i) s += RU[i][j] * D[i][x];
            tmp[j] = s;
        }
        for (int j = 0; j <= order; ++j) D[j][x] = tmp[j];
    }
}

/*
 * Control policy of the integrator (round 5).  The DEFAULT (all zero) is the checker of rounds 1-4: SciPy's BDF control with the
 * iteration matrix evaluated at the predictor of every attempt.  The other settings restate, inside the same quasi-constant-step
 * BDF, the control policy of IDA, the reference's actual integrator (SUNDIALS IDA, "Mathematical considerations"; Assimulo
 * configures it at methanation_set_likelihood.py:167-198) - K8 uses them (csrc/meth_dae_elem.h), and this file is the CPU
 * statement of what K8 does, compared with the default policy and with tight-tolerance runs (tests/test_methanation_dae.py):
 *   reuse   0  iteration matrix at the predictor of every attempt
 *           1  kept while c = h / alpha_k is unchanged (SciPy; K8 of rounds 1-4)
 *           2  kept while cj / cj_at_evaluation stays inside ((1 - xrate) / (1 + xrate), (1 + xrate) / (1 - xrate)), xrate = 0.25,
 *              i.e. (0.6, 1.667), the Newton correction scaled by 2 / (1 + cjratio)  (IDA: idaNls, idaLsSolve)
 *   newton  0  SciPy's test: rate / (1 - rate) * |dy| < newton_tol = max(10 eps / rtol, min(0.03, sqrt(rtol))), rate from two
 *              iterations of THIS step
 *           1  IDA's: converged when ss * |dy| <= epcon (0.33); ss = rate / (1 - rate) is CARRIED from step to step, reset to 20
 *              when the matrix is evaluated and to 100 when cj changed since the previous step; first iteration also converges
 *              on |dy| <= 1e-4 epcon; rate = (|dy_m| / |dy_0|)^(1/m) > 0.9 fails the iteration
 *   stepctl 0  SciPy: after order + 1 equal steps, h *= min(10, safety * best factor) whatever the factor
 *           1  IDA: h doubles when the factor is >= 2, shrinks by clamp(factor, 0.5, 0.9) when it is < 1, and is otherwise KEPT
 *           2  as 1, but a factor >= 2 is applied as SciPy would (min(10, ...)): the start-up phase stays short
 */
typedef struct {
    int32_t reuse, newton, stepctl, reserved;
    double epcon, xrate;
} dae_policy;

typedef struct {
    int64_t newton_iters, stale_retries;
} dae_stats_ext;

/*
 * Integrate F(y, y'; p) = 0 from (0, y0) with y'(0) = 0 to tf.  diff_mask[i] = 1 for differential variables
 * (error test), n = number of active unknowns (<= NS; the generic entry is used by the unit test on an ODE).
 */
int dae_bdf_integrate_policy(resid_fn F, void *user, const double *y0, const double *p, int n, int n_fields,
                             const unsigned char *diff_mask, double tf, double rtol, double atol, double h0, double *y_out,
                             dae_stats *st, const dae_policy *pol, dae_stats_ext *ext) {
    static const double kappa[6] = {0, -0.1850, -1.0 / 9, -0.0823, -0.0415, 0};
    static const dae_policy checker = {0, 0, 0, 0, 0.33, 0.25};
    if (!pol) pol = &checker;
    double gamma[6], alpha[6], error_const[7];
    gamma[0] = 0.0;
    for (int k = 1; k <= MAX_ORDER; ++k) gamma[k] = gamma[k - 1] + 1.0 / k;
    for (int k = 0; k <= MAX_ORDER; ++k) alpha[k] = (1 - kappa[k]) * gamma[k];
    for (int k = 0; k <= MAX_ORDER; ++k) error_const[k] = kappa[k] * gamma[k] + 1.0 / (k + 1);
    error_const[MAX_ORDER + 1] = 1.0 / (MAX_ORDER + 2);

    double(*D)[NS] = calloc(MAX_ORDER + 3, sizeof *D);
    double *ab = malloc(sizeof(double) * LDAB * NS);
    int piv[NS];
    double y_pred[NS], psi[NS], scale[NS], y[NS], d[NS], r[NS], dy[NS], ydot[NS], tmpv[NS];
    memset(st, 0, sizeof *st);
    if (ext) memset(ext, 0, sizeof *ext);
    memcpy(D[0], y0, sizeof(double) * NS);
    const double newton_tol = fmax(10 * 2.220446049250313e-16 / rtol, fmin(0.03, sqrt(rtol)));
    const double cj_lo = (1 - pol->xrate) / (1 + pol->xrate), cj_hi = 1.0 / cj_lo;
    double t = 0.0, h_abs = h0;
    int order = 1, n_equal = 0, rc = 0;
    const int64_t max_steps = 200000;
    int lu_valid = 0, force_rebuild = 0;
    double c_lu = 0.0, c_last = 0.0, ss = 20.0;

    while (t < tf && rc == 0) {
        int accepted = 0, n_iter = 0;
        double error_norm = 0.0, safety = 0.9, t_new = t;
        while (!accepted) {
            if (h_abs < 1e-14 * fmax(1.0, t) || st->steps + st->rejects + st->newton_fail > max_steps) { rc = -1; break; }
            t_new = t + h_abs;
            if (t_new - tf > 0) {
                t_new = tf;
                change_D(D, order, fabs(t_new - t) / h_abs, NS);
                n_equal = 0;
            }
            const double h = t_new - t;
            h_abs = fabs(h);
            for (int i = 0; i < NS; ++i) {
                double s = 0.0, q = 0.0;
                for (int k = 0; k <= order; ++k) s += D[k][i];
                for (int k = 1; k <= order; ++k) q += D[k][i] * gamma[k];
                y_pred[i] = s;
                psi[i] = q / alpha[order];
                scale[i] = atol + rtol * fabs(s);
            }
            const double c = h / alpha[order];
            int fresh;
            if (pol->reuse == 0) fresh = 1;
            else if (pol->reuse == 1) fresh = !lu_valid || c != c_lu || force_rebuild;
            else {
                const double cjratio = c_lu / c;   /* cj / cj_old, cj = 1 / c */
                fresh = !lu_valid || force_rebuild || !(cjratio > cj_lo && cjratio < cj_hi);
            }
            if (pol->newton == 1 && c != c_last) ss = 100.0;
            c_last = c;
            int factored = lu_valid;
            if (fresh) {
                /* iteration matrix at the predictor */
                for (int i = 0; i < NS; ++i) ydot[i] = psi[i] / c;
                iteration_matrix(F, user, y_pred, ydot, p, c, ab, n_fields, &st->nres);
                st->nlu++;
                factored = band_factor(ab, piv, NS) == 0;
                lu_valid = factored;
                c_lu = c;
                force_rebuild = 0;
                ss = 20.0;
            }
            const double corr_scale = (pol->reuse == 2) ? 2.0 / (1.0 + c_lu / c) : 1.0;
            int converged = 0;
            if (factored) {
                memcpy(y, y_pred, sizeof y);
                memset(d, 0, sizeof d);
                double dy_norm_old = -1.0, dy_norm_first = 0.0;
                for (int k = 0; k < NEWTON_MAXITER; ++k) {
                    for (int i = 0; i < NS; ++i) ydot[i] = (psi[i] + d[i]) / c;
                    F(y, ydot, p, r, user);
                    st->nres++;
                    if (ext) ext->newton_iters++;
                    n_iter = k + 1;
                    int finite = 1;
                    for (int i = 0; i < n; ++i)
                        if (!isfinite(r[i])) finite = 0;
                    if (!finite) break;
                    for (int k2 = 0; k2 < NS; ++k2) {
                        int fm = fm_of_nm(k2);
                        tmpv[k2] = (fm < n) ? -r[fm] : 0.0;
                    }
                    band_solve(ab, piv, NS, tmpv);
                    for (int k2 = 0; k2 < NS; ++k2) dy[fm_of_nm(k2)] = tmpv[k2] * corr_scale;
                    const double dy_norm = rms_masked(dy, scale, NULL, n);
                    if (pol->newton == 0) {
                        double rate = -1.0;
                        if (dy_norm_old >= 0) rate = dy_norm / dy_norm_old;
                        if (rate >= 0 && (rate >= 1 || pow(rate, NEWTON_MAXITER - k) / (1 - rate) * dy_norm > newton_tol)) break;
                        for (int i = 0; i < n; ++i) { y[i] += dy[i]; d[i] += dy[i]; }
                        if (dy_norm == 0 || (rate >= 0 && rate / (1 - rate) * dy_norm < newton_tol)) { converged = 1; break; }
                        dy_norm_old = dy_norm;
                    } else {
                        for (int i = 0; i < n; ++i) { y[i] += dy[i]; d[i] += dy[i]; }
                        if (k == 0) {
                            dy_norm_first = dy_norm;
                            if (dy_norm <= 1e-4 * pol->epcon) { converged = 1; break; }
                        } else {
                            const double q = dy_norm / dy_norm_first;
                            const double rate = (k == 1) ? q : (k == 2) ? sqrt(q) : cbrt(q);
                            if (!(rate <= 0.9)) break;
                            ss = rate / (1.0 - rate);
                        }
                        if (ss * dy_norm <= pol->epcon) { converged = 1; break; }
                    }
                }
            }
            if (!converged && !fresh) { /* stale matrix: the same attempt again with a fresh one */
                force_rebuild = 1;
                if (ext) ext->stale_retries++;
                continue;
            }
            if (!converged) {
                st->newton_fail++;
                lu_valid = 0;
                h_abs *= 0.5;
                change_D(D, order, 0.5, NS);
                n_equal = 0;
                continue;
            }
            safety = 0.9 * (2 * NEWTON_MAXITER + 1) / (2.0 * NEWTON_MAXITER + n_iter);
            for (int i = 0; i < NS; ++i) { scale[i] = atol + rtol * fabs(y[i]); tmpv[i] = error_const[order] * d[i]; }
            error_norm = rms_masked(tmpv, scale, diff_mask, n);
            if (error_norm > 1) {
                st->rejects++;
                double factor = fmax(0.2, safety * pow(error_norm, -1.0 / (order + 1)));
                h_abs *= factor;
                change_D(D, order, factor, NS);
                n_equal = 0;
            } else {
                accepted = 1;
            }
        }
        if (rc) break;
        n_equal++;
        t = t_new;
        st->steps++;
        st->order_hist[order]++;
        for (int i = 0; i < NS; ++i) {
            D[order + 2][i] = d[i] - D[order + 1][i];
            D[order + 1][i] = d[i];
            for (int k = order; k >= 0; --k) D[k][i] += D[k + 1][i];
        }
        if (n_equal < order + 1) continue;
        double em = INFINITY, ep = INFINITY;
        if (order > 1) {
            for (int i = 0; i < NS; ++i) tmpv[i] = error_const[order - 1] * D[order][i];
            em = rms_masked(tmpv, scale, diff_mask, n);
        }
        if (order < MAX_ORDER) {
            for (int i = 0; i < NS; ++i) tmpv[i] = error_const[order + 1] * D[order + 2][i];
            ep = rms_masked(tmpv, scale, diff_mask, n);
        }
        double fm = pow(em, -1.0 / order), f0 = pow(error_norm, -1.0 / (order + 1)), fp = pow(ep, -1.0 / (order + 2));
        int delta = 0; /* np.argmax: first maximum */
        double best = fm;
        delta = -1;
        if (f0 > best) { best = f0; delta = 0; }
        if (fp > best) { best = fp; delta = 1; }
        double factor = fmin(10.0, safety * best);
        if (pol->stepctl != 0) {
            if (factor >= 2.0) factor = (pol->stepctl == 1) ? 2.0 : factor;
            else if (factor < 1.0) factor = fmax(0.5, fmin(0.9, factor));
            else factor = 1.0;
        }
        if (pol->stepctl != 0 && factor == 1.0) {
            /* the step size stays: the differences of the new order are valid as they are (R(1) U = I); n_equal keeps counting,
             * so the selection is looked at again after the next step */
            order += delta;
            continue;
        }
        order += delta;
        h_abs *= factor;
        change_D(D, order, factor, NS);
        n_equal = 0;
    }
    memcpy(y_out, D[0], sizeof(double) * NS);
    st->status = rc;
    free(D);
    free(ab);
    return rc;
}

int dae_bdf_integrate(resid_fn F, void *user, const double *y0, const double *p, int n, int n_fields,
                      const unsigned char *diff_mask, double tf, double rtol, double atol, double h0, double *y_out,
                      dae_stats *st) {
    return dae_bdf_integrate_policy(F, user, y0, p, n, n_fields, diff_mask, tf, rtol, atol, h0, y_out, st, NULL, NULL);
}

/* the methanation DAE under a given control policy (NULL: the checker's default) */
int meth_dae_solve_policy(const double *y0, const double *p, double tf, double rtol, double atol, double h0, double *y_out,
                          dae_stats *st, const dae_policy *pol, dae_stats_ext *ext) {
    unsigned char mask[NS];
    for (int i = 0; i < NS; ++i) mask[i] = (i < 6 * NX);
    return dae_bdf_integrate_policy(meth_resid, NULL, y0, p, NS, 7, mask, tf, rtol, atol, h0, y_out, st, pol, ext);
}

/* my_model (methanation_set_likelihood.py:144-277) for ONE experiment: integrate, map the outlet node to the
 * five standard-state flows (:204-208); on failure the reference's sentinel -10000 (:244-249). */
int meth_dae_solve(const double *y0, const double *p, double tf, double rtol, double atol, double h0, double *y_out,
                   dae_stats *st) {
    unsigned char mask[NS];
    for (int i = 0; i < NS; ++i) mask[i] = (i < 6 * NX);
    return dae_bdf_integrate(meth_resid, NULL, y0, p, NS, 7, mask, tf, rtol, atol, h0, y_out, st);
}

void meth_model_one(const double *y0, const double *p, double S, double P_stp, double *flows, double *y_final,
                    dae_stats *st) {
    double y[NS];
    int rc = meth_dae_solve(y0, p, 75.0, 1e-6, 1e-6, 1e-5, y, st);
    if (rc != 0) {
        for (int f = 0; f < 5; ++f) flows[f] = -10000.0;
    } else {
        double P_total = (p[0] + p[1] + p[2] + p[3] + p[4]) * 8.3144589 * p[5]; /* :165 */
        meth_flows(y, P_total, S, P_stp, flows);
    }
    if (y_final) memcpy(y_final, y, sizeof y);
}

/* ---- generic ODE test problem: F = y' - f(y), Robertson-like stiff kinetics on 3 unknowns ----
 * (fields 0..2 at node 0 carry the three unknowns; everything else is padded with identity rows) */
static void ode_resid(const double *y, const double *yd, const double *p, double *res, void *user) {
    (void)user;
    for (int i = 0; i < NS; ++i) res[i] = y[i]; /* padding unknowns: y_i = 0 */
    const double a = y[0], b = y[NX], c = y[2 * NX];
    res[0] = yd[0] - (-p[0] * a + p[1] * b * c);
    res[NX] = yd[NX] - (p[0] * a - p[1] * b * c - p[2] * b * b);
    res[2 * NX] = yd[2 * NX] - (p[2] * b * b);
}
int dae_bdf_test_ode(const double *y0_3, const double *k3, double tf, double rtol, double atol, double h0, double *y3,
                     dae_stats *st) {
    double y0[NS], y[NS];
    unsigned char mask[NS];
    memset(y0, 0, sizeof y0);
    memset(mask, 0, sizeof mask);
    y0[0] = y0_3[0]; y0[NX] = y0_3[1]; y0[2 * NX] = y0_3[2];
    mask[0] = mask[NX] = mask[2 * NX] = 1;
    int rc = dae_bdf_integrate(ode_resid, NULL, y0, k3, NS, 7, mask, tf, rtol, atol, h0, y, st);
    y3[0] = y[0]; y3[1] = y[NX]; y3[2] = y[2 * NX];
    return rc;
}

/* =====================================================================================================================
 * A SECOND integrator for cross-checking (round 5): IDA's own algorithm - the fixed-leading-coefficient, variable-step,
 * variable-order (1-5) BDF of DASSL / SUNDIALS IDA on modified divided differences phi[j] - restated from its published
 * description (Brenan, Campbell, Petzold, "Numerical Solution of Initial-Value Problems in DAEs", ch. 5; SUNDIALS IDA user
 * guide, "Mathematical considerations"; the routines named below are IDA's).  The reference integrates with this method
 * (Assimulo's IDA, methanation_set_likelihood.py:167-198: rtol = atol = 1e-6, suppress_alg, ncp = 10 output points in NORMAL
 * mode, so the state at t = 75 is the INTERPOLANT of the last step's polynomial).  K8 and the checker above use another
 * formulation of the same family (SciPy's quasi-constant-step differences) with IDA's control policy; this routine answers
 * "how far from IDA's METHOD are they" as far as that can be answered without IDA: STILL PARITY-UNPINNED - it is a restatement
 * by the builder, never compared with SUNDIALS itself (absent from the image), and it leaves out IDACalcIC (the run starts from
 * y'(0) = 0 like K8; the first, order-1 step absorbs the inconsistency; t = 75 is near the steady state).
 *   IDASetCoeffs     psi, alpha, beta, sigma, gamma, alphas, alpha0, cj = -alphas / h, ck
 *   IDAPredict       y = sum phi[j], y' = sum gamma[j] phi[j]
 *   IDANls/NewtonIter matrix dF/dy + cj dF/dy' kept while cj / cjold in (0.6, 1/0.6), correction scaled 2 / (1 + cjratio);
 *                    converged when ss * |delta| <= 0.33, ss carried (20 after a setup, 100 when cj changed); maxcor 4, rate > 0.9 fails
 *   IDATestError     err_k = sigma[k] |ee|, lower-order estimates from ee + phi[k] (+ phi[k-1]); test ck |ee| <= 1
 *   IDAHandleNFlag   error-test failures: 0.9 (2 err + 1e-4)^(-1/(k+1)) in [0.25, 0.9], then 0.25, then order 1; Newton failure: 0.25
 *   IDACompleteStep  phase 0 (order up, h doubled every step) until a failure / order reduction / order 5; then the order
 *                    decision from terr_{k-1}, terr_k, terr_{k+1} and h doubled if rr >= 2, reduced to [0.5, 0.9] h if rr <= 1
 *   IDAGetSolution   interpolation at the output point
 * ===================================================================================================================== */
typedef struct {
    int64_t steps, netf, ncfn, nsetups, nni, nres;
    int32_t status;
    int32_t order_hist[6];
} ida_stats;

static double wrms(const double *v, const double *ewt, const unsigned char *mask, int n) {
    double s = 0.0;
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (!mask || mask[i]) {
            const double q = v[i] * ewt[i];
            s += q * q;
            m++;
        }
    return sqrt(s / (m ? m : 1));
}

int dae_ida_integrate(resid_fn F, void *user, const double *y0, const double *p, int n, int n_fields, const unsigned char *diff_mask,
                      double tout_final, int ncp, double rtol, double atol, double *y_out, ida_stats *st) {
    enum { MXORD = 5, MAXCOR = 4, MAXNCF = 10, MAXNEF = 10 };
    const double EPCON = 0.33, XRATE = 0.25, RATEMAX = 0.9;
    double(*phi)[NS] = calloc(MXORD + 2, sizeof *phi);
    double *ab = malloc(sizeof(double) * LDAB * NS);
    int piv[NS];
    double psi[MXORD + 2] = {0}, alpha[MXORD + 2] = {0}, beta[MXORD + 2] = {0}, sigma[MXORD + 2] = {0}, gamma[MXORD + 2] = {0};
    double yy[NS], yp[NS], ee[NS], ewt[NS], delta[NS], res[NS], tmpv[NS], tv[NS];
    memset(st, 0, sizeof *st);
    memcpy(phi[0], y0, sizeof(double) * NS); /* phi[1] = h * y'(0) = 0 */
    const unsigned char *emask = diff_mask;  /* suppress_alg: error tests on the differential variables; Newton norm on all */
    double tn = 0.0;
    /* first output point and initial step (IDASolve: hh = 0.001 * tdist, limited by 0.5 / |y'(0)| - not binding with y'(0) = 0) */
    double hh = 0.001 * (tout_final / ncp);
    int kk = 1, kused = 0, knew = 1, phase = 0, ns = 0, rc = 0;
    double hused = 0.0, cj = 1.0 / hh, cjold = cj, cjlast = cj, ss = 20.0, ck = 1.0;
    psi[0] = hh;
    int have_matrix = 0;
    const double cj_lo = (1 - XRATE) / (1 + XRATE), cj_hi = 1.0 / cj_lo;
    const int64_t max_steps = 200000;

    for (int icp = 1; icp <= ncp && rc == 0; ++icp) {
        const double tout = tout_final * icp / ncp;
        while (tn < tout && rc == 0) { /* IDAStep */
            if (st->steps > max_steps) { rc = -1; break; }
            for (int i = 0; i < NS; ++i) ewt[i] = 1.0 / (rtol * fabs(phi[0][i]) + atol);
            const double saved_t = tn;
            int ncf = 0, nef = 0;
            double err_k = 0.0, err_km1 = 0.0;
            for (;;) { /* attempts */
                /* ---- IDASetCoeffs ---- */
                if (hh != hused || kk != kused) ns = 0;
                ns = (ns + 1 < kused + 2) ? ns + 1 : kused + 2;
                if (kk + 1 >= ns) {
                    beta[0] = 1.0; alpha[0] = 1.0; gamma[0] = 0.0; sigma[0] = 1.0;
                    double temp1 = hh;
                    for (int i = 1; i <= kk; ++i) {
                        const double temp2 = psi[i - 1];
                        psi[i - 1] = temp1;
                        beta[i] = beta[i - 1] * psi[i - 1] / temp2;
                        temp1 = temp2 + hh;
                        alpha[i] = hh / temp1;
                        sigma[i] = i * sigma[i - 1] * alpha[i];
                        gamma[i] = gamma[i - 1] + alpha[i - 1] / hh;
                    }
                    psi[kk] = temp1;
                }
                double alphas = 0.0, alpha0 = 0.0;
                for (int i = 0; i < kk; ++i) { alphas -= 1.0 / (i + 1); alpha0 -= alpha[i]; }
                cjlast = cj;
                cj = -alphas / hh;
                ck = fabs(alpha[kk] + alphas - alpha0);
                if (alpha[kk] > ck) ck = alpha[kk];
                for (int j = ns; j <= kk; ++j)
                    for (int i = 0; i < NS; ++i) phi[j][i] *= beta[j];
                tn = tn + hh;
                /* ---- IDANls ---- */
                int call_setup = !have_matrix;
                {
                    const double cjratio = cj / cjold;
                    if (cjratio < cj_lo || cjratio > cj_hi) call_setup = 1;
                    if (cj != cjlast) ss = 100.0;
                }
                int nflag = 0; /* 0 ok, 1 Newton failed (recoverable), 2 error test failed */
                for (int pass = 0; pass < 2; ++pass) {
                    for (int i = 0; i < NS; ++i) { /* IDAPredict */
                        double s = 0.0, q = 0.0;
                        for (int j = 0; j <= kk; ++j) s += phi[j][i];
                        for (int j = 1; j <= kk; ++j) q += gamma[j] * phi[j][i];
                        yy[i] = s;
                        yp[i] = q;
                        ee[i] = 0.0;
                    }
                    int lu_ok = 1;
                    if (call_setup) {
                        iteration_matrix(F, user, yy, yp, p, 1.0 / cj, ab, n_fields, &st->nres);
                        st->nsetups++;
                        lu_ok = band_factor(ab, piv, NS) == 0;
                        have_matrix = lu_ok;
                        cjold = cj;
                        ss = 20.0;
                    }
                    const double cjratio = cj / cjold;
                    int conv = 0;
                    if (lu_ok) { /* IDANewtonIter */
                        double oldnrm = 0.0;
                        for (int m = 0; m < MAXCOR; ++m) {
                            F(yy, yp, p, res, user);
                            st->nres++;
                            st->nni++;
                            int finite = 1;
                            for (int i = 0; i < n; ++i)
                                if (!isfinite(res[i])) finite = 0;
                            if (!finite) break;
                            for (int k2 = 0; k2 < NS; ++k2) {
                                const int fm = fm_of_nm(k2);
                                tmpv[k2] = (fm < n) ? res[fm] : 0.0;
                            }
                            band_solve(ab, piv, NS, tmpv);
                            const double sc = (cjratio != 1.0) ? 2.0 / (1.0 + cjratio) : 1.0;
                            for (int k2 = 0; k2 < NS; ++k2) delta[fm_of_nm(k2)] = tmpv[k2] * sc;
                            for (int i = 0; i < n; ++i) { yy[i] -= delta[i]; ee[i] -= delta[i]; yp[i] -= cj * delta[i]; }
                            const double delnrm = wrms(delta, ewt, NULL, n);
                            if (m == 0) {
                                oldnrm = delnrm;
                                if (delnrm <= 1e-4 * EPCON) { conv = 1; break; }
                            } else {
                                const double rate = pow(delnrm / oldnrm, 1.0 / m);
                                if (rate > RATEMAX) break;
                                ss = rate / (1.0 - rate);
                            }
                            if (ss * delnrm <= EPCON) { conv = 1; break; }
                        }
                    }
                    if (conv) { nflag = 0; break; }
                    nflag = 1;
                    if (call_setup) break;   /* failed with current Jacobian data */
                    call_setup = 1;          /* retry this step with a fresh matrix (same predicted values) */
                }
                /* ---- IDATestError ---- */
                if (nflag == 0) {
                    const double enorm_k = wrms(ee, ewt, emask, n);
                    err_k = sigma[kk] * enorm_k;
                    const double terr_k = (kk + 1) * err_k;
                    knew = kk;
                    if (kk > 1) {
                        for (int i = 0; i < NS; ++i) tv[i] = phi[kk][i] + ee[i];
                        err_km1 = sigma[kk - 1] * wrms(tv, ewt, emask, n);
                        const double terr_km1 = kk * err_km1;
                        if (kk > 2) {
                            for (int i = 0; i < NS; ++i) tv[i] += phi[kk - 1][i];
                            const double terr_km2 = (kk - 1) * sigma[kk - 2] * wrms(tv, ewt, emask, n);
                            if (fmax(terr_km1, terr_km2) <= terr_k) knew = kk - 1;
                        } else if (terr_km1 <= 0.5 * terr_k) {
                            knew = kk - 1;
                        }
                    }
                    if (ck * enorm_k > 1.0) nflag = 2;
                }
                if (nflag == 0) break;
                /* ---- IDARestore + IDAHandleNFlag ---- */
                tn = saved_t;
                for (int j = 1; j <= kk; ++j) psi[j - 1] = psi[j] - hh;
                for (int j = ns; j <= kk; ++j)
                    for (int i = 0; i < NS; ++i) phi[j][i] /= beta[j];
                phase = 1;
                if (nflag == 1) {
                    st->ncfn++;
                    hh *= 0.25;
                    if (++ncf >= MAXNCF || fabs(hh) < 1e-14) { rc = -1; break; }
                } else {
                    st->netf++;
                    ++nef;
                    if (nef == 1) {
                        const double err_knew = (kk == knew) ? err_k : err_km1;
                        kk = knew;
                        double rr = 0.9 * pow(2.0 * err_knew + 1e-4, -1.0 / (kk + 1));
                        rr = fmax(0.25, fmin(0.9, rr));
                        hh *= rr;
                    } else if (nef == 2) {
                        kk = knew;
                        hh *= 0.25;
                    } else if (nef < MAXNEF) {
                        kk = 1;
                        hh *= 0.25;
                    } else { rc = -1; break; }
                    if (fabs(hh) < 1e-14) { rc = -1; break; }
                }
                if (st->steps == 0) psi[0] = hh; /* still the first step */
            }
            if (rc) break;
            /* ---- IDACompleteStep ---- */
            st->steps++;
            st->order_hist[kk]++;
            const int kdiff = kk - kused;
            kused = kk;
            hused = hh;
            if (knew == kk - 1 || kk == MXORD) phase = 1;
            if (phase == 0) {
                if (st->steps > 1) { kk++; hh = 2.0 * hh; }
            } else {
                int action = 0; /* 0 unset, 1 lower, 2 maintain, 3 raise */
                if (knew == kk - 1) action = 1;
                else if (kk == MXORD) action = 2;
                else if (kk + 1 >= ns || kdiff == 1) action = 2;
                double err_kp1 = 0.0;
                if (action == 0) {
                    for (int i = 0; i < NS; ++i) tv[i] = ee[i] - phi[kk + 1][i];
                    err_kp1 = wrms(tv, ewt, emask, n) / (kk + 2);
                    const double terr_k = (kk + 1) * err_k, terr_kp1 = (kk + 2) * err_kp1;
                    if (kk == 1) action = (terr_kp1 >= 0.5 * terr_k) ? 2 : 3;
                    else {
                        const double terr_km1 = kk * err_km1;
                        if (terr_km1 <= fmin(terr_k, terr_kp1)) action = 1;
                        else if (terr_kp1 >= terr_k) action = 2;
                        else action = 3;
                    }
                }
                double err_knew = err_k;
                if (action == 3) { kk++; err_knew = err_kp1; }
                else if (action == 1) { kk--; err_knew = err_km1; }
                double rr = pow(2.0 * err_knew + 1e-4, -1.0 / (kk + 1));
                if (rr >= 2.0) hh = 2.0 * hh;
                else if (rr <= 1.0) hh = hh * fmax(0.5, fmin(0.9, rr));
            }
            if (kused < MXORD) memcpy(phi[kused + 1], ee, sizeof(double) * NS);
            for (int i = 0; i < NS; ++i) phi[kused][i] += ee[i];
            for (int j = kused - 1; j >= 0; --j)
                for (int i = 0; i < NS; ++i) phi[j][i] += phi[j + 1][i];
        }
    }
    if (rc == 0) { /* IDAGetSolution at tout_final */
        const int kord = kused ? kused : 1;
        const double delt = tout_final - tn;
        double c = 1.0, gam = delt / psi[0];
        memcpy(y_out, phi[0], sizeof(double) * NS);
        for (int j = 1; j <= kord; ++j) {
            c = c * gam;
            gam = (delt + psi[j - 1]) / psi[j];
            for (int i = 0; i < NS; ++i) y_out[i] += c * phi[j][i];
        }
    } else {
        memcpy(y_out, phi[0], sizeof(double) * NS);
    }
    st->status = rc;
    free(phi);
    free(ab);
    return rc;
}

int meth_dae_solve_ida(const double *y0, const double *p, double tf, int ncp, double rtol, double atol, double *y_out, ida_stats *st) {
    unsigned char mask[NS];
    for (int i = 0; i < NS; ++i) mask[i] = (i < 6 * NX);
    return dae_ida_integrate(meth_resid, NULL, y0, p, NS, 7, mask, tf, ncp, rtol, atol, y_out, st);
}
/* the Robertson test problem of dae_bdf_test_ode through the IDA-style integrator */
int dae_ida_test_ode(const double *y0_3, const double *k3, double tf, double rtol, double atol, double *y3, ida_stats *st) {
    double y0[NS], y[NS];
    unsigned char mask[NS];
    memset(y0, 0, sizeof y0);
    memset(mask, 0, sizeof mask);
    y0[0] = y0_3[0]; y0[NX] = y0_3[1]; y0[2 * NX] = y0_3[2];
    mask[0] = mask[NX] = mask[2 * NX] = 1;
    int rc = dae_ida_integrate(ode_resid, NULL, y0, k3, NS, 7, mask, tf, 1, rtol, atol, y, st);
    y3[0] = y[0]; y3[1] = y[NX]; y3[2] = y[2 * NX];
    return rc;
}
