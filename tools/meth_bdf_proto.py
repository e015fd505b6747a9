"""Prototype: SciPy's variable-order BDF/NDF algorithm (scipy/integrate/_ivp/bdf.py) adapted to the fully
implicit DAE residual F(t, y, y') = 0 of the methanation model.  For sizing / validation only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.linalg import solve_banded
from scipy.integrate._ivp.bdf import change_D
from tools.meth_dae_proto import M, NX, NS, perm, DIFF, guess, p0_of, F, jac_banded, to_banded

MAX_ORDER, NEWTON_MAXITER, MIN_FACTOR, MAX_FACTOR = 5, 4, 0.2, 10
kappa = np.array([0, -0.1850, -1 / 9, -0.0823, -0.0415, 0])
gamma = np.hstack((0, np.cumsum(1 / np.arange(1, MAX_ORDER + 1))))
alpha = (1 - kappa) * gamma
error_const = kappa * gamma + 1 / np.arange(1, MAX_ORDER + 2)

def rms(v, mask=None):
    v = v if mask is None else v[mask]
    return np.sqrt(np.mean(v * v))

def integrate(y0, p, tf=75.0, rtol=1e-6, atol=1e-6, h0=1e-5, stats=None):
    st = stats if stats is not None else {}
    for k in ("steps", "rej", "nlu", "njac", "nres", "newton_fail"): st[k] = 0
    n = y0.size
    D = np.zeros((MAX_ORDER + 3, n)); D[0] = y0
    t, h_abs, order, n_equal, LU, J = 0.0, h0, 1, 0, None, None
    newton_tol = max(10 * 2.2e-16 / rtol, min(0.03, rtol ** 0.5))
    orders = []
    while t < tf:
        step_accepted = False
        current_jac = False
        while not step_accepted:
            t_new = t + h_abs
            if t_new - tf > 0:
                t_new = tf
                change_D(D, order, abs(t_new - t) / h_abs); n_equal = 0; LU = None
            h = t_new - t; h_abs = abs(h)
            y_predict = np.sum(D[:order + 1], axis=0)
            scale = atol + rtol * np.abs(y_predict)
            psi = np.dot(D[1:order + 1].T, gamma[1:order + 1]) / alpha[order]
            c = h / alpha[order]
            converged = False
            while not converged:
                if LU is None:
                    if J is None:
                        J = True
                    Jn = jac_banded(y_predict, psi / c, p, 1.0 / c); st["njac"] += 1   # prototype: FD with the current c
                    LU = to_banded(Jn); st["nlu"] += 1
                    current_jac = True
                d = np.zeros(n); y = y_predict.copy(); dy_old = None; converged = False; n_iter = 0
                for k in range(NEWTON_MAXITER):
                    r = F(y, (psi + d) / c, p); st["nres"] += 1; n_iter = k + 1
                    if not np.all(np.isfinite(r)): break
                    dyn = solve_banded((13, 13), LU, -r[perm]); dy = np.zeros(n); dy[perm] = dyn
                    dy_norm = rms(dy / scale)
                    rate = None if dy_old is None else dy_norm / dy_old
                    if rate is not None and (rate >= 1 or rate ** (NEWTON_MAXITER - k) / (1 - rate) * dy_norm > newton_tol): break
                    y += dy; d += dy
                    if dy_norm == 0 or (rate is not None and rate / (1 - rate) * dy_norm < newton_tol):
                        converged = True; break
                    dy_old = dy_norm
                if not converged:
                    if current_jac: break
                    LU = None
            if not converged:
                st["newton_fail"] += 1
                h_abs *= 0.5; change_D(D, order, 0.5); n_equal = 0; LU = None
                continue
            safety = 0.9 * (2 * NEWTON_MAXITER + 1) / (2 * NEWTON_MAXITER + n_iter)
            scale = atol + rtol * np.abs(y)
            error = error_const[order] * d
            error_norm = rms((error / scale), DIFF)
            if error_norm > 1:
                st["rej"] += 1
                factor = max(MIN_FACTOR, safety * error_norm ** (-1 / (order + 1)))
                h_abs *= factor; change_D(D, order, factor); n_equal = 0; LU = None   # c changes -> refactor
            else:
                step_accepted = True
        n_equal += 1; t = t_new; st["steps"] += 1; orders.append(order)
        D[order + 2] = d - D[order + 1]; D[order + 1] = d
        for i in reversed(range(order + 1)): D[i] += D[i + 1]
        if n_equal < order + 1: continue
        em = rms((error_const[order - 1] * D[order] / scale), DIFF) if order > 1 else np.inf
        ep = rms((error_const[order + 1] * D[order + 2] / scale), DIFF) if order < MAX_ORDER else np.inf
        norms = np.array([em, error_norm, ep])
        with np.errstate(divide="ignore"):
            factors = norms ** (-1 / np.arange(order, order + 3))
        order += np.argmax(factors) - 1
        factor = min(MAX_FACTOR, safety * np.max(factors))
        h_abs *= factor; change_D(D, order, factor); n_equal = 0; LU = None
    st["orders"] = np.bincount(orders, minlength=6).tolist()
    return D[0].copy()

if __name__ == "__main__":
    pr = M.BASEPARAMS
    for i in [0, 7, 19]:
        p = p0_of(i, pr); st = {}
        t0 = time.time(); y = integrate(guess[i], p, stats=st)
        print(f"expt {i}: {st} {time.time()-t0:.1f}s outlet C={y[[50,101,152,203,254]]} T={y[305]:.5f} u={y[356]:.6f}")
        st2 = {}; yt = integrate(guess[i], p, rtol=1e-9, atol=1e-9, stats=st2)
        print(f"   tol 1e-9: steps {st2['steps']}  max rel diff vs 1e-6 run: {np.max(np.abs(yt-y)/(np.abs(yt)+1e-6)):.3e}  outlet rel diff {np.max(np.abs(yt-y)[[50,101,152,203,254,305,356]]/np.abs(yt[[50,101,152,203,254,305,356]]+1e-9)):.3e}")
