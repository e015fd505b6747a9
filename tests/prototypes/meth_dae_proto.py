"""Prototype (not product, not oracle): implicit-Euler + Richardson DAE integrator for the methanation model,
to study step counts / stiffness before writing the C and HIP versions."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from scipy.linalg import solve_banded
from oracle import methanation as M

NX, NS = 51, 357
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
cond = M.load_conditions(os.path.join(GOLD, "methanation_information.csv"))
guess = M.initial_guess(cond)

# node-major permutation: k = i*7 + f  <->  field-major f*51 + i
perm = np.array([f * NX + i for i in range(NX) for f in range(7)])   # node-major index -> field-major index
DIFF = np.zeros(NS, bool); DIFF[:6 * NX] = True

def p0_of(i, pr):
    return np.array([cond["Ca_in"][i], cond["Cb_in"][i], cond["Cc_in"][i], cond["Cd_in"][i], cond["Ce_in"][i], cond["T_in"][i],
                     cond["T_jacket"][i], cond["u_in"][i], cond["void"][i], cond["reactorlength"][i] / (NX - 1), *pr[:8]])

def F(y, yd, p):
    return M.reaction(y, yd, p)

def jac_banded(y, yd, p, cj):
    """dF/dy + cj*dF/dyd by coloured finite differences (21 evaluations), returned in node-major dense form."""
    J = np.zeros((NS, NS))
    f0 = F(y, yd, p)
    for f in range(7):
        for c in range(3):
            idx = np.array([f * NX + i for i in range(c, NX, 3)])
            dy = np.zeros(NS)
            h = np.sqrt(2.2e-16) * np.maximum(np.abs(y[idx]), 1e-3)
            dy[idx] = h
            f1 = F(y + dy, yd + cj * dy, p)
            d = (f1 - f0)
            for k, col in enumerate(idx):
                i = col % NX
                for ii in (i - 1, i, i + 1):
                    if 0 <= ii < NX:
                        rows = np.arange(7) * NX + ii
                        J[rows, col] = d[rows] / h[k]
    return J[np.ix_(perm, perm)]   # rows/cols node-major

def to_banded(Jn, kl=13, ku=13):
    ab = np.zeros((kl + ku + 1, NS))
    for d in range(-kl, ku + 1):
        diag = np.diagonal(Jn, d)
        if d >= 0:
            ab[ku - d, d:] = diag
        else:
            ab[ku - d, :NS + d] = diag
    return ab

def wrms(v, y, rtol, atol, mask=DIFF):
    w = 1.0 / (rtol * np.abs(y) + atol)
    return np.sqrt(np.mean((v[mask] * w[mask]) ** 2))

stats = {"jac": 0, "res": 0, "newton": 0, "steps": 0, "rej": 0, "nfail": 0}

def euler_step(y0, h, p, rtol, atol, ypred=None):
    """solve F(y, (y-y0)/h) = 0 by chord Newton from y0."""
    y = y0.copy() if ypred is None else ypred.copy()
    cj = 1.0 / h
    Jn = jac_banded(y, (y - y0) * cj, p, cj); stats["jac"] += 1
    ab = to_banded(Jn)
    for it in range(8):
        r = F(y, (y - y0) * cj, p); stats["res"] += 1
        dx = solve_banded((13, 13), ab, -r[perm])
        d = np.zeros(NS); d[perm] = dx
        y = y + d
        stats["newton"] += 1
        nrm = wrms(d, y, rtol, atol, np.ones(NS, bool))
        if nrm < 1e-3:
            return y, True
        if not np.isfinite(nrm):
            break
    return y, False

def integrate(y0, p, tf=75.0, rtol=1e-6, atol=1e-6, h0=1e-4, verbose=False):
    t, y, h = 0.0, y0.copy(), h0
    while t < tf:
        h = min(h, tf - t)
        y1, ok1 = euler_step(y, h, p, rtol, atol)
        ok2 = ok3 = False
        if ok1:
            ya, ok2 = euler_step(y, h / 2, p, rtol, atol)
            if ok2:
                y2, ok3 = euler_step(ya, h / 2, p, rtol, atol)
        if not (ok1 and ok2 and ok3):
            stats["nfail"] += 1
            h *= 0.25
            if h < 1e-12: raise RuntimeError("step underflow")
            continue
        err = wrms(y2 - y1, y2, rtol, atol)
        if err <= 1.0:
            t += h; y = 2 * y2 - y1; stats["steps"] += 1
            if verbose and stats["steps"] % 20 == 0: print(f"  t={t:.4g} h={h:.3g} err={err:.3g} T_out={y[6*NX-1]:.3f}")
        else:
            stats["rej"] += 1
        h *= min(5.0, max(0.2, 0.9 * (1.0 / max(err, 1e-10)) ** 0.5))
    return y

if __name__ == "__main__":
    pr = M.BASEPARAMS
    for i in [0, 7]:
        p = p0_of(i, pr)
        for k in stats: stats[k] = 0
        t0 = time.time()
        y = integrate(guess[i], p, verbose=(i == 0))
        print(f"expt {i}: {stats} {time.time()-t0:.1f}s  outlet C={y[[50,101,152,203,254]]} T={y[305]:.4f} u={y[356]:.5f}")
        r = F(y, np.zeros(NS), p)
        print("   steady-state residual scaled:", np.abs(r).max())
        for k in stats: stats[k] = 0
        yt = integrate(guess[i], p, rtol=1e-8, atol=1e-8)
        print(f"   tol 1e-8: {stats}  max rel diff vs 1e-6: {np.max(np.abs(yt-y)/(np.abs(yt)+1e-6)):.3e}")
