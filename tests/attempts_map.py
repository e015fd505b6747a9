"""RK45 step attempts of one Michaelis-Menten solve over (Vmax, Km), by the CPU checker (and SciPy itself for a few points): the
longest chains of a sweep sit just above Km ~ 1.5e-3, where attempts ~ 3.7 Vmax / Km; below it solve_ivp's RK45 finishes in a
handful of steps (the first steps carry S through zero), so the longest chain of a prior population does not grow with its size.
Lives under tests/ because it uses the checker: python tests/attempts_map.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.integrate import solve_ivp
import __graft_entry__ as g
g.load_oracle()
from oracle import oracle as O
z = np.load(os.path.join(g.ROOT, "tests", "golden", "mm_data.npz"))
t, P, S0 = z["t"], z["P_obs"], z["S0"]
kms = 10.0 ** np.arange(-4.5, -0.9, 0.25)
print("longest of the six experiments' solves, attempts:")
for vmax in (0.5, 2.0, 5.0, 10.0):
    row = []
    for km in kms:
        per = []
        for e in range(6):
            d = O.MMData(t=t[e:e + 1], P_obs=P[e:e + 1], S0=S0[e:e + 1])
            per.append(O.mm_loglik_batch(np.array([[vmax, km, 1.0]]), d)[2]["n_attempts"])
        row.append(max(per))
    print(f"Vmax {vmax:4.1f}: " + "  ".join(f"Km {km:.0e}: {a}" for km, a in zip(kms, row)))
print("checker against scipy.integrate.solve_ivp on both sides of the cliff (logL of the six experiments, sigma = 1):")
d = O.MMData(t=t, P_obs=P, S0=S0)
for vmax, km in ((5.0, 1e-4), (10.0, 1e-3), (2.0, 1.3e-3), (10.0, 1.6e-3), (0.5, 3e-5)):
    lk, nfev = 0.0, 0
    for e in range(6):
        sol = solve_ivp(lambda tt, y: [-vmax * y[0] / (km + y[0])], [t[e, 0], t[e, -1]], [S0[e]], method="RK45", t_eval=t[e])
        nfev += sol.nfev
        r = P[e] - (S0[e] - sol.y[0])
        lk += -0.5 * len(t[e]) * np.log(2 * np.pi) - np.sum(r ** 2) / 2
    b, _, info = O.mm_loglik_batch(np.array([[vmax, km, 1.0]]), d)
    print(f"  Vmax {vmax}, Km {km:g}: SciPy {lk:.12g} (nfev {nfev}), checker {b[0]:.12g} ({info['n_attempts']} attempts), rel. diff {abs(lk - b[0]) / abs(lk):.1e}")
