"""Which K8 solves fail on the GPU over the prior box, and does the CPU checker's integrator solve them?
(A study script: it uses the checker under oracle/, so it lives under tests/.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
g.load_oracle()
from oracle import methanation as O
M = pkg.methanation
cond = M.load_conditions(os.path.join(g.ROOT, "tests", "golden", "methanation_information.csv"))
guess = M.initial_guess(cond)
lo, hi, pos = M.prior_box()
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_part = 300
prs = np.tile(M.BASEPARAMS, (n_part, 1))
prs[:, :4] = (lo[pos] + (hi[pos] - lo[pos]) * rs.uniform(0, 1, (n_part, 5)))[:, :4]
p0 = np.concatenate([M.p0_rows(cond, pr) for pr in prs])
y0 = np.concatenate([guess[:30] for _ in prs])
flows, status, _, info = M.dae_solve_batch(p0, y0)
bad = np.nonzero(status != 0)[0]
print(f"GPU: {len(bad)} of {len(status)} solves failed; particles affected: {len(set(bad // 30))}", flush=True)
t0 = time.time()
agree = 0
for k in bad[:40]:
    y, rc, st = O.dae_solve(y0[k], p0[k])
    agree += (rc != 0)
    frac = (prs[k // 30, :4] - lo[pos][:4]) / (hi[pos][:4] - lo[pos][:4])
    print(f"solve {k} (particle {k//30}, experiment {k%30}) box position {np.round(frac, 2)}: checker status {rc}, steps {st['steps']}, newton_fail {st['newton_fail']}", flush=True)
print(f"checker also fails on {agree} of {min(40, len(bad))} ({time.time()-t0:.0f} s)")
