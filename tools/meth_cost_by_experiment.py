"""K8: how much do the 30 experiments differ in cost?  Solves of one experiment at a time (posterior-like parameters: within 2 %
of the generating values) - kernel time and BDF steps per solve - and the order the early-rejection sweeps would solve them in."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
M = pkg.methanation
cond = M.load_conditions(os.path.join(g.ROOT, "tests", "golden", "methanation_information.csv"))
guess = M.initial_guess(cond)
rs = np.random.RandomState(0)
n = 2048
rows = []
for e in range(30):
    prs = M.BASEPARAMS * (1.0 + 0.02 * rs.standard_normal((n, len(M.BASEPARAMS))))
    p0 = np.array([M.p0_rows(cond, pr)[e] for pr in prs])
    y0 = np.tile(guess[e], (n, 1))
    flows, status, _, info = pkg.methanation.dae_solve_batch(p0, y0)
    rows.append((e, info["kernel_ms"], info["steps"] / n, info["newton_iters"] / n, int((status != 0).sum())))
    print(f"experiment {e:2d}: {info['kernel_ms']:7.1f} ms for {n} solves ({n / info['kernel_ms'] * 1e3:8.0f} solves/s), steps/solve {info['steps'] / n:6.1f}, "
          f"newton/solve {info['newton_iters'] / n:6.1f}, failed {int((status != 0).sum())}", flush=True)
ms = np.array([r[1] for r in rows])
print(f"cost per solve: min {ms.min() / n * 1e3:.1f} us  max {ms.max() / n * 1e3:.1f} us  mean {ms.mean() / n * 1e3:.1f} us  max/min {ms.max() / ms.min():.2f}")
