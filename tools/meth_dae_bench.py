"""Throughput probe of the K8 DAE kernel: solves/s for batches of (particle, experiment) pairs drawn from the prior box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
M = pkg.methanation
cond = M.load_conditions(os.path.join(g.ROOT, "tests", "golden", "methanation_information.csv"))
guess = M.initial_guess(cond)
lo, hi, pos = M.prior_box()
rs = np.random.RandomState(0)
for n_part in [int(a) for a in sys.argv[1:]] or [64, 512]:
    prs = np.tile(M.BASEPARAMS, (n_part, 1))
    prs[:, :4] = (lo[pos] + (hi[pos] - lo[pos]) * rs.uniform(0, 1, (n_part, 5)))[:, :4]
    p0 = np.concatenate([M.p0_rows(cond, pr) for pr in prs])
    y0 = np.array([guess[i] for pr in prs for i in range(30)])
    t0 = time.perf_counter()
    flows, status, _, info = pkg.methanation.dae_solve_batch(p0, y0)
    dt = time.perf_counter() - t0
    n = len(p0)
    print(f"{n_part} particles x 30 = {n} solves: kernel {info['kernel_ms']:.1f} ms ({n/info['kernel_ms']*1e3:.0f} solves/s), wall {dt:.2f} s, "
          f"failed {int((status!=0).sum())}, steps/solve {info['steps']/n:.0f}, newton its/solve {info['newton_iters']/n:.0f}, factorisations/solve {info.get('factorisations', 0)/n:.0f}, "
          f"rejects/solve {info['rejects']/n:.1f}, newton fails/solve {info['newton_fail']/n:.2f}", flush=True)
