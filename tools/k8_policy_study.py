"""CPU study of K8's control policy: steps, factorisations and Newton iterations per solve, and the distance
of the outlet state from a tight-tolerance run, for the policies oracle/meth_dae_oracle.c knows (dae_policy).  Test tooling."""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

entry.load_oracle()
from oracle import methanation as OM  # noqa: E402


Policy = OM.DaePolicy


def solve(y0, p, pol, rtol=1e-6):
    out, rc, st = OM.dae_solve_policy(y0, p, pol, rtol=rtol, atol=rtol)
    return out, rc, st["steps"], st["rejects"], st["nlu"], st["newton_iters"], st["newton_fail"], st["stale_retries"]


def main():
    n_post, n_prior = int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 4
    cond = OM.load_conditions(os.path.join(ROOT, "tests", "golden", "methanation_information.csv"))
    guess = OM.initial_guess(cond)
    lo, hi, pos = OM.prior_box()
    rs = np.random.RandomState(5)
    items = []
    for k in range(n_post):
        pr = OM.BASEPARAMS * (1.0 + 0.02 * rs.standard_normal(8))
        items += [("post", np.ascontiguousarray(guess[i]), OM.p0_tuple(cond, i, pr)) for i in range(30)]
    for k in range(n_prior):
        pr = OM.BASEPARAMS.copy()
        for q in pos[:4]:
            pr[q] = rs.uniform(lo[q], hi[q])
        items += [("prior", np.ascontiguousarray(guess[i]), OM.p0_tuple(cond, i, pr)) for i in range(30)]
    outlet = [50, 101, 152, 203, 254, 305, 356]
    cores = len(os.sched_getaffinity(0))
    with ThreadPoolExecutor(cores) as ex:
        ref = list(ex.map(lambda it: solve(it[1], it[2], None, rtol=1e-9), items))
        policies = {
            "checker (every attempt, SciPy)": Policy(0, 0, 0, 0, 0.33, 0.25),
            "r4 K8: reuse c==c_lu, SciPy newton/step": Policy(1, 0, 0, 0, 0.33, 0.25),
            "IDA newton only (1,1,0)": Policy(1, 1, 0, 0, 0.33, 0.25),
            "(1,1,0) epcon 0.1": Policy(1, 1, 0, 0, 0.1, 0.25),
            "IDA reuse+newton (2,1,0) xrate .25": Policy(2, 1, 0, 0, 0.33, 0.25),
            "(2,1,0) xrate .15": Policy(2, 1, 0, 0, 0.33, 0.15),
            "(2,1,0) xrate .10": Policy(2, 1, 0, 0, 0.33, 0.10),
            "(2,1,0) xrate .05": Policy(2, 1, 0, 0, 0.33, 0.05),
            "(2,1,0) xrate .02": Policy(2, 1, 0, 0, 0.33, 0.02),
            "(2,1,0) xrate .25 epcon 0.1": Policy(2, 1, 0, 0, 0.1, 0.25),
            "(2,1,0) xrate .10 epcon 0.1": Policy(2, 1, 0, 0, 0.1, 0.10),
            "IDA all (2,1,1)": Policy(2, 1, 1, 0, 0.33, 0.25),
        }
        print(f"{len(items)} solves ({n_post} posterior-like + {n_prior} prior-box particles x 30 experiments), {cores} threads")
        print(f"{'policy':54s} {'steps':>7s} {'rej':>5s} {'nlu':>6s} {'newton':>7s} {'nfail':>5s} {'stale':>5s} {'fail':>4s} | units vs 1e-9 run: median  p95  max | cost")
        def solve_ida(it):
            out, rc, st = OM.dae_solve_ida(it[1], it[2])
            return out, rc, st["steps"], st["error_test_failures"], st["nlu"], st["newton_iters"], st["newton_failures"], 0
        policies["IDA's own algorithm (divided differences, restated)"] = "ida"
        for name, pol in policies.items():
            t0 = time.time()
            res = list(ex.map(solve_ida if pol == "ida" else (lambda it: solve(it[1], it[2], pol)), items))
            ok = [k for k in range(len(items)) if res[k][1] == 0 and ref[k][1] == 0]
            a = np.array([[r[2], r[3], r[4], r[5], r[6], r[7]] for r in res], dtype=float)
            units = np.array([np.max(np.abs(res[k][0] - ref[k][0])[outlet] / (1e-6 + 1e-6 * np.abs(ref[k][0][outlet]))) for k in ok])
            m = a[ok].mean(axis=0)
            cost = m[2] * 60.1e3 + m[3] * 15.5e3 + (m[0] + m[1] + m[4] + m[5]) * 12e3     # cycles per solve by K8's r4 per-phase costs: factorisation, Newton iteration, rest of an attempt
            print(f"{name[:54]:54s} {m[0]:7.1f} {m[1]:5.1f} {m[2]:6.1f} {m[3]:7.1f} {m[4]:5.2f} {m[5]:5.1f} {sum(1 for r in res if r[1] != 0):4d} | "
                  f"{np.median(units):8.3f} {np.percentile(units, 95):8.2f} {units.max():8.2f} | {cost / 1e6:6.2f} M  ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
