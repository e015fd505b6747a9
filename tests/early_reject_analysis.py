#!/usr/bin/env python3
"""Why exact early rejection (include/smc_hip.h: smc_set_early_reject) removes the long solves of the Metropolis sweeps:
CPU replay of a complete tempering run (the checker's statements of Micmem_SMC_main.py:95-262, N particles) that records,
for every proposal whose longest solve needs more than 5000 RK45 attempts, whether the accept test
    exp((lk2 - lk1) * gamma) >= rr        (:231-236)
is already decided - against the proposal - by the OTHER experiments of the same particle (those that finish within 2000
attempts, i.e. within a millisecond) with the long solve counted as a perfect fit (sum of squared residuals 0), and how
often such proposals are accepted at all.  Test infrastructure (uses oracle/); not part of the product.

    python tests/early_reject_analysis.py [N=1000000] > profiles/r02_early_reject_analysis.log
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402


def main():
    O = g.load_oracle()
    data = O.MMData.load()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    t0 = time.time()
    out = O.run_smc(data, O.SMCSettings(n_particle=n), seed=11, record_mh=True, n_threads=os.cpu_count() or 1)
    print(f"CPU replay of one run, N = {n}: {time.time() - t0:.0f} s, {out['step']} tempering steps, "
          f"{out['n_mutation_sweeps']} Metropolis sweeps")
    rows, lk_after = [], None
    for rec in out["records"]:
        gamma = rec.gamma_new
        lk_prev = out["sweeps"][0][1] if rec.step == 1 else lk_after
        anc = np.repeat(np.arange(n), rec.p_is)[:n]
        lk1 = np.empty(n)
        lk1[:len(anc)] = lk_prev[anc]
        for mh in rec.mh:
            prop, rr, lk2, p0, r = mh["proposals"], mh["rr"], mh["lk2"], mh["p0"], mh["r"]
            ratio = prop[:, 0] / np.maximum(prop[:, 1], 1e-300)
            sus = np.nonzero((ratio > 1500) & (p0 == 1))[0]      # every solve above 1e4 attempts has Vmax/Km > 2666
            if len(sus):
                sub = prop[sus]
                _, pred, _ = O.mm_loglik_batch(sub, data, want_pred=True)
                s2 = sub[:, 2] ** 2
                c0 = -0.5 * data.n_t * np.log(2 * np.pi * s2)
                with np.errstate(divide="ignore"):
                    lkmin = lk1[sus] + np.log(rr[sus]) / gamma
                full = ((data.P_obs[None] - pred) ** 2).sum(axis=2)
                for k, i in enumerate(sus):
                    att = np.array([O.rk45_solve(prop[i, 0], prop[i, 1], data.S0[e], data.t[e])[1]["n_attempts"]
                                    for e in range(data.n_ex)])
                    e_long = int(att.argmax())
                    if att[e_long] < 5000:
                        continue
                    bound = c0[k] + sum((c0[k] - full[k, e] / (2 * s2[k])) if att[e] < 2000 else c0[k]
                                        for e in range(data.n_ex) if e != e_long)
                    rows.append((rec.step, gamma, att[e_long], bool(bound < lkmin[k]), int(r[i])))
            lk1 = lk2 * r + lk1 * (1.0 - r)
        lk_after = lk1
    rows = np.array(rows, dtype=float)
    print(f"proposals whose longest solve exceeds 5000 attempts: {len(rows)}; accepted: {int(rows[:, 4].sum())}")
    print(f"rejection already certain from the quickly finished experiments alone: {int(rows[:, 3].sum())} of {len(rows)} "
          f"({100 * rows[:, 3].mean():.1f} %), attempt-weighted {100 * (rows[:, 2] * rows[:, 3]).sum() / rows[:, 2].sum():.1f} %")
    print("step  gamma    long proposals  certain  accepted  longest solve (attempts)")
    for st in sorted(set(rows[:, 0])):
        m = rows[:, 0] == st
        print(f"{int(st):4d}  {rows[m, 1][0]:.4f}  {int(m.sum()):14d}  {int(rows[m, 3].sum()):7d}  {int(rows[m, 4].sum()):8d}  "
              f"{int(rows[m, 2].max()):8d}")


if __name__ == "__main__":
    main()
