"""The reference's SMC_methanation/SMC_methanation_main.py on the GPU: same settings names and defaults, same per-step log line.

    python examples/methanation_run.py [n_particle=256] [numpy|device]

The reference's inlet table (methanation_data/information.csv) is missing upstream, so this runs on the synthetic table committed
under tests/golden/ - 30 experiments in physically plausible ranges - with observations generated as the reference's driver
generates them (:89-101): the model at baseparams plus sigma = 5 noise from NumPy's stream after seed(20250205).
The DAE time integration (my_model -> IDA in the reference) is the HIP kernel K8: parity-unpinned."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import __graft_entry__ as g

pkg = g.load_package()
M = pkg.methanation
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = sys.argv[2] if len(sys.argv) > 2 else "device"
cond = M.load_conditions(os.path.join(g.ROOT, "tests", "golden", "methanation_information.csv"))
guess = M.initial_guess(cond)                                        # SMC_methanation_main.py:47-58
lo, hi, pos = M.prior_box()                                          # methanation_set_conditon.py:59-70
priors = {nm: {"dist": "uniform", "low": float(lo[i]), "high": float(hi[i])} for nm, i in zip(["Af", "Eaf", "Ar", "Ear", "sigma"], pos)}
flows0, status, _, _ = M.dae_solve_batch(M.p0_rows(cond, M.BASEPARAMS), guess)      # the model at baseparams: 30 DAE solves on the GPU
assert (status == 0).all()
np.random.seed(20250205)
obs = flows0.T + 5.0 * np.random.standard_normal((5, 30))            # :94-95
s = pkg.SMCSettings(n_particle=n, priors=priors)
with pkg.HipEngine(n, 5, device=0) as eng:
    eng.set_model_methanation(cond, guess, obs, np.append(M.BASEPARAMS, M.SIGMA_TRUE), pos)
    eng.set_prior(priors)
    out = pkg.run_smc(eng, s, rng=rng, verbose=True)
st = out["stats"]
print("posterior mean", out["p_pred"].mean(axis=0), "\nposterior std ", out["p_pred"].std(axis=0))
print("generated with", np.append(M.BASEPARAMS, M.SIGMA_TRUE)[pos], "log-evidence", out["logZ"])
print(f"{st['dae_solves']} DAE solves (+ {st['dae_solves_cancelled']} skipped by the exact early rejection): per solve "
      f"{st['bdf_steps'] / st['dae_solves']:.0f} BDF steps, {st['newton_iters'] / st['dae_solves']:.0f} Newton iterations, "
      f"{st['factorisations'] / st['dae_solves']:.0f} factorisations")
