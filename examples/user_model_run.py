"""A model of your own instead of a new Micmem_likelihood.py: consecutive reactions A -> B -> C, B observed.

    python examples/user_model_run.py [n_particle]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import __graft_entry__ as g

pkg = g.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
rs = np.random.RandomState(0)
t = np.tile(np.linspace(0.0, 10.0, 30), (4, 1))
A0 = np.array([1.0, 2.0, 0.5, 1.5])
k1, k2, sigma = 0.8, 0.3, 0.01
B = A0[:, None] * k1 / (k2 - k1) * (np.exp(-k1 * t) - np.exp(-k2 * t))      # closed form of the intermediate
obs = B + sigma * rs.standard_normal(B.shape)
priors = {"k1": {"dist": "uniform", "low": 0, "high": 3}, "k2": {"dist": "uniform", "low": 0, "high": 3},
          "sigma": {"dist": "uniform", "low": 0, "high": 1}}
with pkg.HipEngine(n, 3, device=0) as eng:
    eng.set_prior(priors)
    eng.set_model_user(pkg.user_models.CONSECUTIVE_REACTIONS, n_states=2, t=t, obs=obs, cond=A0[:, None])
    out = pkg.run_smc(eng, pkg.SMCSettings(n_particle=n, priors=priors), rng="device", verbose=True)
print("posterior mean", out["p_pred"].mean(axis=0), "(generated with", (k1, k2, sigma), ")")
