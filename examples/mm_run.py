"""The reference's SMC_example/Micmem_SMC_main.py on the GPU: same settings names and defaults, same per-step log line.

    python examples/mm_run.py [n_particle] [numpy|device]

"numpy" draws every random number from NumPy's global generator in the reference's order (identical results on the
reference's seed); "device" keeps the draws on the GPU (Philox)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import __graft_entry__ as g

pkg = g.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rng = sys.argv[2] if len(sys.argv) > 2 else "numpy"
frames = pkg.datagen.make_pseudo_data(write=False)           # the data sets of Micmen_generate_data.py (generated on the GPU)
t = np.array([f["t"].values for f in frames])
P_obs = np.array([f["P_obs"].values for f in frames])
S0 = np.array([f["S_true"].iloc[0] for f in frames])
s = pkg.SMCSettings(n_particle=n)
with pkg.HipEngine(n, 3, device=0) as eng:
    eng.set_model_mm(t, P_obs, S0)
    eng.set_prior(s.priors)
    out = pkg.run_smc(eng, s, rng=rng, verbose=True, dump_dir=os.environ.get("SMC_DUMP_DIR"))
print("posterior mean", out["p_pred"].mean(axis=0), "std", out["p_pred"].std(axis=0), "log-evidence", out["logZ"])
