"""Throughput of the user-model likelihood kernel (hiprtc, N1) against the built-in Michaelis-Menten kernel on the same particles."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
d = np.load(os.path.join(g.ROOT, "tests", "golden", "mm_data.npz"))
t, P_obs, S0 = d["t"], d["P_obs"], d["S0"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
rs = np.random.RandomState(0)
for label, th in (("posterior-like", np.array([1.2254, 0.5218, 0.02048]) + rs.standard_normal((n, 3)) * np.array([0.025, 0.0295, 0.00094])),
                  ("prior-like", rs.uniform(0.01, 10, size=(n, 3)))):
    with pkg.HipEngine(n, 3, device=0) as eng:
        eng.set_prior(pkg.SMCSettings().priors)
        eng.upload_particles(pkg.SMC_SET_PRED, th)
        for name in ("built-in", "user", "user, no cost hint"):
            if name == "built-in":
                eng.set_model_mm(t, P_obs, S0)
            else:
                src = pkg.user_models.MICHAELIS_MENTEN if name == "user" else pkg.user_models.MICHAELIS_MENTEN_PLAIN
                eng.set_model_user(src, 1, t, P_obs, cond=np.asarray(S0)[:, None])
            eng.upload_particles(pkg.SMC_SET_PRED, th)          # (a Metropolis sweep leaves its proposals there)
            eng.loglik(pkg.SMC_SET_PRED)
            eng.synchronize()
            t0 = time.perf_counter()
            info = eng.loglik(pkg.SMC_SET_PRED)
            eng.synchronize()
            dt = time.perf_counter() - t0
            # one Metropolis sweep (proposal, solves with exact early rejection, accept) from the same population: cost-ordered
            # and in phase for a model with a cost hint
            eng.upload_particles(pkg.SMC_SET_FILT, th)
            eng.upload_lk(pkg.SMC_SET_FILT, eng.download_lk(pkg.SMC_SET_PRED))
            tr = np.diag([0.3, 0.3, 0.05]) if label.startswith("prior") else np.diag([0.02, 0.02, 0.001])
            eng.mh_step_device_rng(0.02, 1.0, tr, 5, 1)
            eng.upload_particles(pkg.SMC_SET_FILT, th)
            eng.synchronize()
            t0 = time.perf_counter()
            eng.mh_step_device_rng(0.02, 1.0, tr, 5, 1)
            eng.synchronize()
            dt_mh = time.perf_counter() - t0
            print(f"{label:15s} {name:18s}: {dt*1e3:8.2f} ms per likelihood sweep of {n} particles ({n/dt:.3g} particles/s, {info['rk_attempts']/dt/1e9:.2f} G attempts/s); "
                  f"Metropolis sweep {dt_mh*1e3:.2f} ms", flush=True)
