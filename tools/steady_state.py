"""Steady-state Metropolis sweeps (posterior-like population, gamma = 1) through the fused iteration: solve-kernel and wall
time per sweep.  python tools/steady_state.py [n=1000000] [early_reject=1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rej = int(sys.argv[2]) if len(sys.argv) > 2 else 1
z = np.load(os.path.join(g.ROOT, "tests", "golden", "mm_data.npz"))
s = pkg.SMCSettings(n_particle=n)
rs = np.random.RandomState(0)
with pkg.HipEngine(n, 3) as eng:
    eng.set_model_mm(z["t"], z["P_obs"], z["S0"])
    eng.set_prior(s.priors)
    eng.set_early_reject(rej)
    th = np.array([1.2254, 0.5218, 0.02048]) + rs.standard_normal((n, 3)) * np.array([0.025, 0.0295, 0.00094])
    eng.upload_particles(pkg.SMC_SET_PRED, th)
    eng.loglik(pkg.SMC_SET_PRED)
    eng.upload_particles(pkg.SMC_SET_FILT, th)
    eng.upload_lk(pkg.SMC_SET_FILT, eng.download_lk(pkg.SMC_SET_PRED))
    w = s.w_cov()
    for j in range(3):
        eng.mh_iteration_device_rng(1.0, 1.0, w, 1, j, 0)
    eng.timing_enable(True); eng.timing_reset()
    t0 = time.perf_counter()
    k = 10
    for j in range(k):
        eng.mh_iteration_device_rng(1.0, 1.0, w, 2, j, 0)
    dt = time.perf_counter() - t0
    tm = eng.timing_get()
    print(f"n = {n}, early_reject = {rej}: solve kernel {tm['solve']['ms'] / k:.3f} ms, mh (propose+solve+accept) {tm['mh']['ms'] / k:.3f} ms, "
          f"wall {dt / k * 1e3:.3f} ms per sweep -> {n / (dt / k) / 1e6:.1f} M particle-mutation-steps/s")
