"""Rate of the host-buffer boundary (the drop-in sim_particle path, smc_mm_loglik_host): particles in, logL out over PCIe."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
d = np.load(os.path.join(g.ROOT, "tests", "golden", "mm_data.npz"))
n = 1_000_000
rs = np.random.RandomState(0)
th = np.array([1.2254, 0.5218, 0.02048]) + rs.standard_normal((n, 3)) * np.array([0.025, 0.0295, 0.00094])
with pkg.HipEngine(n, 3, device=0) as eng:
    eng.set_model_mm(d["t"], d["P_obs"], d["S0"])
    eng.loglik_host(th)
    t0 = time.perf_counter()
    for _ in range(5):
        lk = eng.loglik_host(th)
    dt = (time.perf_counter() - t0) / 5
    eng.upload_particles(pkg.SMC_SET_PRED, th)
    eng.loglik(pkg.SMC_SET_PRED); eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.loglik(pkg.SMC_SET_PRED)
    eng.synchronize()
    dr = (time.perf_counter() - t0) / 5
print(f"host buffers (PCIe in/out): {dt*1e3:.2f} ms per 1e6-particle likelihood sweep = {n/dt:.3g} particles/s; resident: {dr*1e3:.2f} ms = {n/dr:.3g} particles/s")
