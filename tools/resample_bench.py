import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
d = np.load(os.path.join(g.ROOT, "tests", "golden", "mm_data.npz"))
n = 1_000_000
rs = np.random.RandomState(0)
th = np.array([1.2254, 0.5218, 0.02048]) + rs.standard_normal((n, 3)) * np.array([0.05, 0.05, 0.002])
for scheme in ("residual_systematic", "systematic", "multinomial"):
    s = pkg.SMCSettings(n_particle=n, resampling=scheme)
    with pkg.HipEngine(n, 3, device=0) as eng:
        eng.set_model_mm(d["t"], d["P_obs"], d["S0"]); eng.set_prior(s.priors); eng.set_resampling(scheme)
        eng.upload_particles(pkg.SMC_SET_PRED, th); eng.loglik(pkg.SMC_SET_PRED)
        es = pkg.ess_search(eng, pkg.SingleComm(), 0.0, s)
        eng.synchronize(); t0 = time.perf_counter()
        for k in range(5):
            out = pkg.resample(eng, pkg.SingleComm(), es, 0.3 + 0.1 * k, s, k == 0)
        eng.synchronize(); dt = (time.perf_counter() - t0) / 5
        off = eng.download_offspring()
        print(f"{scheme:20s}: {dt*1e3:.2f} ms per resampling of {n} particles, offspring sum {int(off.sum())}, max {int(off.max())}, zero-offspring share {np.mean(off==0):.3f}")
