"""Single-chain latency probe: time one stiff (particle, experiment) solve that needs ~1e5 sequential
RK45 attempts, alone on the GPU, and a bulk of posterior-like particles for the issue-bound rate."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
z = np.load(os.path.join(g.ROOT, "tests", "golden", "mm_data.npz"))
rs = np.random.RandomState(0)
th = rs.uniform(0, 10, size=(1_000_000, 3))
ratio = th[:, 0] / th[:, 1]
idx = np.argsort(-ratio)[:200]
with pkg.HipEngine(1_000_000, 3) as eng:
    eng.set_model_mm(z["t"], z["P_obs"], z["S0"])
    eng.set_prior(pkg.SMCSettings().priors)
    # heavy candidates one at a time
    best = None
    for i in idx[:60]:
        lk, _, info = eng.loglik_host(th[i:i + 1])
        if best is None or info["rk_attempts"] > best[1]:
            best = (i, info["rk_attempts"])
    i, att = best
    eng.loglik_host(th[i:i + 1])
    t0 = time.perf_counter()
    for _ in range(3):
        lk, _, info = eng.loglik_host(th[i:i + 1])
    dt = (time.perf_counter() - t0) / 3
    print(f"heaviest of 60: particle {i} theta={th[i]} attempts(6 expts)={att} time={dt*1e3:.2f} ms "
          f"-> {dt/ (att) * 1e6 * 1.0:.3f} us per attempt if one expt dominates")
    # bulk posterior-like throughput
    n = 1_000_000
    post = np.array([1.2254, 0.5218, 0.02048]) + rs.standard_normal((n, 3)) * np.array([0.025, 0.0295, 0.00094])
    eng.upload_particles(pkg.SMC_SET_PRED, post)
    eng.loglik(pkg.SMC_SET_PRED)
    eng.timing_enable(True); eng.timing_reset()
    for _ in range(5):
        info = eng.loglik(pkg.SMC_SET_PRED)
    tm = eng.timing_get()
    ms = tm["solve"]["ms"] / tm["solve"]["launches"]
    print(f"posterior-like 1e6: solve {ms:.3f} ms/launch, attempts {info['rk_attempts']}, "
          f"{info['rk_attempts']/ (ms*1e-3) / 1e9:.2f} G attempts/s, {n/(ms*1e-3)/1e6:.1f} M particle-evals/s")
    # prior 1e6
    eng.upload_particles(pkg.SMC_SET_PRED, th)
    eng.timing_reset()
    info = eng.loglik(pkg.SMC_SET_PRED)
    tm = eng.timing_get()
    print(f"prior 1e6: solve {tm['solve']['ms']:.2f} ms, attempts {info['rk_attempts']}, failed {info['n_failed']}")
