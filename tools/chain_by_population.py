"""How long is the longest RK45 chain of the initial sweep as the prior population grows?  (what N ranks can deliver.)
For a prior of n particles drawn exactly as bench.py's run i draws it (Philox keyed by the GLOBAL particle index, so the prior of
8 ranks x 10^6 IS this prior of 8 x 10^6): wall time of the likelihood sweep on one GPU, attempts of the longest solve, and the
floor that solve alone sets (0.325 us per attempt in the hand-written lone-chain loop)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
z = np.load(os.path.join(g.ROOT, "tests", "golden", "mm_data.npz"))
rows = []
for n in [int(float(a)) for a in sys.argv[1:]] or [1_000_000, 2_000_000, 4_000_000, 8_000_000]:
    with pkg.HipEngine(n, 3, device=0) as eng:
        eng.set_model_mm(z["t"], z["P_obs"], z["S0"])
        eng.set_prior(pkg.SMCSettings().priors)
        for seed in (1000, 1001, 1002):
            eng.sample_prior_device(seed, 0)
            eng.synchronize()
            t0 = time.perf_counter()
            info = eng.loglik(pkg.SMC_SET_PRED)
            eng.synchronize()
            dt = time.perf_counter() - t0
            att = eng.download_item_info() & 0x1FFFFFFF
            rows.append({"particles": n, "seed": seed, "sweep_ms": 1e3 * dt, "max_attempts": int(att.max()), "chain_floor_ms": float(att.max()) * 0.325e-3,
                         "attempts_total": int(info["rk_attempts"]), "items_over_10000_attempts": int((att > 10000).sum())})
            print(json.dumps(rows[-1]), flush=True)
