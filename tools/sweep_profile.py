"""Where one complete 10^6-particle run spends its time, sweep by sweep: wall time and device-counted RK45 attempts of the
initial likelihood sweep and of every fused Metropolis iteration, and the time between them (ESS search, resampling).
python tools/sweep_profile.py [n=1000000] [seed=1000] [stiff_first=1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
stiff = int(sys.argv[3]) if len(sys.argv) > 3 else 1
z = np.load(os.path.join(g.ROOT, "tests", "golden", "mm_data.npz"))
s = pkg.SMCSettings(n_particle=n, stiff_first=bool(stiff))
with pkg.HipEngine(n, 3) as eng:
    eng.set_model_mm(z["t"], z["P_obs"], z["S0"])
    eng.set_prior(s.priors)
    pkg.run_smc(eng, s, rng="device", verbose=False, seed_device=seed - 1)       # warm-up
    log = []
    t_last = [time.perf_counter()]

    def wrap(name):
        f = getattr(eng, name)

        def w(*a, **k):
            t0 = time.perf_counter()
            gap = t0 - t_last[0]
            out = f(*a, **k)
            t1 = time.perf_counter()
            t_last[0] = t1
            att = out.get("rk_attempts", 0) if isinstance(out, dict) else 0
            acc = out.get("accepted_now", 0) if isinstance(out, dict) else 0
            log.append((name, 1e3 * (t1 - t0), att, acc, 1e3 * gap, a[0] if name == "mh_iteration_device_rng" else 0.0))
            return out
        setattr(eng, name, w)
    for nm in ("loglik", "mh_iteration_device_rng"):
        wrap(nm)
    eng.timing_enable(True); eng.timing_reset()
    t0 = time.perf_counter()
    out = pkg.run_smc(eng, s, rng="device", verbose=False, seed_device=seed)
    tot = 1e3 * (time.perf_counter() - t0)
    tm = eng.timing_get()
print(f"n = {n}, seed {seed}, stiff_first {stiff}: run {tot:.2f} ms, {len(out['records'])} tempering steps, {len(log) - 1} Metropolis sweeps")
print(f"{'call':>6} {'gamma':>10} {'wall ms':>8} {'gap before ms':>14} {'attempts':>12} {'att/particle':>12} {'accepted':>9}")
for name, ms, att, acc, gap, gam in log:
    print(f"{'loglik' if name == 'loglik' else 'mh':>6} {gam:10.5f} {ms:8.3f} {gap:14.3f} {att:12d} {att / n:12.1f} {acc:9d}")
sw = sum(l[1] for l in log[1:])
print(f"sum: loglik {log[0][1]:.2f} ms, Metropolis sweeps {sw:.2f} ms, everything between them {sum(l[4] for l in log[1:]):.2f} ms")
print("kernel time by class (HIP events):", {k: round(v["ms"], 2) for k, v in tm.items()})
