import cProfile, pstats, os, sys, time, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
z = np.load(os.path.join(g.ROOT, "tests", "golden", "mm_data.npz"))
n = 1_000_000
s = pkg.SMCSettings(n_particle=n)
eng = pkg.HipEngine(n, 3, device=0)
eng.set_model_mm(z["t"], z["P_obs"], z["S0"]); eng.set_prior(s.priors)
for i in range(3): pkg.run_smc(eng, s, rng="device", verbose=False, seed_device=900 + i)
eng.timing_enable(True); eng.timing_reset(); eng.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for i in range(10): pkg.run_smc(eng, s, rng="device", verbose=False, seed_device=1000 + i)
eng.synchronize()
pr.disable()
dt = time.perf_counter() - t0
tm = eng.timing_get()
ker = sum(v["ms"] for k, v in tm.items() if k not in ("solve", "mh", "loglik")) + tm["mh"]["ms"] + tm["loglik"]["ms"]
print(f"10 runs: {dt*100:.2f} ms per run; kernels (events) {ker/10:.2f} ms per run; difference {dt*100 - ker/10:.2f} ms")
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(22); print(st.getvalue()[:5000])
