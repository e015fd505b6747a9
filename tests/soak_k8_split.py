"""Soak of the two-wave K8 kernel (a test tool, not collected by pytest): complete device-RNG methanation runs with the two-wave
kernel (default) against the one-wave kernel (SMC_K8_SPLIT=0) for a series of seeds, population sizes and observation tables -
tempering schedule, accept counts, Metropolis lengths, final particles, likelihoods and log-evidence must be identical bit for
bit - and the two-wave kernel twice on the same input (a race between its waves would show as a difference between two runs).
    python tests/soak_k8_split.py [n_cases=16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
g.load_oracle()
from oracle import methanation as M

cond = M.load_conditions(os.path.join(g.ROOT, "tests", "golden", "methanation_information.csv"))
guess = M.initial_guess(cond)
lo, hi, pos = M.prior_box()
base = np.array([M.p0_tuple(cond, i, M.BASEPARAMS) for i in range(30)])
flows0 = pkg.methanation.dae_solve_batch(base, np.array([guess[i] for i in range(30)]))[0].T.copy()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rs = np.random.RandomState(404)
t0 = time.time()


def run(n, obs, seed, split):
    os.environ["SMC_K8_SPLIT"] = split
    names = ["Af", "Eaf", "Ar", "Ear", "sigma"]
    priors = {nm: {"dist": "uniform", "low": float(lo[i]), "high": float(hi[i])} for nm, i in zip(names, pos)}
    s, pos2 = pkg.SMCSettings(n_particle=n, priors=priors, seed=20250205), pos
    eng = pkg.HipEngine(n, 5, device=0)
    eng.set_model_methanation(cond, guess, obs, np.append(M.BASEPARAMS, M.SIGMA_TRUE), pos2)
    eng.set_prior(s.priors)
    with eng:
        return pkg.run_smc(eng, s, rng="device", verbose=False, seed_device=seed)


def same(a, b):
    return ([r["gamma_new"] for r in a["records"]] == [r["gamma_new"] for r in b["records"]] and
            [r["n_accept"] for r in a["records"]] == [r["n_accept"] for r in b["records"]] and
            [r["last_j"] for r in a["records"]] == [r["last_j"] for r in b["records"]] and
            np.array_equal(a["p_pred"], b["p_pred"]) and np.array_equal(a["lk"], b["lk"]) and a["logZ"] == b["logZ"])


for case in range(n_cases):
    n = int(rs.choice([64, 96, 192, 384, 1024]))
    seed = int(rs.randint(1, 1 << 30))
    obs = flows0 + float(rs.choice([2.0, 5.0, 10.0])) * rs.standard_normal(flows0.shape)
    a, a2, b = run(n, obs, seed, "1"), run(n, obs, seed, "1"), run(n, obs, seed, "0")
    ok = same(a, b) and same(a, a2)
    print(f"case {case:2d}: n {n:5d} seed {seed:10d}: {len(a['records'])} steps, {a['stats']['mutation_sweeps']} sweeps, "
          f"{a['stats']['dae_solves']} solves, logZ {a['logZ']:.6f}  {'identical' if ok else 'DIFFERENT'}  ({time.time() - t0:.0f} s)", flush=True)
    assert ok, (case, n, seed)
print("k8 two-wave soak ok")
