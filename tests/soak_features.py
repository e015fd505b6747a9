"""Soak of the scheduling features (a test tool): complete device-RNG runs with everything that only reorders or re-routes work
switched ON (stiff / solo lists, in-phase waves, cost order, hand-written lone-chain loop; round 4: Metropolis loop control on the
device in batches of a random size, resampling enqueued without a synchronisation, page-locked result arrays) against the same
runs with all of it OFF (round 3's host loop) - tempering schedule, accept counts, Metropolis lengths, final particles, likelihoods and log-evidence must be identical
bit for bit, for a series of seeds and population sizes.   python tests/soak_features.py [n_cases=24] [device|numpy]
(numpy: the parity mode on the host's NumPy stream - EXACT kernels, host-drawn proposals)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
z = np.load(os.path.join(g.ROOT, "tests", "golden", "mm_data.npz"))
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
mode = sys.argv[2] if len(sys.argv) > 2 else "device"
rs = np.random.RandomState(2025)
t0 = time.time()
for case in range(n_cases):
    n = int(rs.choice([1000, 16384, 20000, 65536, 100000, 300000] if mode == "device" else [1000, 16384, 20000, 40000]))
    seed = int(rs.randint(1, 1 << 30))
    outs = {}
    batch = ["auto", 1, 2, 3, 5, 32][int(rs.randint(0, 6))]
    for on in (True, False):
        with pkg.HipEngine(n, 3) as eng:
            eng.set_model_mm(z["t"], z["P_obs"], z["S0"])
            s = pkg.SMCSettings(n_particle=n, stiff_first=on, in_phase=on, cost_order=on, seed=seed & 0x7fffffff,
                                mh_batch=batch if (on and mode == "device") else 0, defer_resample=on, pinned_results=on)
            eng.set_prior(s.priors)
            eng.set_fast_tail(on)
            outs[on] = pkg.run_smc(eng, s, rng=mode, verbose=False, seed_device=seed)
    a, b = outs[True], outs[False]
    same = ([r["gamma_new"] for r in a["records"]] == [r["gamma_new"] for r in b["records"]] and
            [r["n_accept"] for r in a["records"]] == [r["n_accept"] for r in b["records"]] and
            [r["last_j"] for r in a["records"]] == [r["last_j"] for r in b["records"]] and
            np.array_equal(a["p_pred"], b["p_pred"]) and np.array_equal(a["lk"], b["lk"]) and a["logZ"] == b["logZ"])
    print(f"case {case:2d}: n {n:7d} seed {seed:10d} mh_batch {str(batch):>4s}: {len(a['records'])} steps, {a['stats']['mutation_sweeps']} sweeps, logZ {a['logZ']:.6f}  "
          f"{'identical' if same else 'DIFFERENT'}  ({time.time() - t0:.0f} s)", flush=True)
    assert same, (case, n, seed)
print("feature soak ok")
