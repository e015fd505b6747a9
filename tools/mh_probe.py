"""One Metropolis sweep of a 10^6-particle run, replayed: the population, its likelihoods and the arguments of fused iteration
k are grabbed from a run, and the identical iteration (same Philox keys: same proposals, same uniforms) is timed again and again
with the scheduling switches in different positions.   python tools/mh_probe.py [sweep=8] [n=1000000]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
k_want = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
z = np.load(os.path.join(g.ROOT, "tests", "golden", "mm_data.npz"))
s = pkg.SMCSettings(n_particle=n)


class Stop(Exception):
    pass


with pkg.HipEngine(n, 3) as eng:
    eng.set_model_mm(z["t"], z["P_obs"], z["S0"])
    eng.set_prior(s.priors)
    f = eng.mh_iteration_device_rng
    k = [0]
    grab = {}

    def w(*a, **kw):
        if k[0] == k_want:
            grab["filt"] = eng.download_particles(pkg.SMC_SET_FILT)
            grab["lk"] = eng.download_lk(pkg.SMC_SET_FILT)
            grab["args"] = (a, kw)
            raise Stop()
        k[0] += 1
        return f(*a, **kw)
    eng.mh_iteration_device_rng = w
    try:
        pkg.run_smc(eng, s, rng="device", verbose=False, seed_device=1000)
    except Stop:
        pass
    eng.mh_iteration_device_rng = f
    a, kw = grab["args"]
    print(f"sweep {k_want}: gamma {a[0]:.5f}")

    def replay(label, reps=4, **sw):
        eng.set_cost_order(sw.get("cost", True)); eng.set_in_phase(sw.get("phase", True))
        eng.set_early_reject(sw.get("reject", True)); eng.set_stiff_first(sw.get("stiff", True))
        ms, acc, att = [], None, None
        for r in range(reps + 1):
            eng.upload_particles(pkg.SMC_SET_FILT, grab["filt"])
            eng.upload_lk(pkg.SMC_SET_FILT, grab["lk"])
            eng.timing_enable(True); eng.timing_reset()
            out = f(*a, **kw)
            tm = eng.timing_get()
            if r:
                ms.append(tm["solve"]["ms"])
            acc, att = out["accepted_now"], out["rk_attempts"]
        print(f"  {label:44s}: solve kernel {np.mean(ms):.3f} ms (min {min(ms):.3f}), accepted {acc}, attempts {att / 1e6:.1f} M", flush=True)
    only = sys.argv[3] if len(sys.argv) > 3 else None          # "default" / "index": one variant, for a counter pass
    if only == "default":
        replay("cost order + in phase (default)", reps=6)
        sys.exit(0)
    if only == "index":
        replay("index order, no patience", reps=6, cost=False, phase=False)
        sys.exit(0)
    replay("index order, no patience", cost=False, phase=False)
    replay("cost order + in phase (default)")
    replay("cost order + in phase, no early rejection", reject=False)
    replay("index order, no early rejection", cost=False, phase=False, reject=False)
    replay("cost order + in phase, no stiff list", stiff=False)
    replay("index order, no stiff list", cost=False, phase=False, stiff=False)
