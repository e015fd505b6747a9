"""What bounds each Metropolis sweep of a 10^6-particle run: after every fused iteration the per-item records of the sweep
(include/smc_hip.h: smc_download_item_info) and the proposals are read back, and the longest solves are listed with the
Vmax / Km of their proposal, whether early rejection cancelled them and whether they were accepted.
python tools/sweep_tail_census.py [n=1000000] [seed=1000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
z = np.load(os.path.join(g.ROOT, "tests", "golden", "mm_data.npz"))
s = pkg.SMCSettings(n_particle=n)
with pkg.HipEngine(n, 3) as eng:
    eng.set_model_mm(z["t"], z["P_obs"], z["S0"])
    eng.set_prior(s.priors)
    eng.timing_enable(True)
    f = eng.mh_iteration_device_rng
    k = [0]

    def w(gamma, *a, **kw):
        eng.timing_reset()
        out = f(gamma, *a, **kw)
        tm = eng.timing_get()
        info = eng.download_item_info()
        prop = eng.download_particles(pkg.SMC_SET_PRED)
        att = info & 0x1fffffff
        canc = (info >> 29) & 1
        ratio = prop[:, 0] / prop[:, 1]
        flat = att.ravel()
        top = np.argsort(flat)[-6:][::-1]
        desc = ", ".join(f"{flat[i]}{'c' if canc.ravel()[i] else ''}@{ratio[i % n]:.0f}" for i in top)
        fin = att[canc == 0]
        filt = eng.download_particles(pkg.SMC_SET_FILT)
        accepted = (filt == prop).all(axis=1)                  # the selected particle is the proposal (masked ones count too)
        longf = (att > 256) & (canc == 0)                      # long solves that ran to their end
        lp = longf.any(axis=0)                                 # particles with such a solve
        extra = (f"; particles with a finished solve > 256 attempts: {int(lp.sum())}, of them accepted {int((lp & accepted).sum())}"
                 f" - attempts spent on the rejected ones {att[:, lp & ~accepted].sum() / 1e6:.2f} M")
        print(f"sweep {k[0]:2d} gamma {gamma:.5f}: solve kernel {tm['solve']['ms']:.3f} ms, attempts {att.sum() / 1e6:6.1f} M, "
              f"items > 64 attempts: {int((att > 64).sum())} of {int((att > 0).sum())} solved, > 256: {int((att > 256).sum())} ({int(((att > 256) & (canc == 1)).sum())} cancelled), > 1000: {int((att > 1000).sum())}, "
              f"longest finished {int(fin.max())}{extra}; top (attempts[c]@Vmax/Km): {desc}", flush=True)
        k[0] += 1
        return out
    eng.mh_iteration_device_rng = w
    pkg.run_smc(eng, s, rng="device", verbose=False, seed_device=seed)
