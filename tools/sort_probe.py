"""How much of a mid-run sweep's time is lane imbalance between unlike items?  Takes the proposals of one Metropolis sweep of
a 10^6-particle run (sweep k, default 12: gamma ~ 0.02) and times plain likelihood sweeps over them in the run's order, sorted by
Vmax / Km, sorted by the solves' own attempt counts (the best a predictor could do), and shuffled.
python tools/sort_probe.py [sweep=12] [n=1000000]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
k_want = int(sys.argv[1]) if len(sys.argv) > 1 else 12
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
    grabbed = {}

    def w(gamma, *a, **kw):
        out = f(gamma, *a, **kw)
        if k[0] == k_want:
            grabbed["prop"] = eng.download_particles(pkg.SMC_SET_PRED)
            grabbed["gamma"] = gamma
            raise Stop()
        k[0] += 1
        return out
    eng.mh_iteration_device_rng = w
    try:
        pkg.run_smc(eng, s, rng="device", verbose=False, seed_device=1000)
    except Stop:
        pass
    eng.mh_iteration_device_rng = f
    prop = grabbed["prop"]
    ok = (prop > 0).all(axis=1) & (prop < 10).all(axis=1)
    prop[~ok] = prop[ok][0]                                  # proposals outside the prior box: replaced (a sweep would mask them)
    eng.upload_particles(pkg.SMC_SET_PRED, prop)
    eng.loglik(pkg.SMC_SET_PRED)
    att = (eng.download_item_info() & 0x1fffffff).sum(axis=0)
    # the long chains bound a plain likelihood sweep (no early rejection here) and hide the bulk: the top 0.5 % are replaced
    cut = np.percentile(att, 99.5)
    typical = prop[att <= np.median(att)]
    big = att > cut
    prop[big] = typical[np.random.RandomState(1).randint(len(typical), size=int(big.sum()))]
    eng.upload_particles(pkg.SMC_SET_PRED, prop)
    eng.loglik(pkg.SMC_SET_PRED)
    att = (eng.download_item_info() & 0x1fffffff).sum(axis=0)
    rs = np.random.RandomState(0)
    orders = {"run order": np.arange(n), "sorted by Vmax / Km": np.argsort(prop[:, 0] / prop[:, 1]),
              "sorted by attempts": np.argsort(att, kind="stable"), "shuffled": rs.permutation(n)}
    print(f"proposals of sweep {k_want} (gamma {grabbed['gamma']:.5f}): {att.sum() / 1e6:.1f} M attempts, per particle "
          f"min {att.min()} median {int(np.median(att))} 99 % {int(np.percentile(att, 99))} max {att.max()}")
    only = sys.argv[3] if len(sys.argv) > 3 else None         # e.g. "run order": for a counter pass over one ordering
    for name, o in orders.items():
        if only and name != only:
            continue
        eng.debug_set_order(None)
        eng.upload_particles(pkg.SMC_SET_PRED, prop[o])
        eng.loglik(pkg.SMC_SET_PRED)
        eng.timing_enable(True); eng.timing_reset()
        for _ in range(5):
            eng.loglik(pkg.SMC_SET_PRED)
        tm = eng.timing_get()
        print(f"  {name:22s}: solve kernel {tm['solve']['ms'] / tm['solve']['launches']:.3f} ms per sweep", flush=True)
    # the same through the order indirection of the solve kernel (particles stay where they are; smc_debug_set_order), with patience
    eng.upload_particles(pkg.SMC_SET_PRED, prop)
    for name in ("sorted by Vmax / Km",):
        for patience in (0, 12):
            eng.debug_set_order(orders[name], patience)
            eng.loglik(pkg.SMC_SET_PRED)
            eng.timing_enable(True); eng.timing_reset()
            for _ in range(5):
                eng.loglik(pkg.SMC_SET_PRED)
            tm = eng.timing_get()
            print(f"  order[] = {name}, patience {patience:2d}: solve kernel {tm['solve']['ms'] / tm['solve']['launches']:.3f} ms per sweep", flush=True)
    eng.debug_set_order(None)
