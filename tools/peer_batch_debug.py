"""Debug aid: one rank against W rank threads (collectives inside the engine / host-driven); prints per-iteration accept counts."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import __graft_entry__ as g
import test_gpu_peer_collectives as T
pkg = g.load_package()
O = g.load_oracle()
data = O.MMData.load()
n, seed = 6144 * 3, 77


def maker(world):
    def make(r):
        e = pkg.HipEngine(n // world, 3, device=0, n_global=n)
        e.set_model_mm(data.t, data.P_obs, data.S0)
        e.set_prior(pkg.SMCSettings().priors)
        return e
    return make


runs = {}
for name, world, peer, mb in (("W1 fused", 1, True, 0), ("W1 batched", 1, True, "auto"), ("W2 peer", 2, True, 0), ("W2 host", 2, False, 0), ("W3 peer", 3, True, 0)):
    runs[name] = T._run_ranks(maker(world), pkg.SMCSettings(n_particle=n, mh_batch=mb), world, seed, pkg.run_smc, peer=peer)[0]
names = list(runs)
print(names)
for k in range(len(runs[names[0]]["records"])):
    recs = [runs[nm]["records"][k] if k < len(runs[nm]["records"]) else None for nm in names]
    print(f"step {k + 1}: gamma " + " ".join(f"{r['gamma_new']:.8g}" if r else "-" for r in recs) + " | max_lk " + " ".join(repr(r["max_lk"]) if r else "-" for r in recs)
          + " | offspring " + " ".join(str(r["n_offspring"]) if r else "-" for r in recs))
    for j in range(max(len(r["mh"]) for r in recs if r)):
        print("     it %d: " % j + " | ".join((f"{r['mh'][j]['accepted_now']}/{r['mh'][j]['accepted_ever']} r={r['mh'][j]['mhstep_ratio']} c00={r['mh'][j]['cov_m'][0, 0]:.17g}") if r and j < len(r["mh"]) else "-" for r in recs))
