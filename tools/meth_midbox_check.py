"""K8 on the prior-box midpoint (Af, Eaf, Ar, Ear at 0.5 of their ranges): status and step counts, v3 / v2 / v1."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

def run(frac):
    import __graft_entry__ as g
    pkg = g.load_package()
    M = pkg.methanation
    cond = M.load_conditions(os.path.join(g.ROOT, "tests", "golden", "methanation_information.csv"))
    guess = M.initial_guess(cond)
    lo, hi, pos = M.prior_box()
    pr = M.BASEPARAMS.copy()
    pr[:4] = (lo[pos] + (hi[pos] - lo[pos]) * frac)[:4]
    flows, status, _, info = pkg.methanation.dae_solve_batch(M.p0_rows(cond, pr), guess[:30])
    return status, info

if __name__ == "__main__":
    frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
    if os.environ.get("SMC_CHILD"):
        st, info = run(frac)
        print(os.environ["SMC_CHILD"], "failed", int((st != 0).sum()), "of 30", {k: int(v) if k != "kernel_ms" else round(v, 1) for k, v in info.items()}, flush=True)
        sys.exit(0)
    for tag, env in (("v3", {}), ("v2", {"SMC_METH_DAE_V2": "1"}), ("v1", {"SMC_METH_DAE_V1": "1"})):
        subprocess.run([sys.executable, __file__, str(frac)], env=dict(os.environ, SMC_CHILD=tag, **env), check=True, timeout=300)
