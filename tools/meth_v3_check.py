"""K8 versions against each other on the same solves: flows, status, step counts.   tools/meth_v3_check.py [n_particles=2] [a=v3] [b=v2]
(v2: lane = node scans, v3: element-layout scans in one wave, v4: two waves per solve - meth_dae_split.h)"""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

def run(n_part):
    import __graft_entry__ as g
    pkg = g.load_package()
    M = pkg.methanation
    cond = M.load_conditions(os.path.join(g.ROOT, "tests", "golden", "methanation_information.csv"))
    guess = M.initial_guess(cond)
    lo, hi, pos = M.prior_box()
    rs = np.random.RandomState(0)
    prs = np.tile(M.BASEPARAMS, (n_part, 1))
    prs[:, :4] = (lo[pos] + (hi[pos] - lo[pos]) * rs.uniform(0, 1, (n_part, 5)))[:, :4]
    p0 = np.concatenate([M.p0_rows(cond, pr) for pr in prs])
    y0 = np.array([guess[i] for pr in prs for i in range(30)])
    flows, status, _, info = pkg.methanation.dae_solve_batch(p0, y0)
    return flows, status, info

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    if os.environ.get("SMC_CHILD"):
        f, s, info = run(n)
        np.savez(os.environ["SMC_CHILD"], flows=f, status=s, info=json.dumps({k: float(v) for k, v in info.items()}))
        sys.exit(0)
    out = {}
    envs = {"v2": {"SMC_METH_DAE_V2": "1", "SMC_K8_SPLIT": "0"}, "v3": {"SMC_K8_SPLIT": "0"}, "v4": {"SMC_K8_SPLIT": "1"}}
    ta, tb = (sys.argv[2] if len(sys.argv) > 2 else "v3"), (sys.argv[3] if len(sys.argv) > 3 else "v2")
    for tag, env in ((ta, envs[ta]), (tb, envs[tb])):
        path = f"/tmp/meth_{tag}.npz"
        e = dict(os.environ, SMC_CHILD=path, **env)
        subprocess.run([sys.executable, __file__, str(n)], env=e, check=True, timeout=300)
        out[tag] = np.load(path)
        print(tag, out[tag]["info"], "failed", int((out[tag]["status"] != 0).sum()), flush=True)
    a, b = out[ta], out[tb]
    both = (a["status"] == 0) & (b["status"] == 0)
    rel = np.abs(a["flows"][both] - b["flows"][both]) / (1e-6 + 1e-6 * np.abs(b["flows"][both]))
    print(f"solves {len(a['status'])}: status equal {np.array_equal(a['status'], b['status'])}, "
          f"max |{ta}-{tb}| in tolerance units {rel.max():.2f}, median {np.median(rel):.3f}")
