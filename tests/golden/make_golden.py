#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Runs ONLY in the build container (needs /root/reference; the GPU box has no
reference and only ever reads the committed fixtures).  Nothing from the
reference is copied: the reference's Michaelis-Menten driver
(SMC_example/Micmem_SMC_main.py) is executed unmodified with runpy under inert
stand-ins (tests/golden/_shims) for the packages this image lacks (ray, numba,
assimulo, memory_profiler, seaborn - recipe: SURVEY.md section 8(c)).  All
arithmetic executed is the reference's own plus SciPy/NumPy.

Outputs (data only):
  mm_data.npz            the 6x40 pseudo-data the reference reads (t, P_obs, S0)
  mm_ref_run_n1000.npz   one complete reference run, seed 20250205, N=1000:
                         every likelihood sweep (particles in, logL out), the
                         per-step schedule parsed from the reference's own log
                         line (Micmem_SMC_main.py:254), final particles/logL
  mm_known_answers.npz   single-particle logL + 6x40 predictions from the
                         reference's log_likelihood_mm_multi at hand-picked
                         points (incl. corner cases)
  mm_prior_pdf.npz       reference cal_prior() values at hand-picked points

Usage:  python tests/golden/make_golden.py [--workers 8]
"""
import argparse
import contextlib
import io
import os
import re
import runpy
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/SMC_example"


class _Tee(io.TextIOBase):
    def __init__(self, *streams):
        self.streams = streams

    def write(self, s):
        for st in self.streams:
            st.write(s)
        return len(s)

    def flush(self):
        for st in self.streams:
            st.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=8)
    args = ap.parse_args()
    os.environ["GOLDEN_WORKERS"] = str(args.workers)
    os.environ.setdefault("MPLBACKEND", "Agg")

    if not os.path.isdir(REF):
        sys.exit("reference not present; golden vectors can only be regenerated in the build container")

    scratch = tempfile.mkdtemp(prefix="golden_mm_")
    os.symlink(os.path.join(REF, "data"), os.path.join(scratch, "data"))
    os.chdir(scratch)
    sys.path.insert(0, os.path.join(HERE, "_shims"))
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True

    import ray as ray_shim  # the stand-in

    buf = io.StringIO()
    t0 = time.time()
    with contextlib.redirect_stdout(_Tee(sys.__stdout__, buf)):
        g = runpy.run_path(os.path.join(REF, "Micmem_SMC_main.py"), run_name="__main__")
    wall = time.time() - t0
    log = buf.getvalue()

    # ---- the data the reference loaded -------------------------------------------------
    dataset = g["dataset"]
    t_grid = np.stack([np.asarray(d["t"], dtype=np.float64) for d in dataset])
    p_obs = np.stack([np.asarray(d["P_obs"], dtype=np.float64) for d in dataset])
    s0 = np.array([float(d["S0"]) for d in dataset], dtype=np.float64)
    np.savez(os.path.join(HERE, "mm_data.npz"), t=t_grid, P_obs=p_obs, S0=s0)

    # ---- sweeps --------------------------------------------------------------------------
    sweeps_theta = []
    sweeps_llk = []
    for call_args, outs in ray_shim.SWEEPS:
        sweeps_theta.append(np.array([np.asarray(a[0], dtype=np.float64) for a in call_args]))
        sweeps_llk.append(np.array([o[0] for o in outs], dtype=np.float64))
    sweeps_theta = np.stack(sweeps_theta)
    sweeps_llk = np.stack(sweeps_llk)
    # model predictions of the LAST sweep only (6x40 per particle) - keeps the file small
    last_pred = np.array([[np.asarray(p, dtype=np.float64) for p in o[1]] for o in ray_shim.SWEEPS[-1][1]])

    # ---- schedule parsed from the reference's own log line -------------------------------
    pat = re.compile(
        r"iteration:(\d+), nMH:(\d+), Calculation time:[^,]+, ESS:([^,]+), "
        r"Max Likelihood:([^,]+), New Gamma:([^,]+), Number of Adoption:(\S+)")
    rows = [m.groups() for m in pat.finditer(log)]
    sched_step = np.array([int(r[0]) for r in rows])
    sched_last_j = np.array([int(r[1]) for r in rows])
    sched_ess = np.array([float(r[2]) for r in rows])
    sched_maxlk = np.array([float(r[3]) for r in rows])
    sched_gamma = np.array([float(r[4]) for r in rows])
    sched_accept = np.array([float(r[5]) for r in rows])
    n_tmp = np.array([int(x) for x in re.findall(r"^n_tmp: (-?\d+)$", log, flags=re.M)])

    next_rand = np.random.rand()  # pins how many draws the run consumed from the global stream

    np.savez_compressed(
        os.path.join(HERE, "mm_ref_run_n1000.npz"),
        seed=np.int64(20250205),
        n_particle=np.int64(g["n_particle"]),
        sweeps_theta=sweeps_theta, sweeps_llk=sweeps_llk, last_sweep_pred=last_pred,
        sched_step=sched_step, sched_last_j=sched_last_j, sched_ess=sched_ess,
        sched_maxlk=sched_maxlk, sched_gamma=sched_gamma, sched_accept=sched_accept,
        n_tmp=n_tmp,
        final_p_pred=np.asarray(g["p_pred"], dtype=np.float64),
        final_lk=np.asarray(g["lk"], dtype=np.float64),
        final_gamma=np.float64(g["gamma_new"]), final_step=np.int64(g["step"]),
        next_rand_after_run=np.float64(next_rand),
        ref_wall_seconds=np.float64(wall), ref_workers=np.int64(args.workers),
        settings=np.array([g["ess_limit"], g["mhstep_factor"], g["mhstep_factor_cov"], g["ad_mhstep_num"],
                           g["mhstep_num"], g["r_threshold"], g["r_threshold_f"], g["r_threshold_min"],
                           g["d_gamma_max"], g["gm_reduction_itr"], g["gm_reduction_rate"], g["itr_max"]],
                          dtype=np.float64),
    )

    # ---- single-particle known answers (the reference's own function, unwrapped) ---------
    f = g["log_likelihood_mm_multi"]._f
    rng = np.random.RandomState(7)
    pts = [
        [1.2, 0.5, 0.02], [1.0, 0.4, 0.05], [1.2254248062336444, 0.5218210534927539, 0.020477688956947467],
        [2.34002158, 8.67554171, 2.07250299],
        [9.99, 1e-3, 0.5], [1e-3, 9.99, 0.5], [10.0, 10.0, 10.0], [1e-6, 1e-6, 1e-3],
        [5.0, 1e-8, 0.1], [0.05, 0.01, 0.01], [9.5, 0.02, 3.0], [0.3, 5.0, 1e-4],
        [1.2, 0.5, 1e-6], [3.3, 0.0, 1.0], [0.0, 1.0, 1.0], [7.7, 2.2, 9.9],
    ]
    pts += rng.uniform(0, 10, size=(24, 3)).tolist()
    pts += (np.array([1.2254, 0.5218, 0.02048]) + rng.standard_normal((24, 3)) * np.array([0.025, 0.0295, 0.00094])).tolist()
    pts = np.array(pts, dtype=np.float64)
    ka_l = np.empty(len(pts))
    ka_pred = np.empty((len(pts), 6, 40))
    ka_raised = np.zeros(len(pts), dtype=bool)
    for i, p in enumerate(pts):
        try:
            with np.errstate(all="ignore"):
                out = f(p)
            ka_l[i] = out[0]
            ka_pred[i] = np.array(out[1])
        except Exception as e:  # the reference raises when solve_ivp fails (ragged sol.y)
            print(f"[golden] reference raised at {p.tolist()}: {type(e).__name__}: {e}")
            ka_raised[i] = True
            ka_l[i] = np.nan
            ka_pred[i] = np.nan
    minus_inf_case = f(np.array([1.0, 1.0, 0.0]))          # sigma <= 0 -> bare -inf (Micmem_likelihood.py:53-54)
    assert minus_inf_case == -np.inf
    np.savez_compressed(os.path.join(HERE, "mm_known_answers.npz"), theta=pts, logL=ka_l, pred=ka_pred, raised=ka_raised)

    # ---- prior pdf known answers ---------------------------------------------------------
    cal_prior = g["cal_prior"]
    th = np.array([[1, 1, 1], [0, 0, 0], [10, 10, 10], [-1e-300, 1, 1], [1, 10.000000000000002, 1],
                   [5, 5, -0.0], [np.nan, 1, 1], [1, 1, np.inf], [9.999999999999998, 0.0, 5e-324]], dtype=np.float64)
    with np.errstate(all="ignore"):
        pv = cal_prior(th, g["priors"])
    normal_priors = {"Vmax": {"dist": "normal", "mu": 1.0, "sigma": 0.1},
                     "Km": {"dist": "normal", "mu": 0.0, "sigma": 5.0},
                     "sigma": {"dist": "uniform", "low": 0, "high": 10}}
    th2 = np.array([[1.0, 0.0, 1.0], [1.3, -4.0, 2.0], [5.0, 0.0, 1.0], [1.0, 0.0, 11.0], [0.9, 200.0, 3.0]])
    with np.errstate(all="ignore"):
        pv2 = cal_prior(th2, normal_priors)
    np.savez(os.path.join(HERE, "mm_prior_pdf.npz"), theta_uniform=th, pdf_uniform=pv,
             theta_mixed=th2, pdf_mixed=pv2)

    print(f"\n[golden] reference run: {wall:.1f}s, {len(ray_shim.SWEEPS)} sweeps, {len(rows)} tempering steps")
    print(f"[golden] gamma schedule: {sched_gamma.tolist()}")
    print(f"[golden] posterior mean {np.mean(g['p_pred'], axis=0).tolist()}")
    print(f"[golden] posterior std  {np.std(g['p_pred'], axis=0).tolist()}")


if __name__ == "__main__":
    main()
