"""The adaptive likelihood-tempering loop on the HIP engine.

Host-side mirror of the reference's driver, which is inline script code
(SMC_example/Micmem_SMC_main.py:95-262; SMC_methanation/SMC_methanation_main.py:194-418 is the same
loop).  Stage by stage the control flow, the hyper-parameter names and the per-step log line are the
reference's; the per-particle work of every stage runs in HIP kernels through the engine:

  A2  engine.loglik                      sim_particle                     main:98,229
  A3  engine.max_lk_local/ess_partials   weights, sum, ESS                main:116-134
  A4  ess_search (this file)             geometric gamma back-off         main:111-113,120-144
  A5  engine.resample_phase1..3          residual-systematic resampling   main:147-184
  A6  engine.moment_*                    np.cov(p_filt.T, bias=True)      main:212-215
  A7-A9 engine.mh_step_*                 proposal, prior mask, likelihood, accept/select  main:220-241
  A10 run_smc (this file)                MH loop control                  main:187-208,243-252

Two RNG modes:
  rng="numpy"   parity mode: every random number is drawn by NumPy's global legacy generator in the
                reference's order and uploaded - identical streams on identical seeds;
  rng="device"  Philox4x32-10 in the kernels (keyed by seed, GLOBAL particle index, step, iteration),
                nothing but a few scalars crosses PCIe; results are independent of the number of GPUs.

Particles are sharded by rank in contiguous blocks (rank-major global order); `comm` supplies the four
small collectives, the engine exchanges resampled particles itself.
"""
from __future__ import annotations

import math
import time
from dataclasses import dataclass, field

import numpy as np

from .binding import SMC_MAX_ESS_CAND, SMC_SET_FILT, SMC_SET_PRED
from .comm import SingleComm


@dataclass
class SMCSettings:
    """Hyper-parameters, names and defaults of Micmem_settings.py:15-31,47,53,63-67,90."""
    n_particle: int = 1000
    ess_limit: float = 0.5
    mhstep_factor: float = 0.5
    mhstep_factor_cov: float = 0.5
    ad_mhstep_num: int = 20
    mhstep_num: int = 5
    r_threshold: float = 0.5
    r_threshold_f: float = 0.7
    r_threshold_min: float = 0.1
    d_gamma_max: float = 1
    gm_reduction_itr: int = 80
    gm_reduction_rate: float = 0.7
    itr_max: int = 50
    est_sigma: bool = True
    sigma_true: float = 5
    seed: int = 20250205
    rtol: float = 1e-3
    atol: float = 1e-6
    prior_mode: str = "mask"          # "mask" | "ratio_mask" | "ratio"  (HipEngine.set_prior_mode)
    resampling: str = "residual_systematic"   # | "systematic"  (HipEngine.set_resampling; the reference has only the first)
    ess_search: str = "backoff"       # the reference's geometric back-off (main:111-144) | "bisection"
    ess_bisect_tol: float = 1e-9      # bisection: bracket width in gamma at which the search stops
    early_reject: bool = True         # stop a solve once its proposal is certainly rejected (exact; HipEngine.set_early_reject)
    stiff_first: bool = True          # hand the predictably long solves out first (same results; HipEngine.set_stiff_first)
    in_phase: bool = True             # homogeneous Metropolis sweeps run their waves in phase (same results; HipEngine.set_in_phase)
    cost_order: bool = True           # heterogeneous ones hand their solves out by cost class, in phase (same results; set_cost_order)
    exact_pow: object = None          # step controller of the RK45 kernels: None = parity arithmetic (correctly rounded pow(x, -0.2), the
                                      # mode pinned to 1e-9 / equal step sequences against the reference) with rng="numpy", the fast
                                      # inverse fifth root with rng="device"; True / False force one (HipEngine.set_exact_pow)
    defer_resample: bool = True       # enqueue the resampling without a host synchronisation (one rank; HipEngine.resample_enqueue)
    pinned_results: bool = True       # final particles / likelihoods arrive in page-locked host arrays (HipEngine.download_*(pinned=True))
    mh_batch: object = "auto"         # Metropolis iterations enqueued per host synchronisation, their loop control (main:243-249) on the
                                      # device (HipEngine.mh_sweeps_device_rng; device RNG; not for user models): "auto" = as many as the
                                      # previous tempering step needed, an int = that many, 0 = one call and one decision per iteration
    priors: dict = field(default_factory=lambda: {
        "Vmax": {"dist": "uniform", "low": 0, "high": 10},
        "Km": {"dist": "uniform", "low": 0, "high": 10},
        "sigma": {"dist": "uniform", "low": 0, "high": 10},
    })

    @property
    def num_est_params(self):
        return len(self.priors)

    def w_cov(self):
        d = self.num_est_params            # Micmem_settings.py:94-97
        w = np.ones((d, d))
        for i in range(d):
            w[i, :] = self.mhstep_factor_cov
            w[i, i] = self.mhstep_factor
        return w


def sample_prior(priors: dict, n_particle: int) -> np.ndarray:
    """Micmem_settings.py:69-87 - global NumPy RNG, one draw of size N per parameter, parameter-major."""
    p_pred = np.zeros((n_particle, len(priors)))
    for j, (name, p) in enumerate(priors.items()):
        if p["dist"] in ("normal", "flat"):
            p_pred[:, j] = np.random.normal(loc=p["mu"], scale=p["sigma"], size=n_particle)
        elif p["dist"] == "uniform":
            p_pred[:, j] = np.random.uniform(low=p["low"], high=p["high"], size=n_particle)
        else:
            raise ValueError(f"Unknown distribution: {p['dist']}")
    return p_pred


def ess_candidates(gamma_old: float, s: SMCSettings):
    """The grid of tempering increments the reference's back-off visits, produced by its own recurrence
    (main:111-113,121,141) in Python floats so the values are bit-identical to the reference's."""
    gamma_new = gamma_old + s.d_gamma_max
    if gamma_new > 1.0:
        gamma_new = 1.0
    gms, gammas = [], []
    for _ in range(s.gm_reduction_itr):
        gms.append(gamma_new - gamma_old)
        gammas.append(gamma_new)
        gamma_new = (gamma_new - gamma_old) * s.gm_reduction_rate + gamma_old
    return gms, gammas, gamma_new


def _on_device(comm) -> bool:
    """True when the communicator's reductions can run inside the engine (one rank, or the engine's own RCCL communicator):
    the driver then calls the *_global entry points - partials stay on the device, RCCL reduces them in place on the
    engine's stream, one read-back per stage.  Host-side communicators (tests) get the *_local calls + their own reduce."""
    return bool(getattr(comm, "on_device", False))


def _max_lk(engine, comm):
    if _on_device(comm):
        return float(engine.max_lk_global())
    return float(comm.allreduce_max([engine.max_lk_local()])[0])


def _ess_sums(engine, comm, max_lk, gms):
    """sum_w[k], sum_w2[k] over ALL particles for the candidate increments gms."""
    if _on_device(comm):
        return engine.ess_partials_global(max_lk, gms)
    sw, sw2 = engine.ess_partials(max_lk, gms)
    tot = comm.allreduce_sum(np.concatenate([sw, sw2]))
    return tot[:len(gms)], tot[len(gms):]


def ess_search(engine, comm, gamma_old: float, s: SMCSettings, chunk: int = SMC_MAX_ESS_CAND, hint=None):
    """main:111-144.  Candidates are evaluated `chunk` at a time by one fused pass over lk each.  With the reductions in
    the engine (one rank / RCCL) the maximum and the first 32 candidates of the grid cost ONE synchronisation
    (smc_ess_search_global): the back-off grid gamma_old + (1 - gamma_old) * 0.7^k is known in advance, the decision which
    candidate is the first to pass is the reference's own expression on the returned sums."""
    n = s.n_particle
    fused = _on_device(comm) and s.ess_search != "bisection" and getattr(engine, "ess_search_global", None) is not None
    gms, gammas, gamma_after_all = ess_candidates(gamma_old, s)
    if fused:
        # the first tempering step needs 17-18 candidates (gamma_1 ~ 2e-3 = 0.7^17), later ones sometimes more than 16: a
        # second 24-us pass is cheaper than a second synchronisation, so a call evaluates 32 - unless the previous search of the
        # run (`hint` = its iteration count) stopped well inside the first 16, as most do (11.8 on average): then one pass, and
        # the rare search that needs more pays its second synchronisation below
        first = chunk if (hint is not None and hint <= chunk - 4) else 2 * chunk
        max_lk, sw0, sw20 = engine.ess_search_global(gms[:first], with_max=True)
    else:
        max_lk = _max_lk(engine, comm)                                        # :116
        if s.ess_search == "bisection":
            return ess_bisection(engine, comm, gamma_old, s, max_lk, chunk)
    iters = 0
    launches = 0
    syncs = 0 if fused else 1                                                 # the maximum's own round trip
    ess = sum_w = None
    k0 = 0
    while k0 < len(gms):
        if fused:
            width = first if k0 == 0 else 2 * chunk
            part = gms[k0:k0 + width]
            if k0 == 0:
                sw, sw2 = sw0, sw20
            else:
                _, sw, sw2 = engine.ess_search_global(part, with_max=False)
            launches += (len(part) + chunk - 1) // chunk
        else:
            width = chunk
            part = gms[k0:k0 + chunk]
            sw, sw2 = _ess_sums(engine, comm, max_lk, part)
            launches += 1
        syncs += 1
        for i in range(len(part)):
            sum_w = float(sw[i])
            ess = 1.0 / (float(sw2[i]) / (sum_w * sum_w)) / n                 # :130-134
            iters += 1
            if ess > s.ess_limit:                                             # :136
                return {"gamma_new": gammas[k0 + i], "gm": part[i], "ess": ess, "sum_weight": sum_w, "max_lk": max_lk,
                        "iters": iters, "launches": launches, "syncs": syncs, "warning": False}
        k0 += width
    # no candidate passed: the weights of the last trial are kept, gamma has been shrunk once more (:141-144)
    return {"gamma_new": gamma_after_all, "gm": gms[-1], "ess": ess, "sum_weight": sum_w, "max_lk": max_lk,
            "iters": iters, "launches": launches, "syncs": syncs, "warning": True}


def ess_bisection(engine, comm, gamma_old: float, s: SMCSettings, max_lk: float, chunk: int = SMC_MAX_ESS_CAND):
    """The tempering increment at which ESS/N crosses ess_limit, located by sectioning (BASELINE.json: "adaptive
    tempering via ESS bisection"; not in the reference, whose back-off shrinks the increment by 0.7 until the ESS
    passes and so stops up to 30 % short).  Every pass over lk evaluates `chunk` increments at once, so the bracket
    shrinks (chunk+1)-fold per launch.  Returns the largest increment found with ESS/N > ess_limit."""
    n = s.n_particle
    g_hi = min(1.0, gamma_old + s.d_gamma_max)
    hi_gm = g_hi - gamma_old
    state = {"iters": 0, "launches": 0}

    def evaluate(gms):
        sw, sw2 = _ess_sums(engine, comm, max_lk, list(gms))
        state["launches"] += 1
        state["iters"] += len(gms)
        return sw, 1.0 / (sw2 / (sw * sw)) / n

    sw, ess = evaluate([hi_gm])
    if ess[0] > s.ess_limit:                       # the whole remaining increment passes
        return {"gamma_new": g_hi, "gm": hi_gm, "ess": float(ess[0]), "sum_weight": float(sw[0]),   # g_hi: exactly 1.0 at the end
                "max_lk": max_lk, "iters": state["iters"], "launches": state["launches"], "syncs": 1 + state["launches"],
                "warning": False}
    lo_gm, best = 0.0, None
    while hi_gm - lo_gm > s.ess_bisect_tol:
        gms = lo_gm + (hi_gm - lo_gm) * np.arange(1, chunk + 1) / (chunk + 1)
        sw, ess = evaluate(gms)
        ok = np.nonzero(ess > s.ess_limit)[0]
        if len(ok):                                # ESS falls with gamma: the last passing point bounds from below
            k = int(ok[-1])
            best = (float(gms[k]), float(ess[k]), float(sw[k]))
            lo_gm = float(gms[k])
            hi_gm = float(gms[k + 1]) if k + 1 < len(gms) else hi_gm
        else:
            hi_gm = float(gms[0])
    if best is None:                               # not even the smallest increment passes: take it, warn like :141-144
        sw, ess = evaluate([hi_gm])
        best = (hi_gm, float(ess[0]), float(sw[0]))
    gm, e, w = best
    return {"gamma_new": gamma_old + gm, "gm": gm, "ess": e, "sum_weight": w, "max_lk": max_lk,
            "iters": state["iters"], "launches": state["launches"], "syncs": 1 + state["launches"], "warning": e <= s.ess_limit}


def resample(engine, comm, es: dict, wrand_u: float, s: SMCSettings, first_step: bool, defer: bool = False):
    """main:147-184 across ranks: residual sums -> prefix; offspring -> output slot bases; gather/exchange.
    defer (reductions in the engine only): the resampling is ENQUEUED - with one rank without any synchronisation - and the
    returned callable delivers the two logged numbers later, after the caller's next synchronisation."""
    inv_Np = 1 / s.n_particle                                                 # Micmem_settings.py:17
    wrand = wrand_u * inv_Np                                                  # :156
    if defer and _on_device(comm) and getattr(engine, "resample_enqueue", None) is not None:
        engine.resample_enqueue(es["max_lk"], es["gm"], es["sum_weight"], wrand, first_step)

        def later():
            r = engine.resample_result()
            return {"n_offspring": r["n_offspring"], "n_tmp_before": s.n_particle - r["count_sum"]}
        return later
    if _on_device(comm):                                                      # all three phases + their all-gathers in the engine
        r = engine.resample_global(es["max_lk"], es["gm"], es["sum_weight"], wrand, first_step)
        return {"n_offspring": r["n_offspring"], "n_tmp_before": s.n_particle - r["count_sum"]}
    r_loc, c_loc = engine.resample_phase1(es["max_lk"], es["gm"], es["sum_weight"])
    r_all = comm.allgather([r_loc])[:, 0]
    prefix = 0.0
    for q in range(comm.rank):                                               # running sum in rank order
        prefix = prefix + float(r_all[q])
    o_loc = engine.resample_phase2(es["max_lk"], es["gm"], es["sum_weight"], prefix, wrand)
    o_all = comm.allgather_i64([o_loc])[:, 0]
    bases = np.concatenate([[0], np.cumsum(o_all)[:-1]]).astype(np.int64)
    engine.resample_phase3(bases, o_all, first_step)
    c_all = int(comm.allreduce_sum_i64([c_loc])[0])
    return {"n_offspring": int(o_all.sum()), "n_tmp_before": s.n_particle - c_all}


def proposal_cov(engine, comm, s: SMCSettings, w_cov):
    """np.cov(p_filt.T, bias=True) * w_cov  (main:212-215) from device partial sums."""
    n = s.n_particle
    mean = comm.allreduce_sum(engine.moment_sums_local()) / n                 # X.mean(axis=1)
    cent = comm.allreduce_sum(engine.moment_centered_local(mean).ravel()).reshape(len(mean), len(mean))
    cov = cent * np.true_divide(1, n)                                         # c *= 1/fact, fact = N (bias)
    return cov * w_cov


def mvn_transform(cov_m):
    """The factor NumPy's legacy multivariate_normal multiplies standard normals with:
    x = z @ (sqrt(s)[:,None] * v), (u,s,v) = svd(cov)."""
    (u, sv, v) = np.linalg.svd(cov_m)
    return np.sqrt(sv)[:, None] * v


def write_particles_csv(path, arr, header=None):
    """One particle table in the reference's two on-disk formats: bare `np.savetxt(..., delimiter=',')`
    (SMC_methanation_main.py:181,422; methanation_functions.py:233-234) or, with column names, the pandas CSV of
    `Posterior_Distribution.csv` (methanation_functions.py:231-232).  The per-step dumps of run_smc and the drop-in
    `SavePosteriorcsv` both write through here."""
    if header is None:
        np.savetxt(path, arr, delimiter=',')
    else:
        import pandas as pd
        pd.DataFrame(arr, columns=list(header)).to_csv(path, index=False)
    return path


def _dump(dump_dir, name, arr, rank, world, header=None):
    """Per-step particle dumps in the reference's format.  With several ranks every rank writes its
    own block (suffix _rank<r>); concatenating them in rank order gives the reference's file."""
    import os
    os.makedirs(os.path.join(dump_dir, "pred"), exist_ok=True)
    suffix = "" if world == 1 else f"_rank{rank}"
    return write_particles_csv(os.path.join(dump_dir, name + suffix + ".csv"), arr, header)


def _save_state(dump_dir, step, rank, world, state):
    """Next to pred/{step}_p_pred.csv: what the loop needs to go on from there (the reference only writes, it cannot
    resume: SURVEY.md 2, checkpoint row)."""
    import json
    import os
    suffix = "" if world == 1 else f"_rank{rank}"
    with open(os.path.join(dump_dir, "pred", f"{step}_state{suffix}.json"), "w") as fh:
        json.dump(state, fh)


def _rng_state_to_json(st):
    return [st[0], [int(v) for v in st[1]], int(st[2]), int(st[3]), float(st[4])]


def _rng_state_from_json(j):
    return (j[0], np.array(j[1], dtype=np.uint32), j[2], j[3], j[4])


def run_smc(engine, s: SMCSettings | None = None, comm=None, rng: str = "numpy", verbose: bool = True,
            p_pred0=None, seed_device: int | None = None, log=print, dump_dir: str | None = None,
            resume_from: tuple | None = None):
    """Run the tempering loop (main:95-262).  The engine must already hold the model and the prior.

    dump_dir: per-step dumps in the reference's formats plus a small state file per step.
    resume_from=(dir, step): continue a dumped run after tempering step `step` - particles from pred/{step}_p_pred.csv,
    likelihoods recomputed, gamma / logZ / random state from pred/{step}_state.json; with rng="device" the continuation
    is bit-identical to the uninterrupted run (tests/test_gpu_parity.py).

    Returns a dict: final particles (this rank's block), lk, schedule records, logZ, counters.
    """
    s = s or SMCSettings()
    comm = comm or SingleComm()
    n = s.n_particle
    d = s.num_est_params
    world, rank = comm.size, comm.rank
    n_local = engine.n_local
    assert n_local * world == n, "n_particle must equal world * n_local"
    lo = rank * n_local
    w_cov = s.w_cov()
    engine.set_prior_mode(s.prior_mode)
    engine.set_resampling(s.resampling)
    if hasattr(engine, "set_early_reject"):
        engine.set_early_reject(s.early_reject)
    if hasattr(engine, "set_stiff_first"):
        engine.set_stiff_first(s.stiff_first)
    if hasattr(engine, "set_in_phase"):
        engine.set_in_phase(s.in_phase)
    if hasattr(engine, "set_cost_order"):
        engine.set_cost_order(s.cost_order)
    if hasattr(engine, "set_exact_pow"):           # parity mode (the reference's NumPy stream): libm-grade step-controller power
        engine.set_exact_pow(rng == "numpy" if s.exact_pow is None else bool(s.exact_pow))
    start_time = time.perf_counter()
    stats = {"rk_attempts": 0, "rk_attempts_mh": 0, "n_failed": 0, "mutation_sweeps": 0, "ess_iters": 0,
             "ess_launches": 0, "ess_syncs": 0, "ess_search_s": 0.0, "particle_mutation_steps": 0,
             "mh_syncs": 0, "mh_noop_sweeps": 0}       # host synchronisations of the Metropolis loops; enqueued sweeps that found the loop ended

    meth = getattr(engine, "model", ("",))[0] == "methanation"

    def account(info, counters=None):
        stats["rk_attempts"] += info["rk_attempts"]
        stats["n_failed"] += info["n_failed"]
        if meth:                                   # K8 work counters of the sweep just done
            if counters is None:                   # one sweep per call: the engine's last-sweep counters
                k8, chk = engine.meth_sweep_counters(), engine.meth_sweep_check()
            else:                                  # an iteration of a batch: its own snapshot from the batch log
                k8 = {"bdf_steps": counters["rk_attempts"], "newton_iters": counters["newton_iters"],
                      "factorisations": counters["factorisations"], "failed_solves": counters["failed_solves"]}
                chk = counters
            for k, v in k8.items():
                stats[k] = stats.get(k, 0) + v
            stats["dae_solves"] = stats.get("dae_solves", 0) + chk["completed_solves"]      # solves done / skipped by the exact early rejection
            stats["dae_solves_cancelled"] = stats.get("dae_solves_cancelled", 0) + chk.get("cancelled_solves", 0)

    gamma_old, gamma_new = 0.0, 1.0
    logZ = 0.0
    first_step_no = 1
    if rng not in ("numpy", "device"):
        raise ValueError("rng must be 'numpy' or 'device'")
    seed_device = s.seed if seed_device is None else seed_device
    host_rng = np.random.RandomState(seed_device & 0xFFFFFFFF) if rng == "device" else None   # one scalar draw per step (:156)
    if resume_from is not None:
        import json
        import os
        rdir, rstep = resume_from
        suffix = "" if world == 1 else f"_rank{rank}"
        block = np.loadtxt(os.path.join(rdir, "pred", f"{rstep}_p_pred{suffix}.csv"), delimiter=",", ndmin=2)
        assert block.shape == (n_local, d), "the dump does not match this engine's block"
        with open(os.path.join(rdir, "pred", f"{rstep}_state{suffix}.json")) as fh:
            st = json.load(fh)
        assert st["rng"] == rng and st["n_particle"] == n
        engine.upload_particles(SMC_SET_PRED, block)
        engine.upload_particles(SMC_SET_FILT, block)          # after the commit of a step both sets hold the same rows
        gamma_old, logZ, first_step_no = st["gamma"], st["logZ"], rstep + 1
        if rng == "numpy":
            np.random.set_state(_rng_state_from_json(st["rng_state"]))
        else:
            host_rng.set_state(_rng_state_from_json(st["rng_state"]))
    elif rng == "numpy":                                                      # prior draw (Micmem_settings.py:47,84-87)
        if p_pred0 is None:
            np.random.seed(s.seed)
            p_pred0 = sample_prior(s.priors, n)
        engine.upload_particles(SMC_SET_PRED, np.ascontiguousarray(p_pred0[lo:lo + n_local]))
    else:
        if p_pred0 is None:
            engine.sample_prior_device(seed_device, lo)
        else:
            engine.upload_particles(SMC_SET_PRED, np.ascontiguousarray(p_pred0[lo:lo + n_local]))

    if dump_dir and resume_from is None:
        _dump(dump_dir, "pred/first_p_pred", engine.download_particles(SMC_SET_PRED), rank, world)
    info = engine.loglik(SMC_SET_PRED)                                        # main:98
    account(info)
    if int(comm.allreduce_sum_i64([info["n_failed"]])[0]):
        raise RuntimeError("an RK45 solve failed in the initial sweep (the reference raises here)")
    if resume_from is not None:
        engine.upload_lk(SMC_SET_FILT, engine.download_lk(SMC_SET_PRED))

    records = []
    step = 0
    for step in range(first_step_no, s.itr_max):                                          # :109
        t_search = time.perf_counter()
        es = ess_search(engine, comm, gamma_old, s, hint=records[-1]["ess_iters"] if records else None)   # :111-144
        stats["ess_search_s"] += time.perf_counter() - t_search              # wall time of the search: max + passes + read-backs
        gamma_new, ess, max_lk = es["gamma_new"], es["ess"], es["max_lk"]
        stats["ess_iters"] += es["iters"]
        stats["ess_launches"] += es["launches"]
        stats["ess_syncs"] += es.get("syncs", es["launches"] + 1)
        if verbose and rank == 0:
            if es["warning"]:
                log("ess reduction warning: ess = ", ess)
            else:
                log(f"ess>ess_limit:{ess}")
        dlogZ = es["gm"] * max_lk + math.log(es["sum_weight"] / n)
        logZ += dlogZ
        wrand_u = np.random.rand() if rng == "numpy" else host_rng.rand()     # :156
        # :147-184.  Not verbose: enqueued; its two logged numbers are read after the Metropolis loop's synchronisation
        rs = resample(engine, comm, es, wrand_u, s, first_step=(step == 1), defer=s.defer_resample and not verbose)
        if verbose and rank == 0:
            log("n_tmp:", rs["n_tmp_before"])
        engine.reset_accept_flags()                                           # :187
        mhstep_ratio = 1.0                                                    # :190
        if gamma_new >= 1.0:                                                  # :193-208
            nMH, r_th = s.ad_mhstep_num, s.r_threshold_f
        else:
            nMH, r_th = s.mhstep_num, s.r_threshold
        j = 0
        acc_ever = 0
        mh_log = []
        fused = rng == "device" and _on_device(comm)
        batched = (fused and s.mh_batch not in (0, None) and getattr(engine, "mh_sweeps_device_rng", None) is not None
                   and getattr(engine, "model", ("",))[0] in ("mm", "methanation"))
        if batched:
            # The same loop with its control on the device: the iterations are enqueued back to back, the engine's control
            # kernel takes the decisions of :243-249 between them, and the host synchronises once per batch (as a rule once per
            # tempering step).  What Python did per iteration - counting, logging, raising on a failed solve - happens here
            # on the batch's log, in the same order.  Every rank computes the same batch sizes (they depend on records only).
            done, stopped = 0, False
            depth = s.mh_batch
            if depth == "auto":      # as many as the previous step's loop took; a first batch that turns out short is continued
                depth = (records[-1]["last_j"] + 1) if records else 2
            while done < nMH and not stopped:
                k = max(1, min(int(depth), nMH - done, 32))
                out = engine.mh_sweeps_device_rng(gamma_new, mhstep_ratio, w_cov, seed_device, (step << 16) | done, k,
                                                  r_th * n, s.r_threshold_min * n, lo)
                stats["mh_syncs"] += 1
                stats["mh_noop_sweeps"] += k - out["n_done"]
                for it in out["iterations"]:
                    account({"rk_attempts": it["rk_attempts"], "n_failed": 0}, it.get("counters"))
                    stats["rk_attempts_mh"] += it["rk_attempts"]
                    stats["mutation_sweeps"] += 1
                    stats["particle_mutation_steps"] += n
                    acc_ever = it["accepted_ever"]
                    mh_log.append({"cov_m": it["cov_m"], "mhstep_ratio": it["mhstep_ratio"], "accepted_now": it["accepted_now"],
                                   "accepted_ever": acc_ever})
                    if it["n_failed"]:
                        raise RuntimeError("an RK45 solve failed during an MH sweep (the reference raises here)")
                    if verbose and rank == 0:
                        if acc_ever > r_th * n:                               # :243
                            log(f"r_ac.sum() > r_th * n_particle:{float(acc_ever)}")
                        elif acc_ever < s.r_threshold_min * n:                # :247
                            log(it["mhstep_ratio"] * 0.5)
                done += out["n_done"]
                stopped = out["stopped"]
                mhstep_ratio = out["ratio_next"]
                depth = 1 if s.mh_batch == "auto" else depth
            j = done - 1
        for j in ([] if batched else range(nMH)):                             # :209
            stats["mh_syncs"] += 1
            if fused:                                                         # :212-241 in one call, one synchronisation
                out = engine.mh_iteration_device_rng(gamma_new, mhstep_ratio, w_cov, seed_device, (step << 16) | j, lo)
                cov_m = out["cov_m"]
                tot = [out["accepted_ever"], out["accepted_now"], out["n_failed"]]     # already summed over the ranks
                out = dict(out, n_failed=0)                                   # account() counts this rank's share only once
            else:
                cov_m = proposal_cov(engine, comm, s, w_cov)                  # :212-215
                if rng == "numpy":
                    noise = np.random.multivariate_normal(np.zeros(d), cov_m, n)  # :220
                    rr = np.random.uniform(0, 1, n)                           # :235
                    out = engine.mh_step_host_rng(gamma_new, mhstep_ratio, noise[lo:lo + n_local], rr[lo:lo + n_local])
                else:
                    out = engine.mh_step_device_rng(gamma_new, mhstep_ratio, mvn_transform(cov_m), seed_device,
                                                    (step << 16) | j, lo)
                tot = comm.allreduce_sum_i64([out["accepted_ever"], out["accepted_now"], out["n_failed"]])
            account(out)
            stats["rk_attempts_mh"] += out["rk_attempts"]
            stats["mutation_sweeps"] += 1
            stats["particle_mutation_steps"] += n
            acc_ever = int(tot[0])
            mh_log.append({"cov_m": cov_m, "mhstep_ratio": mhstep_ratio, "accepted_now": int(tot[1]),
                           "accepted_ever": acc_ever})
            if int(tot[2]):
                raise RuntimeError("an RK45 solve failed during an MH sweep (the reference raises here)")
            if acc_ever > r_th * n:                                           # :243
                if verbose and rank == 0:
                    log(f"r_ac.sum() > r_th * n_particle:{float(acc_ever)}")
                break
            if acc_ever < s.r_threshold_min * n:                              # :247
                mhstep_ratio = mhstep_ratio * 0.5
                if verbose and rank == 0:
                    log(mhstep_ratio)
        if callable(rs):
            rs = rs()
        engine.commit_filt_to_pred()                                          # :251-252
        if verbose and rank == 0:
            log(f"iteration:{step}, nMH:{j}, Calculation time:{time.perf_counter() - start_time}, ESS:{ess}, "
                f"Max Likelihood:{max_lk}, New Gamma:{gamma_new}, Number of Adoption:{float(acc_ever)}")
        records.append({"step": step, "gamma_old": gamma_old, "gamma_new": gamma_new, "gm": es["gm"], "ess": ess,
                        "ess_iters": es["iters"], "sum_weight": es["sum_weight"], "max_lk": max_lk, "dlogZ": dlogZ,
                        "wrand_u": wrand_u, "last_j": j, "n_accept": float(acc_ever), "n_offspring": rs["n_offspring"],
                        "n_tmp_before": rs["n_tmp_before"], "mh": mh_log})
        if gamma_new == 1.0:                                                  # :259
            break
        gamma_old = gamma_new
        if dump_dir:                                                          # SMC_methanation_main.py:422
            _dump(dump_dir, f"pred/{step}_p_pred", engine.download_particles(SMC_SET_PRED), rank, world)
            rst = np.random.get_state() if rng == "numpy" else host_rng.get_state()
            _save_state(dump_dir, step, rank, world, {"step": step, "gamma": gamma_new, "logZ": logZ, "rng": rng,
                                                      "n_particle": n, "rng_state": _rng_state_to_json(rst)})
    if gamma_new < 1.0 and verbose and rank == 0:
        log("tempering does't complete: last gamma =", gamma_new)             # :270-271
    engine.synchronize()
    if dump_dir:                                                              # SavePosteriorcsv, :434-436
        final = engine.download_particles(SMC_SET_PRED)
        _dump(dump_dir, "pred/last_p_pred", final, rank, world)
        _dump(dump_dir, "Posterior_Distribution", final, rank, world, header=list(s.priors.keys()))
    pin = {"pinned": True} if (s.pinned_results and getattr(engine, "pinned_downloads", False)) else {}
    return {"p_pred": engine.download_particles(SMC_SET_PRED, **pin), "lk": engine.download_lk(SMC_SET_PRED, **pin),
            "records": records, "logZ": logZ, "gamma": gamma_new, "step": step, "stats": stats,
            "wall_s": time.perf_counter() - start_time}
