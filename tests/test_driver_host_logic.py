"""Host-side driver logic on the CPU (no GPU): the gamma searches of driver.py over a test-double engine, and the
resume state round trip.  The double computes the same partial sums as the device kernels, in NumPy."""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
from _cpu_engine import PRED, OracleEngine  # noqa: E402


def _engine(O, data, lk):
    e = OracleEngine(O, data, None, len(lk), len(lk), 0, 1)
    e.upload_lk(PRED, lk)
    return e


def _ess(lk, gm):
    w = np.exp((lk - lk.max()) * gm)
    return w.sum() ** 2 / (w * w).sum() / len(lk)


@pytest.mark.parametrize("spread", [3.0, 300.0, 3e4])
def test_bisection_lands_on_the_ess_limit(pkg, O, data, spread):
    rs = np.random.RandomState(int(spread))
    lk = rs.standard_normal(5000) * spread
    s = pkg.SMCSettings(n_particle=len(lk), ess_search="bisection")
    es = pkg.ess_search(_engine(O, data, lk), pkg.SingleComm(), 0.1, s)
    gm = es["gm"]
    assert es["gamma_new"] == 0.1 + gm and not es["warning"]
    assert abs(es["ess"] - _ess(lk, gm)) < 1e-12
    if es["gamma_new"] < 1.0:
        assert 0.5 < es["ess"] < 0.5 + 1e-4                  # from above
        assert _ess(lk, gm + 1.01 * s.ess_bisect_tol) <= 0.5   # ... and within the bracket tolerance in gamma
    # the reference's back-off stops at or before that increment (its grid is gamma_old + (1 - gamma_old) 0.7^k)
    sb = pkg.SMCSettings(n_particle=len(lk))
    eb = pkg.ess_search(_engine(O, data, lk), pkg.SingleComm(), 0.1, sb)
    assert eb["gm"] <= gm * (1 + 1e-9) and eb["gm"] >= 0.7 * gm * (1 - 1e-9) or eb["gamma_new"] == 1.0


def test_backoff_candidates_are_the_references_floats(pkg):
    """ess_candidates reproduces the recurrence of Micmem_SMC_main.py:111-113,121,141 in Python floats."""
    s = pkg.SMCSettings()
    gms, gammas, after = pkg.ess_candidates(0.3, s)
    g = 1.0
    for k in range(s.gm_reduction_itr):
        assert gammas[k] == g and gms[k] == g - 0.3
        g = (g - 0.3) * s.gm_reduction_rate + 0.3
    assert after == g and len(gms) == 80


def test_rng_state_json_round_trip(pkg):
    from smc_lt_amd import driver
    rs = np.random.RandomState(7)
    rs.standard_normal(11)                                      # leaves a cached gaussian in the state
    st = driver._rng_state_from_json(json.loads(json.dumps(driver._rng_state_to_json(rs.get_state()))))
    r2 = np.random.RandomState()
    r2.set_state(st)
    assert np.array_equal(rs.standard_normal(5), r2.standard_normal(5)) and rs.rand() == r2.rand()


@pytest.mark.parametrize("spread,gamma_old,limit_itr", [(0.5, 0.0, 80), (40.0, 0.0, 80), (3e3, 0.0, 80), (3e5, 0.0, 80),
                                                        (3e9, 0.0, 80), (40.0, 0.3, 80), (3e3, 0.9, 80), (3e9, 0.0, 40)])
def test_fused_search_equals_the_chunked_search(pkg, O, data, spread, gamma_old, limit_itr):
    """The back-off search through smc_ess_search_global's shape (maximum + 32 candidates per synchronisation) returns exactly what the chunk-by-chunk search returns - gamma, ESS, sums, iteration count, the warning
    when no candidate passes (Micmem_SMC_main.py:143-144) - for first passing candidates at k = 0, within the first 16,
    between 16 and 32, beyond 32, and never; and it synchronises less often."""
    rs = np.random.RandomState(7)
    lk = rs.standard_normal(4000) * spread
    s = pkg.SMCSettings(n_particle=len(lk), gm_reduction_itr=limit_itr)

    class HostSide(pkg.SingleComm):
        on_device = False
    eng = _engine(O, data, lk)
    a = pkg.ess_search(eng, pkg.SingleComm(), gamma_old, s)                # on_device: the fused shape
    b = pkg.ess_search(eng, HostSide(), gamma_old, s)                      # max + one call per 16 candidates
    for k in ("gamma_new", "gm", "ess", "sum_weight", "max_lk", "iters", "warning"):
        assert a[k] == b[k], k
    assert a["syncs"] < b["syncs"]
    assert a["syncs"] == 1 + max(0, (a["iters"] - 32 + 31) // 32)
    assert a["warning"] == (spread == 3e9 and limit_itr == 40)
    print(spread, gamma_old, "first passing candidate:", a["iters"], "synchronisations", a["syncs"], "vs", b["syncs"])


# ---- Metropolis loop control in batches (round 4): the driver's bookkeeping over a scripted engine ------------------------------
class _ScriptedEngine(OracleEngine):
    """OracleEngine + the device-RNG entry points run_smc uses, with the Metropolis sweeps SCRIPTED: sweep number q of the run
    reports accepted_ever = script[q] * n and changes nothing else.  mh_sweeps_device_rng plays the control kernel in Python
    (break above thr_stop, halve below thr_halve, enqueued sweeps after the break do not happen) and records every call."""
    model = ("mm",)

    def __init__(self, O, data, priors, n, script):
        super().__init__(O, data, priors, n, n, 0, 1)
        self.script, self.q, self.calls, self.deferred = list(script), 0, [], 0

    def sample_prior_device(self, seed, lo):
        self.theta[PRED] = np.random.RandomState(seed).uniform(0.05, 3.0, (self.n_local, 3))

    def _one(self, ratio):
        acc = int(self.script[self.q % len(self.script)] * self.n_local)
        self.q += 1
        return {"accepted_now": acc // 2, "accepted_ever": acc, "n_failed": 0, "rk_attempts": 100, "mhstep_ratio": ratio,
                "cov_m": np.eye(3) * self.q}

    def mh_iteration_device_rng(self, gamma, ratio, w_cov, seed, stream, lo=0):
        self.calls.append(("iter", stream, 1, ratio))
        return self._one(ratio)

    def mh_sweeps_device_rng(self, gamma, ratio, w_cov, seed, stream0, k, thr_stop, thr_halve, lo=0):
        self.calls.append(("batch", stream0, k, ratio))
        its, stopped = [], False
        for i in range(k):
            it = self._one(ratio)
            its.append(it)
            if it["accepted_ever"] > thr_stop:
                stopped = True
                break
            if it["accepted_ever"] < thr_halve:
                ratio = ratio * 0.5
        return {"n_done": len(its), "stopped": stopped, "ratio_next": ratio, "iterations": its}

    def resample_enqueue(self, *a):
        self._rs = self.resample_global(*a)
        self.deferred += 1

    def resample_result(self):
        return self._rs


@pytest.mark.parametrize("mh_batch", [1, 2, "auto", 32])
def test_batched_metropolis_loop_keeps_the_books_of_the_per_iteration_loop(pkg, O, data, mh_batch):
    """run_smc with the loop control in the engine (SMCSettings.mh_batch) against the loop with one call and one Python decision
    per sweep, over the same scripted acceptance: loop lengths (the reference's j), every sweep's ratio and counts, the stream
    numbers (step << 16 | j) and the statistics agree; the batch sizes follow the documented policy; enqueued sweeps after a
    break are counted as such; the resampling was enqueued and its numbers read afterwards."""
    n = 64
    # acceptance per sweep: low values force halvings and long loops, a value above r_th (0.5; 0.7 at gamma = 1) ends a loop
    script = [0.05, 0.3, 0.6, 0.2, 0.55, 0.04, 0.08, 0.2, 0.3, 0.9, 0.75, 0.1, 0.8]
    runs = {}
    for mb in (0, mh_batch):
        eng = _ScriptedEngine(O, data, None, n, script)
        runs[mb] = (pkg.run_smc(eng, pkg.SMCSettings(n_particle=n, mh_batch=mb), rng="device", verbose=False, seed_device=5), eng)
    (a, ea), (b, eb) = runs[0], runs[mh_batch]
    assert a["gamma"] == b["gamma"] == 1.0 and a["step"] == b["step"] >= 2
    for ra, rb in zip(a["records"], b["records"]):
        assert ra["last_j"] == rb["last_j"] and ra["n_accept"] == rb["n_accept"] and ra["gamma_new"] == rb["gamma_new"]
        assert ra["n_offspring"] == rb["n_offspring"] == n and ra["n_tmp_before"] == rb["n_tmp_before"]
        assert [m["mhstep_ratio"] for m in ra["mh"]] == [m["mhstep_ratio"] for m in rb["mh"]]
        assert [m["accepted_ever"] for m in ra["mh"]] == [m["accepted_ever"] for m in rb["mh"]]
        assert all(np.array_equal(x["cov_m"], y["cov_m"]) for x, y in zip(ra["mh"], rb["mh"]))
    sweeps = a["stats"]["mutation_sweeps"]
    assert b["stats"]["mutation_sweeps"] == sweeps == sum(r["last_j"] + 1 for r in a["records"])
    assert b["stats"]["particle_mutation_steps"] == sweeps * n and b["stats"]["rk_attempts_mh"] == 100 * sweeps
    assert a["stats"]["mh_syncs"] == sweeps == len(ea.calls) and a["stats"]["mh_noop_sweeps"] == 0
    assert b["stats"]["mh_syncs"] == len(eb.calls) <= sweeps
    assert eb.deferred == b["step"] and ea.deferred == a["step"]
    # every batch starts at the stream number of its first sweep, (step << 16) | j
    j, step_no, k_prev = 0, 0, None
    for kind, stream0, k, ratio in eb.calls:
        assert kind == "batch"
        if (stream0 >> 16) != step_no:
            step_no, j = stream0 >> 16, 0
        assert stream0 == (step_no << 16) | j
        j += min(k, (b["records"][step_no - 1]["last_j"] + 1) - j)
    if mh_batch == 32:
        assert b["stats"]["mh_syncs"] == b["step"] and b["stats"]["mh_noop_sweeps"] > 0
    if mh_batch == 1:
        assert b["stats"]["mh_syncs"] == sweeps and b["stats"]["mh_noop_sweeps"] == 0


class _ScriptedMethEngine(_ScriptedEngine):
    """The same scripted engine presenting itself as the methanation model: every sweep has K8 work counters, which the
    per-iteration loop reads from the engine after each sweep (meth_sweep_counters / meth_sweep_check) and the batched loop takes
    from the batch log's per-iteration counter blocks (round 5)."""
    model = ("methanation", 30, 357)

    def _k8(self, q):
        return {"rk_attempts": 300 * q, "newton_iters": 500 * q + 1, "factorisations": 50 * q + 2, "failed_solves": q % 3,
                "expected_solves": 1920, "completed_solves": 1920 - 7 * q, "unsolved_items": 0, "wave_split": 0, "cancelled_solves": 7 * q}

    def loglik(self, which):
        self.last = self._k8(0)
        return super().loglik(which)

    def _one(self, ratio):
        it = super()._one(ratio)
        self.last = self._k8(self.q)
        it["rk_attempts"] = self.last["rk_attempts"]
        it["counters"] = dict(self.last)
        return it

    def mh_iteration_device_rng(self, *a, **k):
        out = dict(super().mh_iteration_device_rng(*a, **k))
        out.pop("counters")                      # the per-iteration call has no log: the driver asks the engine
        return out

    def meth_sweep_counters(self):
        c = self.last
        return {"bdf_steps": c["rk_attempts"], "newton_iters": c["newton_iters"], "factorisations": c["factorisations"], "failed_solves": c["failed_solves"]}

    def meth_sweep_check(self):
        return dict(self.last)


@pytest.mark.parametrize("mh_batch", [2, "auto", 32])
def test_batched_methanation_loop_accounts_k8_work_from_the_batch_log(pkg, O, data, mh_batch):
    n = 64
    script = [0.05, 0.3, 0.6, 0.2, 0.55, 0.04, 0.08, 0.2, 0.3, 0.9, 0.75, 0.1, 0.8]
    runs = {}
    for mb in (0, mh_batch):
        eng = _ScriptedMethEngine(O, data, None, n, script)
        runs[mb] = (pkg.run_smc(eng, pkg.SMCSettings(n_particle=n, mh_batch=mb), rng="device", verbose=False, seed_device=5), eng)
    (a, ea), (b, eb) = runs[0], runs[mh_batch]
    assert all(c[0] == "iter" for c in ea.calls) and all(c[0] == "batch" for c in eb.calls)      # methanation takes the batched entry point now
    assert a["step"] == b["step"] and [r["last_j"] for r in a["records"]] == [r["last_j"] for r in b["records"]]
    for k in ("bdf_steps", "newton_iters", "factorisations", "failed_solves", "dae_solves", "dae_solves_cancelled", "rk_attempts",
              "mutation_sweeps", "particle_mutation_steps"):
        assert a["stats"][k] == b["stats"][k] > 0, k
    assert b["stats"]["mh_syncs"] <= a["stats"]["mh_syncs"]
