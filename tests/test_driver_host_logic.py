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
