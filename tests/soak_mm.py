"""Randomised soak of the Michaelis-Menten likelihood sweep against the CPU checker (a test tool, not product code): 160 cases
with 1 .. 10^4 particles, 1 .. 8 experiments, 1 .. 60 data times, Km log-uniform over 1e-3 .. 10 (a quarter of the particles
in the stiff band: tolerance as in tests/test_gpu_fuzz_shapes.py::test_stiff_band_parity_and_its_tolerance).  Every case is
also run in PARITY mode (smc_set_exact_pow: correctly rounded pow(x, -0.2), RK stages without FMA): every particle within 1e-9
and the device-counted RK45 attempts EQUAL to the checker's.
  python tests/soak_mm.py      (about 40 s on an MI355X)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); g.load_oracle()
from oracle import oracle as O
rs = np.random.RandomState(12345)
worst = 0.0
worst_x = 0.0
t0 = time.time()
for case in range(160):
    n = int(rs.choice([1, 2, 3, 31, 63, 64, 65, 127, 129, 200, 777, 1024, 4097, 10000]))
    n_ex = int(rs.randint(1, 9)); n_t = int(rs.choice([1, 2, 5, 17, 40, 60]))
    t = np.sort(rs.uniform(0, 12, (n_ex, n_t)), axis=1)
    if case % 2 == 0: t[:, 0] = 0.0
    S0 = rs.uniform(0.05, 3.0, n_ex); P_obs = rs.uniform(0, 2, (n_ex, n_t))
    th = np.column_stack([rs.uniform(0.05, 10, n), 10.0 ** rs.uniform(-3, 1, n), rs.uniform(0.01, 5, n)])
    ref, _, iref = O.mm_loglik_batch(th, O.MMData(t=t, P_obs=P_obs, S0=S0))
    with pkg.HipEngine(n, 3) as eng:
        eng.set_model_mm(t, P_obs, S0)
        lk, pred, info = eng.loglik_host(th, want_pred=(case % 3 == 0))
        eng.upload_particles(pkg.SMC_SET_PRED, th)
        info2 = eng.loglik(pkg.SMC_SET_PRED)
        lk2 = eng.download_lk(pkg.SMC_SET_PRED)
        eng.set_fast_tail(False)                      # the compiled step function instead of the hand-written lone-chain loop:
        lk_nf, pred_nf, info_nf = eng.loglik_host(th, want_pred=(case % 3 == 0))      # the same bits, the same attempts
        assert np.array_equal(lk, lk_nf) and info_nf["rk_attempts"] == info["rk_attempts"], ("fast tail", case, n, n_ex, n_t)
        assert pred is None or np.array_equal(pred, pred_nf, equal_nan=True), ("fast tail, predictions", case)
        eng.set_fast_tail(True)
        eng.set_exact_pow(True)
        eng.set_stiff_first(case % 2 == 0)
        lk_x, _, info_x = eng.loglik_host(th, want_pred=(case % 3 == 1))
    err_x = np.max(np.abs(lk_x - ref) / np.maximum(1.0, np.abs(ref)))
    assert info_x["n_failed"] == 0 and err_x < 1e-9 and info_x["rk_attempts"] == iref["n_attempts"], \
        ("parity mode", case, n, n_ex, n_t, err_x, info_x["rk_attempts"], iref["n_attempts"])
    worst_x = max(worst_x, err_x)
    err = np.max(np.abs(lk - ref) / np.maximum(1.0, np.abs(ref)))
    assert info["n_failed"] == 0 and np.array_equal(lk, lk2), (case, n, n_ex, n_t)
    errs = np.abs(lk - ref) / np.maximum(1.0, np.abs(ref))
    assert err < 1e-6 and (errs > 1e-9).sum() <= max(2, 2e-3 * n), (case, n, n_ex, n_t, err, int((errs > 1e-9).sum()))
    worst = max(worst, err)
    if case % 20 == 0: print(case, n, n_ex, n_t, f"{err:.2e}", f"{time.time() - t0:.0f}s", flush=True)
print("soak ok, worst rel err", worst, "- parity mode:", worst_x, "(attempt totals equal to the checker's in all 160 cases)")
