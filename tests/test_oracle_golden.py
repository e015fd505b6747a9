"""CPU tests: the oracle (oracle/) against the golden vectors generated from the reference run
(tests/golden/make_golden.py) and against SciPy/NumPy called live.  These pin the oracle; the GPU
parity tests (-m gpu) then compare the HIP path with the oracle."""
import numpy as np
import pytest

from conftest import relerr

# Tolerance of the oracle's C RK45 + log-likelihood against the reference's own outputs: the only
# differences are BLAS summation order inside SciPy's np.dot calls and libm vs NumPy SIMD log.
TOL_LOGL = 1e-9


def test_known_answers(O, data, known_answers):
    lk, pred, info = O.mm_loglik_batch(known_answers["theta"], data, want_pred=True)
    assert not known_answers["raised"].any()
    assert info["n_failed"] == 0
    assert relerr(lk, known_answers["logL"]).max() < TOL_LOGL
    assert np.abs(pred - known_answers["pred"]).max() < 1e-9


def test_sigma_nonpositive_is_minus_inf(O, data):
    lk, _, _ = O.mm_loglik_batch(np.array([[1.0, 1.0, 0.0], [1.0, 1.0, -2.0]]), data)
    assert np.all(np.isneginf(lk))


def test_reference_run_replay(O, data, golden_run):
    """The oracle driver on the reference's seed reproduces the reference run: identical random stream,
    bit-identical particles in all 34 sweeps, gamma schedule bit-exact, logL within TOL_LOGL."""
    g = golden_run
    out = O.run_smc(data, O.SMCSettings(), seed=int(g["seed"]), n_threads=0)
    assert out["step"] == int(g["final_step"]) and len(out["sweeps"]) == g["sweeps_theta"].shape[0]
    gam = np.array([r.gamma_new for r in out["records"]])
    assert np.array_equal(gam, g["sched_gamma"])
    assert np.allclose([r.ess for r in out["records"]], g["sched_ess"], rtol=1e-12, atol=0)
    assert np.array_equal([r.n_accept for r in out["records"]], g["sched_accept"])
    assert np.array_equal([r.last_j for r in out["records"]], g["sched_last_j"])
    assert np.array_equal([r.n_tmp for r in out["records"]], np.zeros(len(gam)))
    for k, (th, l) in enumerate(out["sweeps"]):
        assert np.array_equal(th, g["sweeps_theta"][k]), f"sweep {k}: particles differ"
        assert relerr(l, g["sweeps_llk"][k]).max() < TOL_LOGL, f"sweep {k}"
    assert np.array_equal(out["p_pred"], g["final_p_pred"])
    assert relerr(out["lk"], g["final_lk"]).max() < TOL_LOGL
    assert np.random.rand() == float(g["next_rand_after_run"])   # same number of draws consumed
    assert np.isfinite(out["logZ"])


def test_rk45_against_scipy_live(O, data):
    """C restatement of solve_ivp(RK45) vs SciPy itself: values, nfev and step counts."""
    from scipy.integrate import solve_ivp
    rs = np.random.RandomState(3)
    pts = np.vstack([rs.uniform(0, 10, size=(60, 2)),
                     np.array([1.2254, 0.5218]) + rs.standard_normal((40, 2)) * np.array([0.025, 0.03])])
    for Vmax, Km in pts:
        for e in range(data.n_ex):
            t = data.t[e]
            sol = solve_ivp(lambda tt, S: -Vmax * S / (Km + S), (t[0], t[-1]), [data.S0[e]], t_eval=t, method="RK45")
            y, st = O.rk45_solve(Vmax, Km, data.S0[e], t)
            assert st["status"] == 0 and st["n_out"] == len(t)
            assert st["nfev"] == sol.nfev
            assert np.abs(y - sol.y[0]).max() < 1e-9 * max(1.0, np.abs(sol.y[0]).max())


def test_scipy_port_matches_c(O, data):
    rs = np.random.RandomState(5)
    theta = rs.uniform(0.01, 10, size=(24, 3))
    a, _, _ = O.mm_loglik_batch(theta, data)
    b = O.mm_loglik_batch_scipy(theta, data, n_workers=1)
    assert relerr(a, b).max() < TOL_LOGL


def test_np_pairwise_sum_bitexact(O):
    import ctypes
    rs = np.random.RandomState(0)
    for n in [1, 7, 8, 9, 40, 127, 128, 129, 1000, 4099]:
        a = rs.standard_normal(n) ** 2
        got = O.lib().oracle_np_sum(a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), n)
        assert got == np.sum(a)


def test_resample_c_equals_python_loop(O):
    rs = np.random.RandomState(11)
    for N, d in [(50, 3), (1000, 3), (777, 5)]:
        lk = rs.standard_normal(N) * 30
        w = np.exp((lk - lk.max()) * 0.05)
        w = w / w.sum()
        p_pred = rs.standard_normal((N, d))
        u = rs.rand()
        f1, l1 = np.full((N, d), -7.0), np.full(N, -7.0)
        f2, l2 = f1.copy(), l1.copy()
        pis1, n1, nt1 = O.resample(w, u, p_pred, lk, f1, l1)
        pis2, n2, nt2 = O.resample_python(w.copy(), u, p_pred, lk, f2, l2)
        assert np.array_equal(pis1, pis2) and n1 == n2 and nt1 == nt2
        assert np.array_equal(f1, f2) and np.array_equal(l1, l2)
        assert pis1.sum() == n1


def test_cal_prior_against_reference(O, prior_pdf_golden):
    g = prior_pdf_golden
    pri_u = O.SMCSettings().priors
    with np.errstate(all="ignore"):
        got = O.cal_prior(g["theta_uniform"], pri_u)
    assert np.array_equal(np.isnan(got), np.isnan(g["pdf_uniform"]))
    assert np.array_equal(np.nan_to_num(got), np.nan_to_num(g["pdf_uniform"]))
    pri_m = {"Vmax": {"dist": "normal", "mu": 1.0, "sigma": 0.1}, "Km": {"dist": "normal", "mu": 0.0, "sigma": 5.0},
             "sigma": {"dist": "uniform", "low": 0, "high": 10}}
    got2 = O.cal_prior(g["theta_mixed"], pri_m)
    assert np.allclose(got2, g["pdf_mixed"], rtol=1e-15, atol=0)


def test_ess_candidates_match_search(O):
    s = O.SMCSettings()
    rs = np.random.RandomState(2)
    lk = -np.abs(rs.standard_normal(500)) * 400
    es = O.ess_search(lk, 0.0, s)
    gms, gammas, _ = O.ess_candidates(0.0, s)
    assert es["gm"] == gms[es["iters"] - 1] and es["gamma_new"] == gammas[es["iters"] - 1]
