"""Ragged / degenerate data shapes of the Michaelis-Menten path against the CPU checker: 1..7 experiments, 1..50 data times
(a single time means t_span of zero length: solve_ivp returns without a step), output grids that do not start at 0,
S0 = 0, fixed sigma; a repeated output time must be refused as solve_ivp refuses it (ivp.py:606-609)."""
import numpy as np
import pytest

TOL = 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_random_data_shapes_vs_checker(pkg, O, seed):
    rs = np.random.RandomState(100 + seed)
    n_ex = int(rs.randint(1, 8))
    n_t = int(rs.choice([1, 2, 3, 7, 40, 50]))
    t = np.sort(rs.uniform(0, 12, (n_ex, n_t)), axis=1)
    if seed % 2 == 0:
        t[:, 0] = 0.0
    S0 = rs.uniform(0.05, 3.0, n_ex)
    if seed == 3:
        S0[0] = 0.0
    P_obs = rs.uniform(0, 2, (n_ex, n_t))
    est_sigma = seed % 3 != 0
    data = O.MMData(t=t, P_obs=P_obs, S0=S0)
    n = 3000
    th = np.column_stack([rs.uniform(0.05, 10, n), rs.uniform(0.05, 10, n), rs.uniform(0.01, 5, n)])
    ref = O.mm_loglik_batch(th, data, est_sigma=est_sigma, sigma_fixed=0.7)[0]
    with pkg.HipEngine(n, 3, device=0) as eng:
        eng.set_model_mm(t, P_obs, S0, est_sigma=est_sigma, sigma_fixed=0.7)
        eng.upload_particles(pkg.SMC_SET_PRED, th)
        info = eng.loglik(pkg.SMC_SET_PRED)
        lk = eng.download_lk(pkg.SMC_SET_PRED)
    assert info["n_failed"] == 0
    assert np.max(np.abs(lk - ref) / np.maximum(1.0, np.abs(ref))) < TOL, (n_ex, n_t)


@pytest.mark.gpu
def test_repeated_output_time_is_refused(pkg):
    t = np.array([[0.0, 1.0, 1.0, 2.0]])
    with pkg.HipEngine(8, 3, device=0) as eng:
        with pytest.raises(pkg.SmcError, match="strictly increasing"):
            eng.set_model_mm(t, np.zeros_like(t), np.array([1.0]))


@pytest.mark.gpu
def test_weights_with_minus_infinity_and_huge_spread(pkg):
    """sigma <= 0 gives logL = -inf (Micmem_likelihood.py:53-54): such particles must get weight exactly 0 in the fused
    ESS sums, as exp((lk - max) * gm) does in NumPy; a spread of 1e6 in logL must not overflow anything."""
    n = 10_000
    rs = np.random.RandomState(5)
    lk = rs.standard_normal(n) * 1e3
    lk[::7] = -np.inf
    lk[3] = 1e6
    t = np.linspace(0, 1, 4)[None, :]
    with pkg.HipEngine(n, 3, device=0) as eng:
        eng.set_model_mm(t, np.zeros_like(t), np.array([1.0]))
        eng.upload_lk(pkg.SMC_SET_PRED, lk)
        mx = eng.max_lk_local()
        assert mx == 1e6
        gms = [1.0, 1e-3, 1e-6, 1e-9]
        sw, sw2 = eng.ess_partials(mx, gms)
    for k, gm in enumerate(gms):
        with np.errstate(under="ignore"):
            w = np.exp((lk - mx) * gm)
        assert np.isfinite(sw[k]) and abs(sw[k] / w.sum() - 1) < 1e-12 and abs(sw2[k] / (w * w).sum() - 1) < 1e-12


@pytest.mark.gpu
def test_ess_searches_on_degenerate_likelihoods(pkg):
    """Both gamma searches at the two ends: equal likelihoods (ESS = 1 for every gamma: the whole remaining increment is
    taken, gamma = 1.0 exactly) and one particle towering above the rest (ESS ~ 1/N for every increment worth
    taking: the back-off runs through its 80 candidates and warns, main:141-144; the bisection ends on its smallest
    bracket and warns likewise)."""
    n = 4096
    t = np.linspace(0, 1, 4)[None, :]
    for how in ("backoff", "bisection"):
        s = pkg.SMCSettings(n_particle=n, ess_search=how)
        with pkg.HipEngine(n, 3, device=0) as eng:
            eng.set_model_mm(t, np.zeros_like(t), np.array([1.0]))
            eng.upload_lk(pkg.SMC_SET_PRED, np.full(n, -12.5))
            es = pkg.ess_search(eng, pkg.SingleComm(), 0.25, s)
            assert es["gamma_new"] == 1.0 and not es["warning"] and abs(es["ess"] - 1.0) < 1e-12
            lk = np.full(n, -1e15)          # even the 80th back-off candidate (0.75 * 0.7**79 ~ 4e-13) leaves weight exp(-430)
            lk[17] = 0.0
            eng.upload_lk(pkg.SMC_SET_PRED, lk)
            es = pkg.ess_search(eng, pkg.SingleComm(), 0.25, s)
            assert es["warning"] and 0.25 < es["gamma_new"] < 0.25 + 1e-8 and es["ess"] < 0.5
