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
def test_largest_data_set_vs_checker(pkg, O):
    """16 experiments x 256 data times: the (time, observation) table of the solve kernel takes 66 KB of LDS, above the
    default 64 KB limit of a launch."""
    rs = np.random.RandomState(77)
    n_ex, n_t, n = 16, 256, 192
    t = np.sort(rs.uniform(0, 12, (n_ex, n_t)), axis=1)
    t[::2, 0] = 0.0
    S0 = rs.uniform(0.05, 3.0, n_ex)
    P_obs = rs.uniform(0, 2, (n_ex, n_t))
    data = O.MMData(t=t, P_obs=P_obs, S0=S0)
    th = np.column_stack([rs.uniform(0.05, 10, n), rs.uniform(0.05, 10, n), rs.uniform(0.01, 5, n)])
    ref = O.mm_loglik_batch(th, data, est_sigma=True, sigma_fixed=0.7)[0]
    with pkg.HipEngine(n, 3, device=0) as eng:
        eng.set_model_mm(t, P_obs, S0)
        lk, pred, info = eng.loglik_host(th, want_pred=True)
    assert info["n_failed"] == 0 and pred.shape == (n, n_ex, n_t) and np.isfinite(pred).all()
    assert np.max(np.abs(lk - ref) / np.maximum(1.0, np.abs(ref))) < TOL


@pytest.mark.gpu
def test_stiff_band_parity_and_its_tolerance(pkg, O):
    """Km log-uniform over 1e-3 .. 10: a quarter of the particles sit in the band where RK45 runs on its stability limit
    (thousands of attempts per solve).
    * Default (fast) mode: the device's inverse fifth root (<= 1.5 ulp about x^(-1/5)) is not libm's pow(x, -0.2) - whose
      DOUBLE exponent is -(1/5 + 1.1e-17), more than an ulp away from the fifth root for small error norms: a last-bit
      difference in one step size can flip one accept / reject decision of the step controller, after which the two
      integrations follow different, equally valid step sequences and logL differs at 1e-8 .. 1e-6 instead of 1e-12 - far
      inside the rtol = 1e-3 of the solve itself.  Stated tolerance: every particle within 1e-6, at most 0.2 % of them
      beyond 1e-9, attempt totals within 1e-5 (a randomised soak over 160 shapes and sizes gave 2.8e-7 at worst).
    * Parity mode (smc_set_exact_pow, what run_smc(rng="numpy") and the drop-in sim_particle use): the power is finished to
      the correctly rounded pow(x, -0.2) - libm's value except for ~8 of 10^4 arguments where glibc itself is not
      correctly rounded (tests/test_pow_fifth_exact.py).  EVERY particle within 1e-9 and the attempt totals EQUAL."""
    rs = np.random.RandomState(2024)
    n = 20000
    t = np.tile(np.linspace(0, 10, 40), (6, 1))
    S0 = np.array([2, 0.1, 0.25, 0.5, 1, 2.0])
    P_obs = rs.uniform(0, 2, (6, 40))
    th = np.column_stack([rs.uniform(0.05, 10, n), 10.0 ** rs.uniform(-3, 1, n), rs.uniform(0.01, 5, n)])
    ref, _, iref = O.mm_loglik_batch(th, O.MMData(t=t, P_obs=P_obs, S0=S0))
    with pkg.HipEngine(n, 3, device=0) as eng:
        eng.set_model_mm(t, P_obs, S0)
        lk, _, info = eng.loglik_host(th)
        eng.set_exact_pow(True)
        lk_x, _, info_x = eng.loglik_host(th)
    err = np.abs(lk - ref) / np.maximum(1.0, np.abs(ref))
    assert info["n_failed"] == 0 and err.max() < 1e-6, err.max()
    assert (err > 1e-9).sum() <= 2e-3 * n, int((err > 1e-9).sum())
    assert abs(info["rk_attempts"] - iref["n_attempts"]) <= 1e-5 * iref["n_attempts"]
    err_x = np.abs(lk_x - ref) / np.maximum(1.0, np.abs(ref))
    print(f"stiff band, {n} particles: fast mode max err {err.max():.2e}, {int((err > 1e-9).sum())} beyond 1e-9, attempts "
          f"{info['rk_attempts']} vs {iref['n_attempts']}; parity mode max err {err_x.max():.2e}, "
          f"{int((err_x > 1e-9).sum())} beyond 1e-9, attempts {info_x['rk_attempts']}")
    assert info_x["n_failed"] == 0 and err_x.max() < 1e-9, (err_x.max(), int((err_x > 1e-9).sum()))
    assert info_x["rk_attempts"] == iref["n_attempts"]


@pytest.mark.gpu
def test_data_times_outside_the_supported_range_are_refused(pkg):
    t = np.array([[0.0, 1e-200, 1.0, 2.0]])
    with pkg.HipEngine(8, 3, device=0) as eng:
        with pytest.raises(pkg.SmcError, match="data times"):
            eng.set_model_mm(t, np.zeros_like(t), np.array([1.0]))


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


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 130, 257, 1000, 4097])
def test_fused_iterations_and_whole_runs_at_odd_population_sizes(pkg, O, data, n):
    """The fused Metropolis iteration (smc_mh_iteration_device_rng: carried moments, per-block moment rows, the factor
    kernel) and the on-device reductions of a whole device-RNG run at population sizes around the wave / block / tile
    boundaries, down to a single particle (cov_m = 0: the proposal is the particle itself and must be accepted).  After
    every iteration: p_filt differs from its predecessor in exactly accepted_now rows (or fewer when a proposal equals
    its particle), the flags count accepted_ever, and the next iteration's cov_m is NumPy's covariance of what is there."""
    s = pkg.SMCSettings(n_particle=n)
    w_cov = s.w_cov()
    rs = np.random.RandomState(n)
    th = np.array([1.2254, 0.5218, 0.02048]) + rs.standard_normal((n, 3)) * np.array([0.025, 0.0295, 0.00094])
    with pkg.HipEngine(n, 3, device=0) as eng:
        eng.set_model_mm(data.t, data.P_obs, data.S0)
        eng.set_prior(s.priors)
        eng.upload_particles(pkg.SMC_SET_PRED, th)
        eng.loglik(pkg.SMC_SET_PRED)
        lk_ref = O.mm_loglik_batch(th, data)[0]
        assert np.max(np.abs(eng.download_lk(pkg.SMC_SET_PRED) - lk_ref) / np.maximum(1.0, np.abs(lk_ref))) < TOL
        eng.upload_particles(pkg.SMC_SET_FILT, th)
        eng.upload_lk(pkg.SMC_SET_FILT, eng.download_lk(pkg.SMC_SET_PRED))
        eng.reset_accept_flags()
        cur = th
        for j in range(3):
            out = eng.mh_iteration_device_rng(1.0, 1.0, w_cov, 11, (3 << 16) | j, 0)
            ref = np.cov(cur.T, bias=True) * w_cov if n > 1 else np.zeros((3, 3))
            assert np.abs(out["cov_m"] - ref).max() <= 1e-10 * max(np.abs(ref).max(), 1e-300), (j, out["cov_m"], ref)
            new = eng.download_particles(pkg.SMC_SET_FILT)
            moved = int(np.any(new != cur, axis=1).sum())
            assert moved <= out["accepted_now"] <= n and out["n_failed"] == 0
            if n > 1:
                assert moved == out["accepted_now"]
            assert int(eng.download_accept_flags().sum()) == out["accepted_ever"]
            lk_new = O.mm_loglik_batch(new, data)[0]
            assert np.max(np.abs(eng.download_lk(pkg.SMC_SET_FILT) - lk_new) / np.maximum(1.0, np.abs(lk_new))) < TOL
            cur = new
        if n >= 63:
            run = pkg.run_smc(eng, s, rng="device", verbose=False, seed_device=n)
            assert run["gamma"] == 1.0 and all(r["n_offspring"] in (n - 1, n) for r in run["records"])
            lk_end = O.mm_loglik_batch(run["p_pred"], data)[0]
            assert np.max(np.abs(run["lk"] - lk_end) / np.maximum(1.0, np.abs(lk_end))) < TOL
