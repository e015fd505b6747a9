"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the pinned CPU oracle
on the same seeded inputs, against the golden fixtures, and - at BASELINE.json's full size (1e6
particles) - through size-independent properties.

Floating-point tolerances (all arithmetic is float64):
  TOL_LOGL   1e-9  relative on a particle's log-likelihood.  Sources of difference vs the oracle:
             FMA contraction inside the RK45 stage sums, device exp/pow/log (<= 1-2 ulp) vs glibc,
             amplified by ~116 adaptive steps per solve.  Observed: ~1e-12.
  TOL_SUM    1e-12 relative on reductions over N (different summation order).
Integer / index work (offspring counts, ancestor indices, accept flags and counts) is compared
exactly.
"""
import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu

TOL_LOGL = 1e-9
TOL_SUM = 1e-12


@pytest.fixture(scope="module")
def settings(pkg):
    return pkg.SMCSettings()


def make_engine(pkg, data, n, priors=None, **kw):
    eng = pkg.HipEngine(n, 3, device=0, **kw)
    eng.set_model_mm(data.t, data.P_obs, data.S0)
    eng.set_prior(priors or pkg.SMCSettings().priors)
    return eng


def mixed_particles(n, seed=0):
    rs = np.random.RandomState(seed)
    th = np.array([1.2254, 0.5218, 0.02048]) + rs.standard_normal((n, 3)) * np.array([0.025, 0.0295, 0.00094])
    k = n // 2
    th[:k] = rs.uniform(0, 10, size=(k, 3))
    return th


# ---------------------------------------------------------------------------------------------------
# A2: likelihood
# ---------------------------------------------------------------------------------------------------
def test_loglik_known_answers_from_reference(pkg, data, known_answers):
    """Directly against values the reference's log_likelihood_mm_multi produced (golden fixture)."""
    th = known_answers["theta"]
    with make_engine(pkg, data, len(th)) as eng:
        lk, pred, info = eng.loglik_host(th, want_pred=True)
    assert info["n_failed"] == 0
    assert relerr(lk, known_answers["logL"]).max() < TOL_LOGL
    assert np.abs(pred - known_answers["pred"]).max() < 1e-9


def test_loglik_first_sweep_of_reference_run(pkg, data, golden_run):
    th, ref = golden_run["sweeps_theta"][0], golden_run["sweeps_llk"][0]
    with make_engine(pkg, data, len(th)) as eng:
        eng.upload_particles(pkg.SMC_SET_PRED, th)
        info = eng.loglik(pkg.SMC_SET_PRED)
        lk = eng.download_lk(pkg.SMC_SET_PRED)
    assert info["n_failed"] == 0
    assert relerr(lk, ref).max() < TOL_LOGL


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 20000])
def test_loglik_vs_oracle(pkg, O, data, n):
    th = mixed_particles(n, seed=n)
    ref, _, oinfo = O.mm_loglik_batch(th, data)
    with make_engine(pkg, data, n) as eng:
        eng.upload_particles(pkg.SMC_SET_PRED, th)
        info = eng.loglik(pkg.SMC_SET_PRED)
        lk = eng.download_lk(pkg.SMC_SET_PRED)
    assert info["n_failed"] == 0 and oinfo["n_failed"] == 0
    assert relerr(lk, ref).max() < TOL_LOGL
    # same adaptive step sequence: the device-counted attempts equal the oracle's
    assert info["rk_attempts"] == oinfo["n_attempts"]


def test_loglik_corner_cases(pkg, O, data):
    th = np.array([[1.0, 1.0, 0.0], [1.0, 1.0, -3.0], [0.0, 1.0, 1.0], [10.0, 10.0, 10.0], [1e-6, 1e-6, 1e-3],
                   [5.0, 1e-8, 0.1], [9.99, 1e-3, 0.5], [1.2, 0.5, 1e-6], [1.0, 1.0, np.nan]])
    ref, _, _ = O.mm_loglik_batch(th, data)
    with make_engine(pkg, data, len(th)) as eng:
        lk, _, info = eng.loglik_host(th)
    assert np.isneginf(lk[0]) and np.isneginf(lk[1])          # sigma <= 0 -> -inf (Micmem_likelihood.py:53-54)
    assert np.isnan(lk[-1]) and np.isnan(ref[-1])
    assert relerr(lk[2:-1], ref[2:-1]).max() < TOL_LOGL


def test_loglik_fixed_sigma(pkg, O, data):
    th = mixed_particles(500, seed=9)
    ref, _, _ = O.mm_loglik_batch(th, data, est_sigma=False, sigma_fixed=0.05)
    with pkg.HipEngine(500, 3) as eng:
        eng.set_model_mm(data.t, data.P_obs, data.S0, est_sigma=False, sigma_fixed=0.05)
        lk, _, _ = eng.loglik_host(th)
    assert relerr(lk, ref).max() < TOL_LOGL


def test_loglik_ragged_data_shapes(pkg, O):
    """Other data-set shapes than 6x40 (n_ex, n_t are run-time parameters of the kernel)."""
    rs = np.random.RandomState(4)
    for n_ex, n_t in [(1, 2), (3, 17), (16, 40), (2, 256)]:
        t = np.sort(rs.uniform(0, 8, size=(n_ex, n_t)), axis=1)
        t[:, 0] = 0.0
        P = rs.uniform(0, 1, size=(n_ex, n_t))
        S0 = rs.uniform(0.1, 3, size=n_ex)
        d = O.MMData(t, P, S0)
        th = rs.uniform(0.05, 5, size=(130, 3))
        ref, _, _ = O.mm_loglik_batch(th, d)
        with pkg.HipEngine(130, 3) as eng:
            eng.set_model_mm(t, P, S0)
            lk, _, info = eng.loglik_host(th)
        assert info["n_failed"] == 0
        assert relerr(lk, ref).max() < TOL_LOGL, (n_ex, n_t)


# ---------------------------------------------------------------------------------------------------
# A3/A4: weights, ESS, max
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 100, 1000, 100003])
def test_max_and_ess_partials(pkg, O, data, n):
    rs = np.random.RandomState(n)
    lk = -np.abs(rs.standard_normal(n)) * 300 + 100
    s = O.SMCSettings()
    gms, gammas, _ = O.ess_candidates(0.1, s)
    with make_engine(pkg, data, n) as eng:
        eng.upload_lk(pkg.SMC_SET_PRED, lk)
        mx = eng.max_lk_local()
        assert mx == lk.max()
        for k0, k1 in [(0, 16), (16, 21), (40, 41), (50, 58)]:
            sw, sw2 = eng.ess_partials(mx, gms[k0:k1])
            for i, gm in enumerate(gms[k0:k1]):
                w = np.exp((lk - mx) * gm)
                assert abs(sw[i] - w.sum()) <= TOL_SUM * w.sum()
                assert abs(sw2[i] - (w * w).sum()) <= TOL_SUM * (w * w).sum()


def test_ess_search_matches_oracle(pkg, O, data, golden_run):
    lk = golden_run["sweeps_llk"][0]
    s = O.SMCSettings()
    ref = O.ess_search(lk, 0.0, s)
    with make_engine(pkg, data, len(lk)) as eng:
        eng.upload_lk(pkg.SMC_SET_PRED, lk)
        es = pkg.ess_search(eng, pkg.SingleComm(), 0.0, pkg.SMCSettings())
    assert es["gamma_new"] == ref["gamma_new"] == golden_run["sched_gamma"][0]
    assert es["iters"] == ref["iters"] and es["gm"] == ref["gm"]
    assert abs(es["ess"] - ref["ess"]) < 1e-12
    assert abs(es["sum_weight"] - ref["sum_weight"]) <= TOL_SUM * ref["sum_weight"]


# ---------------------------------------------------------------------------------------------------
# A5: resampling
# ---------------------------------------------------------------------------------------------------
def _resample_case(pkg, O, data, n, gm, seed, first_step=True, spread=30.0):
    rs = np.random.RandomState(seed)
    lk = rs.standard_normal(n) * spread
    p_pred = rs.standard_normal((n, 3))
    u = rs.rand()
    mx = lk.max()
    w = np.exp((lk - mx) * gm)
    sum_w = np.sum(w)
    w = w / sum_w
    p_filt = np.zeros((n, 3))
    lk1 = np.zeros(n)
    p_is, n_written, n_tmp = O.resample(w, u, p_pred, lk, p_filt, lk1)
    with make_engine(pkg, data, n) as eng:
        eng.upload_particles(pkg.SMC_SET_PRED, p_pred)
        eng.upload_lk(pkg.SMC_SET_PRED, lk)
        s = pkg.SMCSettings(n_particle=n)
        es = {"max_lk": mx, "gm": gm, "sum_weight": float(sum_w)}
        out = pkg.resample(eng, pkg.SingleComm(), es, u, s, first_step=first_step)
        got_is = eng.download_offspring()
        got_f = eng.download_particles(pkg.SMC_SET_FILT)
        got_l = eng.download_lk(pkg.SMC_SET_FILT)
    return (p_is, n_written, p_filt, lk1), (got_is, out["n_offspring"], got_f, got_l)


@pytest.mark.parametrize("n,gm", [(1, 1.0), (2, 0.3), (255, 0.05), (1000, 0.002), (1024, 0.05), (1025, 1.0),
                                  (4097, 0.01), (50000, 0.004)])
def test_resample_vs_oracle(pkg, O, data, n, gm):
    ref, got = _resample_case(pkg, O, data, n, gm, seed=n)
    assert np.array_equal(got[0], ref[0])          # offspring counts: exact
    assert got[1] == ref[1] == n
    assert np.array_equal(got[2], ref[2])          # gathered rows: exact copies in ancestor order
    assert np.array_equal(got[3], ref[3])


@pytest.mark.parametrize("pattern", ["equal", "one_hot", "two_spikes", "last_only", "underflow_tail"])
@pytest.mark.parametrize("n", [7, 1024, 3001])
def test_resample_degenerate_weights(pkg, O, data, n, pattern):
    """Weight patterns at the edges of the resampler: all equal (every residual is a rounding residue), all mass on
    one particle (one ancestor fills the whole output), two far-apart spikes, all mass on the last particle (the
    running sum stays 0 until the end), weights that underflow to exactly 0 - offspring and gathered rows must equal
    the sequential loop of Micmem_SMC_main.py:147-184 exactly."""
    rs = np.random.RandomState(n)
    lk = np.zeros(n)
    if pattern == "one_hot":
        lk[:] = -1e6
        lk[n // 3] = 0.0
    elif pattern == "two_spikes":
        lk[:] = -50.0
        lk[1], lk[n - 2] = 0.0, -0.3
    elif pattern == "last_only":
        lk[:] = -1e6
        lk[n - 1] = 0.0
    elif pattern == "underflow_tail":
        lk = -np.arange(n, dtype=float) * 30.0
    p_pred = rs.standard_normal((n, 3))
    u, gm = 0.999, 1.0
    mx = lk.max()
    w = np.exp((lk - mx) * gm)
    sum_w = np.sum(w)
    p_filt, lk1 = np.zeros((n, 3)), np.zeros(n)
    p_is, n_written, _ = O.resample(w / sum_w, u, p_pred, lk, p_filt, lk1)
    with make_engine(pkg, data, n) as eng:
        eng.upload_particles(pkg.SMC_SET_PRED, p_pred)
        eng.upload_lk(pkg.SMC_SET_PRED, lk)
        es = {"max_lk": mx, "gm": gm, "sum_weight": float(sum_w)}
        out = pkg.resample(eng, pkg.SingleComm(), es, u, pkg.SMCSettings(n_particle=n), first_step=True)
        assert np.array_equal(eng.download_offspring(), p_is)
        assert out["n_offspring"] == n_written
        assert np.array_equal(eng.download_particles(pkg.SMC_SET_FILT)[:n_written], p_filt[:n_written])
        assert np.array_equal(eng.download_lk(pkg.SMC_SET_FILT)[:n_written], lk1[:n_written])


def test_resample_golden_first_step(pkg, O, data, golden_run):
    """The reference run's first tempering step: weights from its first sweep, wrand from its seed."""
    g = golden_run
    s = O.SMCSettings()
    np.random.seed(int(g["seed"]))
    p_pred = O.sample_prior(s.priors, 1000)
    assert np.array_equal(p_pred, g["sweeps_theta"][0])
    lk = g["sweeps_llk"][0]
    es = O.ess_search(lk, 0.0, s)
    u = np.random.rand()
    p_filt, lk1 = np.zeros((1000, 3)), np.zeros(1000)
    p_is, _, _ = O.resample(es["p_weight"], u, p_pred, lk, p_filt, lk1)
    with make_engine(pkg, data, 1000) as eng:
        eng.upload_particles(pkg.SMC_SET_PRED, p_pred)
        eng.upload_lk(pkg.SMC_SET_PRED, lk)
        des = pkg.ess_search(eng, pkg.SingleComm(), 0.0, pkg.SMCSettings())
        pkg.resample(eng, pkg.SingleComm(), des, u, pkg.SMCSettings(), first_step=True)
        assert np.array_equal(eng.download_offspring(), p_is)
        assert np.array_equal(eng.download_particles(pkg.SMC_SET_FILT), p_filt)


def test_resample_full_size_properties(pkg, data):
    """N = 1e6 (BASELINE config 2): offspring sum to N, every count >= trunc(N w), and the output is
    exactly np.repeat(ancestors, counts) - pure index work, compared bit for bit."""
    n = 1_000_000
    rs = np.random.RandomState(123)
    lk = rs.standard_normal(n) * 5
    p_pred = rs.standard_normal((n, 3))
    mx, gm = lk.max(), 0.37
    w = np.exp((lk - mx) * gm)
    sum_w = float(np.sum(w))
    with make_engine(pkg, data, n) as eng:
        eng.upload_particles(pkg.SMC_SET_PRED, p_pred)
        eng.upload_lk(pkg.SMC_SET_PRED, lk)
        s = pkg.SMCSettings(n_particle=n)
        out = pkg.resample(eng, pkg.SingleComm(), {"max_lk": mx, "gm": gm, "sum_weight": sum_w}, 0.4321, s, True)
        p_is = eng.download_offspring()
        f = eng.download_particles(pkg.SMC_SET_FILT)
        l = eng.download_lk(pkg.SMC_SET_FILT)
    assert out["n_offspring"] == n == p_is.sum()
    base = np.trunc(w / sum_w * n).astype(np.int64)
    extra = p_is - base
    assert extra.min() >= 0 and extra.max() <= 1
    anc = np.repeat(np.arange(n), p_is)
    assert np.array_equal(f, p_pred[anc]) and np.array_equal(l, lk[anc])


# ---------------------------------------------------------------------------------------------------
# A6: moments
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [2, 1000, 65537])
def test_proposal_cov_vs_numpy(pkg, data, n):
    rs = np.random.RandomState(n)
    x = np.array([1.2, 0.5, 0.02]) + rs.standard_normal((n, 3)) * np.array([0.03, 0.03, 0.001])
    s = pkg.SMCSettings(n_particle=n)
    with make_engine(pkg, data, n) as eng:
        eng.upload_particles(pkg.SMC_SET_FILT, x)
        cov = pkg.proposal_cov(eng, pkg.SingleComm(), s, s.w_cov())
    ref = np.cov(x.T, bias=True) * s.w_cov()
    assert np.abs(cov - ref).max() <= 1e-11 * np.abs(ref).max()


# ---------------------------------------------------------------------------------------------------
# A7-A9: fused MH iteration
# ---------------------------------------------------------------------------------------------------
def _mh_reference(O, data, priors, p_filt, lk1, noise, rr, gamma, ratio):
    p_pred = p_filt + noise * ratio
    p0 = np.int32(O.cal_prior(p_pred, priors) > 0)
    p_pred = p_pred * p0[:, None] + p_filt * (1.0 - p0[:, None])
    lk2, _, _ = O.mm_loglik_batch(p_pred, data)
    with np.errstate(over="ignore"):
        pp = np.exp((lk2 - lk1) * gamma) * p0
    r = np.int32(pp >= rr)
    return p_pred, p0, lk2, r, p_pred * r[:, None] + p_filt * (1.0 - r[:, None]), lk2 * r + lk1 * (1.0 - r)


@pytest.mark.parametrize("n,gamma,ratio", [(1000, 0.0023, 1.0), (1000, 1.0, 0.5), (4099, 0.2, 1.0)])
def test_mh_step_host_rng_vs_oracle(pkg, O, data, n, gamma, ratio):
    rs = np.random.RandomState(n + int(gamma * 1000))
    p_filt = mixed_particles(n, seed=n + 1)
    lk1, _, _ = O.mm_loglik_batch(p_filt, data)
    cov = np.cov(p_filt.T, bias=True) * pkg.SMCSettings().w_cov()
    noise = rs.multivariate_normal(np.zeros(3), cov, n)
    rr = rs.uniform(0, 1, n)
    priors = pkg.SMCSettings().priors
    prop, p0, lk2, r, f_ref, l_ref = _mh_reference(O, data, priors, p_filt, lk1, noise, rr, gamma, ratio)
    with make_engine(pkg, data, n) as eng:
        eng.set_debug_capture(True)
        eng.upload_particles(pkg.SMC_SET_FILT, p_filt)
        eng.upload_lk(pkg.SMC_SET_FILT, lk1)
        out = eng.mh_step_host_rng(gamma, ratio, noise, rr)
        d_prop, d_lk2, d_p0, d_r = eng.download_debug_proposals()
        f = eng.download_particles(pkg.SMC_SET_FILT)
        l = eng.download_lk(pkg.SMC_SET_FILT)
        flags = eng.download_accept_flags()
    assert np.array_equal(d_prop, prop)                 # proposals: same separately rounded mul/add as NumPy
    assert np.array_equal(d_p0, p0)
    assert relerr(d_lk2, lk2).max() < TOL_LOGL
    assert np.array_equal(d_r, r), "accept decisions differ"
    assert np.array_equal(f, f_ref)
    assert relerr(l, l_ref).max() < TOL_LOGL
    assert out["accepted_now"] == int(r.sum()) == out["accepted_ever"] == int(flags.sum())
    assert out["n_failed"] == 0


def test_mh_prior_mask_mixed_priors(pkg, O, data):
    """normal + uniform priors: the support mask is cal_prior(...) > 0 (Micmem_SMC_main.py:225-226)."""
    n = 777
    priors = {"Vmax": {"dist": "normal", "mu": 1.0, "sigma": 0.1}, "Km": {"dist": "normal", "mu": 0.0, "sigma": 5.0},
              "sigma": {"dist": "uniform", "low": 0.01, "high": 0.03}}
    rs = np.random.RandomState(8)
    p_filt = np.array([1.2, 0.5, 0.02]) + rs.standard_normal((n, 3)) * np.array([0.02, 0.02, 0.0005])
    p_filt[:, 2] = np.clip(p_filt[:, 2], 0.0101, 0.0299)
    lk1, _, _ = O.mm_loglik_batch(p_filt, data)
    noise = rs.standard_normal((n, 3)) * np.array([0.05, 0.05, 0.01])
    noise[::50, 0] = 5.0   # |z| = 42 sigma: norm.pdf underflows to 0 -> out of support
    rr = rs.uniform(0, 1, n)
    prop, p0, lk2, r, f_ref, l_ref = _mh_reference(O, data, priors, p_filt, lk1, noise, rr, 1.0, 1.0)
    assert 0 < p0.sum() < n
    with make_engine(pkg, data, n, priors=priors) as eng:
        eng.set_debug_capture(True)
        eng.upload_particles(pkg.SMC_SET_FILT, p_filt)
        eng.upload_lk(pkg.SMC_SET_FILT, lk1)
        eng.mh_step_host_rng(1.0, 1.0, noise, rr)
        d_prop, d_lk2, d_p0, d_r = eng.download_debug_proposals()
        f = eng.download_particles(pkg.SMC_SET_FILT)
    assert np.array_equal(d_p0, p0) and np.array_equal(d_r, r) and np.array_equal(f, f_ref)


def test_accept_flags_accumulate(pkg, O, data):
    n = 512
    rs = np.random.RandomState(1)
    p_filt = mixed_particles(n, seed=5)
    lk1, _, _ = O.mm_loglik_batch(p_filt, data)
    with make_engine(pkg, data, n) as eng:
        eng.upload_particles(pkg.SMC_SET_FILT, p_filt)
        eng.upload_lk(pkg.SMC_SET_FILT, lk1)
        ever = np.zeros(n, dtype=np.uint8)
        for it in range(3):
            noise = rs.standard_normal((n, 3)) * 0.01
            rr = rs.uniform(0, 1, n)
            before = eng.download_particles(pkg.SMC_SET_FILT)
            out = eng.mh_step_host_rng(0.5, 1.0, noise, rr)
            after = eng.download_particles(pkg.SMC_SET_FILT)
            ever |= np.any(before != after, axis=1).astype(np.uint8)
            assert out["accepted_ever"] >= int(ever.sum())
            assert out["accepted_ever"] == int(eng.download_accept_flags().sum())
        eng.reset_accept_flags()
        assert eng.download_accept_flags().sum() == 0


# ---------------------------------------------------------------------------------------------------
# whole loop
# ---------------------------------------------------------------------------------------------------
def test_full_run_numpy_rng_matches_reference(pkg, data, golden_run):
    """Config 1 (N=1000, seed 20250205) end to end on the GPU in host-RNG parity mode against the
    reference run itself: gamma schedule bit-exact, ESS, accept counts and MH lengths exact,
    posterior particles within 1e-9."""
    g = golden_run
    s = pkg.SMCSettings()
    with make_engine(pkg, data, 1000) as eng:
        out = pkg.run_smc(eng, s, rng="numpy", verbose=False)
    rec = out["records"]
    assert out["step"] == int(g["final_step"])
    assert np.array_equal([r["gamma_new"] for r in rec], g["sched_gamma"])
    assert np.allclose([r["ess"] for r in rec], g["sched_ess"], rtol=1e-10, atol=0)
    assert np.array_equal([r["n_accept"] for r in rec], g["sched_accept"])
    assert np.array_equal([r["last_j"] for r in rec], g["sched_last_j"])
    assert np.allclose([r["max_lk"] for r in rec], g["sched_maxlk"], rtol=TOL_LOGL, atol=0)
    assert np.abs(out["p_pred"] - g["final_p_pred"]).max() < 1e-9
    assert relerr(out["lk"], g["final_lk"]).max() < TOL_LOGL
    assert np.random.rand() == float(g["next_rand_after_run"])
    assert abs(out["p_pred"].mean(axis=0) - g["final_p_pred"].mean(axis=0)).max() < 1e-12
    assert out["stats"]["mutation_sweeps"] == g["sweeps_theta"].shape[0] - 1


def test_full_run_numpy_rng_at_20000_particles_matches_oracle(pkg, O, data):
    """Parity mode with enough particles for everything the scheduler does to a sweep (stiff and solo lists, cost-ordered
    in-phase hand-out of the heterogeneous sweeps, early rejection): a complete run on the NumPy stream against the checker's
    run on the same stream - tempering schedule, accept counts and Metropolis lengths exact, particles and log-evidence to 1e-9."""
    n = 20000
    so = O.SMCSettings()
    so.n_particle = n
    o = O.run_smc(data, so, seed=13, n_threads=0, record_mh=False)
    s = pkg.SMCSettings(n_particle=n, seed=13)
    with make_engine(pkg, data, n) as eng:
        out = pkg.run_smc(eng, s, rng="numpy", verbose=False)
    assert [r["gamma_new"] for r in out["records"]] == [r.gamma_new for r in o["records"]]
    assert [r["n_accept"] for r in out["records"]] == [r.n_accept for r in o["records"]]
    assert [r["last_j"] for r in out["records"]] == [r.last_j for r in o["records"]]
    assert abs(out["logZ"] - o["logZ"]) < 1e-9 * abs(o["logZ"])
    assert np.abs(out["p_pred"] - o["p_pred"]).max() < 1e-9


def test_full_run_logz_matches_oracle(pkg, O, data):
    o = O.run_smc(data, O.SMCSettings(), seed=7, n_threads=0, record_mh=False)
    s = pkg.SMCSettings(seed=7)
    with make_engine(pkg, data, 1000) as eng:
        out = pkg.run_smc(eng, s, rng="numpy", verbose=False)
    assert abs(out["logZ"] - o["logZ"]) < 1e-9 * abs(o["logZ"])
    assert np.abs(out["p_pred"] - o["p_pred"]).max() < 1e-9


@pytest.mark.parametrize("mode", ["ratio_mask", "ratio"])
def test_full_run_prior_density_ratio_modes(pkg, O, data, mode):
    """normal / mixed priors with the density ratio p0_2/p0_1 in the acceptance
    (SMC_methanation_main.py:320-374; dead branches in the reference, so the pin is the oracle's restatement of those
    lines on the same NumPy stream).  "ratio_mask": normal + uniform priors, proposals outside the support reset;
    "ratio": normal priors with one parameter left out of the density (as sigma is in methanation_functions.py:132-138),
    no reset."""
    if mode == "ratio_mask":
        priors = {"Vmax": {"dist": "normal", "mu": 3.0, "sigma": 0.7}, "Km": {"dist": "uniform", "low": 0, "high": 10},
                  "sigma": {"dist": "uniform", "low": 0, "high": 10}}
    else:   # everything stays > 4 prior sigmas away from 0 (sigma <= 0 gives logL = -inf and then -inf * 0 = NaN, main:240)
        # without the reset a proposal may leave the domain of the model (Km < 0: the reference's solver raises), so
        # the priors sit tightly around the posterior
        priors = {"Vmax": {"dist": "flat", "mu": 1.3, "sigma": 0.1}, "Km": {"dist": "normal", "mu": 0.6, "sigma": 0.05},
                  "sigma": {"dist": "normal", "mu": 0.0205, "sigma": 0.001}}
    so = O.SMCSettings(priors=priors)
    so.prior_mode = mode
    o = O.run_smc(data, so, seed=11, n_threads=0, record_mh=False)
    s = pkg.SMCSettings(seed=11, priors=priors, prior_mode=mode)
    with make_engine(pkg, data, 1000, priors=priors) as eng:
        out = pkg.run_smc(eng, s, rng="numpy", verbose=False)
        eng.set_prior_mode("mask")
    assert out["gamma"] == 1.0 and len(out["records"]) == len(o["records"])
    assert np.array_equal([r["gamma_new"] for r in out["records"]], [r.gamma_new for r in o["records"]])
    assert np.array_equal([r["n_accept"] for r in out["records"]], [r.n_accept for r in o["records"]])
    assert np.abs(out["p_pred"] - o["p_pred"]).max() < 1e-9
    assert relerr(out["lk"], o["lk"]).max() < TOL_LOGL
    # the ratio does change the chain: the same run with the support mask alone ends elsewhere
    o_mask = O.run_smc(data, O.SMCSettings(priors=priors), seed=11, n_threads=0, record_mh=False)
    assert np.abs(o_mask["p_pred"] - o["p_pred"]).max() > 1e-6


def test_full_size_properties_at_one_million_particles(pkg, O, data):
    """BASELINE.json configs[1] size (10^6 particles on one GPU), through size-independent properties: the likelihood
    of a 10^4-particle block tiled 100 times is bit-periodic (no dependence on scheduling or position) and equals the
    oracle on the block; the fused ESS sums and the two-pass moments equal NumPy's on the downloaded values;
    residual-systematic offspring sum to N (or N-1), every count lies within 1 of N w_i, the gathered particles are the
    ancestors in order; one Metropolis sweep keeps every particle either at its old or at its proposed position."""
    n, blk = 1_000_000, 10_000
    block = mixed_particles(blk, seed=9)
    th = np.tile(block, (n // blk, 1))
    s = pkg.SMCSettings(n_particle=n)
    with make_engine(pkg, data, n) as eng:
        eng.upload_particles(pkg.SMC_SET_PRED, th)
        info = eng.loglik(pkg.SMC_SET_PRED)
        lk = eng.download_lk(pkg.SMC_SET_PRED)
        assert info["n_failed"] == 0
        assert np.array_equal(lk.reshape(-1, blk), np.tile(lk[:blk], (n // blk, 1)))
        ref = O.mm_loglik_batch(block, data)[0]
        assert relerr(lk[:blk], ref).max() < TOL_LOGL
        # ESS partial sums for three increments at once, moments
        max_lk = eng.max_lk_local()
        assert max_lk == lk.max()
        gms = [1.0, 0.01, 1e-4]
        sw, sw2 = eng.ess_partials(max_lk, gms)
        for k, gm in enumerate(gms):
            w = np.exp((lk - max_lk) * gm)
            assert abs(sw[k] / w.sum() - 1) < 1e-12 and abs(sw2[k] / (w * w).sum() - 1) < 1e-12
        # resampling at an increment found by the reference's back-off
        es = pkg.ess_search(eng, pkg.SingleComm(), 0.0, s)
        out = pkg.resample(eng, pkg.SingleComm(), es, 0.61, s, True)
        off = eng.download_offspring()
        w = np.exp((lk - es["max_lk"]) * es["gm"])
        w = w / w.sum()
        assert out["n_offspring"] in (n - 1, n) and off.sum() == out["n_offspring"]
        assert np.all(off >= np.floor(n * w * (1 - 1e-12))) and np.all(np.abs(off - n * w) < 1.0 + 1e-6)
        anc = eng.download_particles(pkg.SMC_SET_FILT)
        src = np.repeat(np.arange(n), off)
        assert np.array_equal(anc[:len(src)], th[src])
        mean = eng.moment_sums_local() / n
        assert np.allclose(mean, anc.mean(axis=0), rtol=1e-12, atol=0)
        cen = eng.moment_centered_local(mean) / n
        assert np.allclose(cen, np.cov(anc.T, bias=True), rtol=1e-9, atol=0)
        # one device-RNG Metropolis sweep
        eng.set_debug_capture(True)
        cov_m = pkg.proposal_cov(eng, pkg.SingleComm(), s, s.w_cov())
        mh = eng.mh_step_device_rng(es["gamma_new"], 1.0, pkg.mvn_transform(cov_m), 5, 0, 0)
        prop, lk2, p0, r = eng.download_debug_proposals()
        after = eng.download_particles(pkg.SMC_SET_FILT)
        assert mh["accepted_now"] == int(r.sum()) and 0 < mh["accepted_now"] < n
        assert np.array_equal(after, np.where(r[:, None] == 1, prop, anc))


def test_resume_from_dump_is_bit_identical(pkg, data, tmp_path):
    """SURVEY.md 8(f) N3 "+ resume from a dump" (the reference only writes its dumps): a run continued from
    pred/{k}_p_pred.csv + {k}_state.json ends in exactly the particles, schedule and evidence of the uninterrupted
    run (device RNG: every draw is keyed by seed, particle, step and iteration)."""
    n = 8192
    s = pkg.SMCSettings(n_particle=n)
    with make_engine(pkg, data, n) as eng:
        full = pkg.run_smc(eng, s, rng="device", verbose=False, seed_device=21, dump_dir=str(tmp_path))
    k = 3
    assert full["step"] > k + 2
    with make_engine(pkg, data, n) as eng:
        cont = pkg.run_smc(eng, s, rng="device", verbose=False, seed_device=21, resume_from=(str(tmp_path), k))
    assert cont["step"] == full["step"] and cont["gamma"] == 1.0
    assert np.array_equal(cont["p_pred"], full["p_pred"]) and np.array_equal(cont["lk"], full["lk"])
    assert cont["logZ"] == full["logZ"]
    tail = full["records"][k:]
    assert [r["gamma_new"] for r in cont["records"]] == [r["gamma_new"] for r in tail]
    assert [r["n_accept"] for r in cont["records"]] == [r["n_accept"] for r in tail]


def test_resume_in_numpy_parity_mode_continues_the_reference_run(pkg, data, golden_run, tmp_path):
    """Parity mode draws from NumPy's global generator: the state file keeps that generator's state, so a run resumed
    after step 5 ends exactly where the reference's uninterrupted run ends (its final particles, its next random)."""
    g = golden_run
    with make_engine(pkg, data, 1000) as eng:
        pkg.run_smc(eng, pkg.SMCSettings(), rng="numpy", verbose=False, dump_dir=str(tmp_path))
    np.random.seed(12345)                                   # whatever happened to the global stream in between
    with make_engine(pkg, data, 1000) as eng:
        cont = pkg.run_smc(eng, pkg.SMCSettings(), rng="numpy", verbose=False, resume_from=(str(tmp_path), 5))
    assert cont["step"] == int(g["final_step"]) and cont["gamma"] == 1.0
    assert np.abs(cont["p_pred"] - g["final_p_pred"]).max() < 1e-9
    assert np.array_equal([r["gamma_new"] for r in cont["records"]], g["sched_gamma"][5:])
    assert np.random.rand() == float(g["next_rand_after_run"])


def test_systematic_resampling_option(pkg, data):
    """BASELINE.json names systematic resampling; the reference only has the residual variant, so the pin is the
    textbook definition in NumPy: offspring_i = #{k : (u + k)/N in (C_{i-1}, C_i]} with C = cumsum(w)."""
    n = 4096
    th = mixed_particles(n, seed=3)
    s = pkg.SMCSettings(n_particle=n, resampling="systematic")
    with make_engine(pkg, data, n) as eng:
        eng.set_resampling("systematic")
        eng.upload_particles(pkg.SMC_SET_PRED, th)
        eng.loglik(pkg.SMC_SET_PRED)
        lk = eng.download_lk(pkg.SMC_SET_PRED)
        es = pkg.ess_search(eng, pkg.SingleComm(), 0.0, s)
        u = 0.37
        out = pkg.resample(eng, pkg.SingleComm(), es, u, s, True)
        off = eng.download_offspring()
        anc = eng.download_particles(pkg.SMC_SET_FILT)
        eng.set_resampling("residual_systematic")
    w = np.exp((lk - es["max_lk"]) * es["gm"])
    w = w / w.sum()
    C = np.cumsum(w)
    thr = (u + np.arange(n)) / n
    ref = np.diff(np.concatenate([[0], np.searchsorted(thr, C, side="right")]))
    assert out["n_offspring"] in (n - 1, n) and off.sum() == out["n_offspring"]
    assert np.all(np.abs(off - n * w) < 1.0 + 1e-9)            # systematic: every count within 1 of N w_i
    assert np.abs(off - ref).sum() <= 2                         # a threshold within rounding of a boundary may move
    assert out["n_tmp_before"] == n                             # no deterministic copies
    src = np.repeat(np.arange(n), off)
    assert np.array_equal(anc[:len(src)], th[src])


def test_multinomial_resampling_option(pkg, data):
    """BASELINE.json names multinomial resampling (absent from the reference).  Statistical pin: over repeated
    draws the offspring counts must have the multinomial mean N w_i AND variance N w_i (1 - w_i) - the systematic
    schemes have variance < 1, so the second moment tells them apart - and always sum to N."""
    n, reps = 2048, 400
    th = mixed_particles(n, seed=5)
    s = pkg.SMCSettings(n_particle=n, resampling="multinomial")
    rs = np.random.RandomState(1)
    with make_engine(pkg, data, n) as eng:
        eng.set_resampling("multinomial")
        eng.upload_particles(pkg.SMC_SET_PRED, th)
        eng.loglik(pkg.SMC_SET_PRED)
        lk = eng.download_lk(pkg.SMC_SET_PRED)
        es = pkg.ess_search(eng, pkg.SingleComm(), 0.0, s)
        offs = np.empty((reps, n))
        for k in range(reps):
            out = pkg.resample(eng, pkg.SingleComm(), es, float(rs.uniform()), s, k == 0)
            offs[k] = eng.download_offspring()
            assert out["n_offspring"] == n and offs[k].sum() == n
        anc = eng.download_particles(pkg.SMC_SET_FILT)
        eng.set_resampling("residual_systematic")
    assert np.array_equal(anc, th[np.repeat(np.arange(n), offs[-1].astype(int))])
    w = np.exp((lk - es["max_lk"]) * es["gm"])
    w = w / w.sum()
    heavy = np.argsort(w)[-40:]                       # N w between ~2 and ~20 here
    mean, var = offs[:, heavy].mean(axis=0), offs[:, heavy].var(axis=0, ddof=1)
    exp_mean, exp_var = n * w[heavy], n * w[heavy] * (1 - w[heavy])
    assert np.all(np.abs(mean - exp_mean) < 5 * np.sqrt(exp_var / reps))
    assert np.all(np.abs(var / exp_var - 1) < 0.45) and abs(np.mean(var / exp_var) - 1) < 0.1
    assert len({tuple(o) for o in offs[:5]}) == 5     # different seeds, different draws


def test_ess_bisection_option(pkg, data):
    """BASELINE.json: "adaptive tempering via ESS bisection" (the reference backs off geometrically and stops up to
    30 % short of the crossing).  Each step must land on ESS/N = ess_limit from above, the run must need fewer
    tempering steps than the back-off, and both must agree on the posterior."""
    n = 65536
    runs = {}
    for how in ("backoff", "bisection"):
        s = pkg.SMCSettings(n_particle=n, ess_search=how)
        with make_engine(pkg, data, n) as eng:
            runs[how] = pkg.run_smc(eng, s, rng="device", verbose=False, seed_device=5)
    a, b = runs["backoff"], runs["bisection"]
    assert a["gamma"] == 1.0 and b["gamma"] == 1.0
    assert b["step"] < a["step"]
    for r in b["records"][:-1]:
        assert 0.5 < r["ess"] < 0.5 + 1e-4
    ma, mb = a["p_pred"].mean(axis=0), b["p_pred"].mean(axis=0)
    assert np.all(np.abs(ma - mb) <= 6 * a["p_pred"].std(axis=0) / np.sqrt(n) * 3)
    # the evidence estimate depends on the schedule through the few Metropolis sweeps per step (574.4 vs 575.5 here)
    assert abs(a["logZ"] - b["logZ"]) < 2.5


def test_full_run_device_rng_statistics(pkg, data, golden_run):
    """Device-RNG mode cannot share NumPy's stream; posterior moments must agree statistically with the
    reference posterior (N=1000): |mean_gpu - mean_ref| <= 5 * std / sqrt(1000) per parameter."""
    g = golden_run["final_p_pred"]
    n = 65536
    s = pkg.SMCSettings(n_particle=n)
    with make_engine(pkg, data, n) as eng:
        out = pkg.run_smc(eng, s, rng="device", verbose=False, seed_device=99)
    assert out["gamma"] == 1.0
    m, sd = out["p_pred"].mean(axis=0), out["p_pred"].std(axis=0)
    assert np.all(np.abs(m - g.mean(axis=0)) <= 5 * g.std(axis=0) / np.sqrt(1000))
    assert np.all(np.abs(sd / g.std(axis=0) - 1) < 0.2)
    assert 560 < out["logZ"] < 575


def test_log_evidence_device_rng_vs_oracle_replicates(pkg, O, data):
    """in device-RNG mode the streams differ from NumPy's, so the log-evidence is compared through
    replicates - 8 seeds each of the oracle driver (NumPy stream) and of the GPU run (Philox) at N = 1000: the means
    must agree within 4 combined standard errors, and the posterior means likewise."""
    n, reps = 1000, 8
    zo, zg, mo, mg = [], [], [], []
    for k in range(reps):
        o = O.run_smc(data, O.SMCSettings(), seed=100 + k, n_threads=0, record_mh=False)
        zo.append(o["logZ"])
        mo.append(o["p_pred"].mean(axis=0))
    with make_engine(pkg, data, n) as eng:
        for k in range(reps):
            g = pkg.run_smc(eng, pkg.SMCSettings(), rng="device", verbose=False, seed_device=500 + k)
            assert g["gamma"] == 1.0
            zg.append(g["logZ"])
            mg.append(g["p_pred"].mean(axis=0))
    zo, zg, mo, mg = np.array(zo), np.array(zg), np.array(mo), np.array(mg)
    se = np.sqrt(zo.var(ddof=1) / reps + zg.var(ddof=1) / reps)
    assert abs(zo.mean() - zg.mean()) < 4 * se, (zo, zg)
    sem = np.sqrt(mo.var(axis=0, ddof=1) / reps + mg.var(axis=0, ddof=1) / reps)
    assert np.all(np.abs(mo.mean(axis=0) - mg.mean(axis=0)) < 4 * sem), (mo.mean(axis=0), mg.mean(axis=0))


def test_philox_known_answer_and_prior_draw(pkg, data):
    """Philox4x32-10 known-answer vector (Random123) through a pure-Python restatement, and the device
    prior draw against it."""
    def philox(c, k):
        M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
        c = list(c)
        k = list(k)
        for _ in range(10):
            p0, p1 = M0 * c[0], M1 * c[2]
            c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & 0xFFFFFFFF, (p0 >> 32) ^ c[3] ^ k[1], p0 & 0xFFFFFFFF]
            k = [(k[0] + W0) & 0xFFFFFFFF, (k[1] + W1) & 0xFFFFFFFF]
        return c
    assert philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    n, seed, goff = 300, 0x1234567890ABCDEF, 5_000_000_000
    pri = {"a": {"dist": "uniform", "low": 0, "high": 1}, "b": {"dist": "uniform", "low": 2, "high": 6},
           "c": {"dist": "uniform", "low": -1, "high": 1}}
    with make_engine(pkg, data, n, priors=pri) as eng:
        eng.sample_prior_device(seed, goff)
        x = eng.download_particles(pkg.SMC_SET_PRED)
    for p in [0, 1, 77, 299]:
        g = goff + p
        for c, (lo, hi) in enumerate([(0, 1), (2, 6), (-1, 1)]):
            stream = 0xFFFFFFFF00000000 | c
            r = philox([g & 0xFFFFFFFF, g >> 32, stream & 0xFFFFFFFF, ((stream >> 32) << 8) & 0xFFFFFFFF],
                       [seed & 0xFFFFFFFF, seed >> 32])
            u = float(((r[0] >> 5) << 26) | (r[1] >> 6)) / 9007199254740992.0
            assert x[p, c] == lo + (hi - lo) * u


def test_errors_are_reported(pkg, data):
    with pkg.HipEngine(10, 3) as eng:
        with pytest.raises(pkg.SmcError):
            eng.loglik(pkg.SMC_SET_PRED)              # model not set
        with pytest.raises(pkg.SmcError):
            eng.upload_particles(pkg.SMC_SET_PRED, np.zeros((11, 3)))   # exceeds capacity
    with pytest.raises(pkg.SmcError):
        pkg.HipEngine(10, 3, device=99)


def test_rccl_call_sequence_single_rank(pkg, data, golden_run, monkeypatch):
    """The RCCL communicator path (ncclCommInitRank, allreduce / allgather on device scratch) with one
    rank: the collectives must be the identity and the full run must equal the reference run."""
    monkeypatch.setenv("SMC_FORCE_RCCL", "1")
    g = golden_run
    with make_engine(pkg, data, 1000) as eng:
        comm = pkg.RcclComm(eng, 0, 1, lambda uid: uid)
        assert np.array_equal(comm.allreduce_sum([1.5, -2.0, 3.25]), [1.5, -2.0, 3.25])
        assert np.array_equal(comm.allreduce_max([7.0]), [7.0])
        assert np.array_equal(comm.allreduce_sum_i64([5, 6]), [5, 6])
        assert np.array_equal(comm.allgather([1.0, 2.0]), [[1.0, 2.0]])
        assert np.array_equal(comm.allgather_i64([9]), [[9]])
        comm.barrier()
        # the particle exchange of smc_resample_phase3 with real ncclSend / ncclRecv, addressed to this rank itself
        th = mixed_particles(1000, seed=4)
        eng.upload_particles(pkg.SMC_SET_PRED, th)
        eng.upload_lk(pkg.SMC_SET_PRED, np.arange(1000.0))
        eng.upload_particles(pkg.SMC_SET_FILT, np.zeros((1000, 3)))
        eng.upload_lk(pkg.SMC_SET_FILT, np.zeros(1000))
        eng.debug_rccl_self_exchange(100, 300, 650)
        eng.debug_rccl_self_exchange(0, 7, 3)                      # a second, smaller block reuses the staging buffers
        f, flk = eng.download_particles(pkg.SMC_SET_FILT), eng.download_lk(pkg.SMC_SET_FILT)
        expect, elk = np.zeros((1000, 3)), np.zeros(1000)
        expect[650:950], elk[650:950] = th[100:400], np.arange(100.0, 400.0)
        expect[3:10], elk[3:10] = th[0:7], np.arange(0.0, 7.0)
        assert np.array_equal(f, expect) and np.array_equal(flk, elk)
        out = pkg.run_smc(eng, pkg.SMCSettings(), comm=comm, rng="numpy", verbose=False)
    assert np.array_equal([r["gamma_new"] for r in out["records"]], g["sched_gamma"])
    assert np.array_equal([r["n_accept"] for r in out["records"]], g["sched_accept"])
    assert np.abs(out["p_pred"] - g["final_p_pred"]).max() < 1e-9


def _row_signs(a):
    """Rows with their largest component positive (the sign convention of mh_transform_kernel; LAPACK's is arbitrary)."""
    a = np.array(a, dtype=float)
    for i in range(len(a)):
        if a[i, np.argmax(np.abs(a[i]))] < 0:
            a[i] = -a[i]
    return a


@pytest.mark.parametrize("force_rccl", [False, True])
def test_fused_mh_iteration_equals_its_parts(pkg, data, force_rccl, monkeypatch):
    """smc_mh_iteration_device_rng (Micmem_SMC_main.py:212-241 in one call: device-side moments -> cov_m -> the
    multivariate_normal factor on the device -> propose -> solve -> accept -> counts) against the same iteration composed
    from its parts on the host.  Pins:
      * cov_m equals np.cov(p_filt.T, bias=True) * w_cov to summation-order tolerance;
      * the device factor equals driver.mvn_transform(cov_m) - NumPy's sqrt(s)[:, None] * v of svd(cov_m) - row for row up
        to the sign of a row (1e-10 relative), and factor^T factor == cov_m;
      * fed with that very factor, smc_mh_step_device_rng makes bit-identical decisions and particles (same Philox keys).
    With SMC_FORCE_RCCL=1 the three in-place ncclAllReduce calls of the fused path run on a one-rank communicator."""
    if force_rccl:
        monkeypatch.setenv("SMC_FORCE_RCCL", "1")
    n = 20000
    s = pkg.SMCSettings(n_particle=n)
    th = mixed_particles(n, seed=9)
    th[: n // 2] = np.array([1.2, 0.5, 0.03]) + np.random.RandomState(2).standard_normal((n // 2, 3)) * [0.2, 0.1, 0.004]
    w_cov = s.w_cov()
    outs = []
    for fused in (True, False):
        with make_engine(pkg, data, n) as eng:
            comm = pkg.RcclComm(eng, 0, 1, lambda uid: uid) if force_rccl else pkg.SingleComm()
            eng.upload_particles(pkg.SMC_SET_PRED, th)
            eng.loglik(pkg.SMC_SET_PRED)
            eng.upload_particles(pkg.SMC_SET_FILT, th)
            eng.upload_lk(pkg.SMC_SET_FILT, eng.download_lk(pkg.SMC_SET_PRED))
            if fused:
                out = eng.mh_iteration_device_rng(0.3, 0.7, w_cov, 99, (5 << 16) | 2, 0)
                xf = eng.mh_iteration_last_transform()
            else:
                cov_host = pkg.proposal_cov(eng, comm, s, w_cov)
                out = eng.mh_step_device_rng(0.3, 0.7, outs[0]["xf"], 99, (5 << 16) | 2, 0)
                out["cov_m"] = cov_host
                xf = outs[0]["xf"]
            outs.append({"out": out, "xf": xf, "filt": eng.download_particles(pkg.SMC_SET_FILT),
                         "lk": eng.download_lk(pkg.SMC_SET_FILT), "flags": eng.download_accept_flags()})
    a, b = outs
    cov_np = np.cov(th.T, bias=True) * w_cov
    assert np.abs(a["out"]["cov_m"] - cov_np).max() <= 1e-11 * np.abs(cov_np).max()
    assert np.abs(a["out"]["cov_m"] - b["out"]["cov_m"]).max() <= 1e-13 * np.abs(cov_np).max()
    ref = pkg.mvn_transform(a["out"]["cov_m"])                      # NumPy / LAPACK on the host
    assert np.abs(a["xf"].T @ a["xf"] - a["out"]["cov_m"]).max() <= 1e-13 * np.abs(cov_np).max()
    assert np.abs(_row_signs(a["xf"]) - _row_signs(ref)).max() <= 1e-10 * np.abs(ref).max()
    assert np.array_equal(_row_signs(a["xf"]), a["xf"])           # the kernel's own sign convention
    for k in ("accepted_now", "accepted_ever", "n_failed"):
        assert a["out"][k] == b["out"][k], k
    # (rk_attempts: with early rejection on, where a certainly-rejected solve stops depends on when its siblings finished)
    assert abs(a["out"]["rk_attempts"] - b["out"]["rk_attempts"]) <= 0.01 * b["out"]["rk_attempts"]
    assert 0 < a["out"]["accepted_now"] < n
    assert np.array_equal(a["filt"], b["filt"]) and np.array_equal(a["lk"], b["lk"]) and np.array_equal(a["flags"], b["flags"])


def test_fused_mh_iterations_carry_their_moments(pkg, data):
    """From the second fused iteration of a step on, cov_m comes from moments the accept kernel of the previous iteration
    accumulated over the particles it selected (about the previous mean) - no pass over the particles.  It must still be
    np.cov(p_filt.T, bias=True) * w_cov of the population the iteration starts from, to summation-order tolerance, also
    for a population whose spread (1e-3 in sigma) is far smaller than its location; and anything else that rewrites p_filt
    (here: an upload) must send the next iteration back to the two-pass start."""
    n = 50000
    s = pkg.SMCSettings(n_particle=n)
    w_cov = s.w_cov()
    rs = np.random.RandomState(21)
    th = np.array([1.2254, 0.5218, 0.02048]) + rs.standard_normal((n, 3)) * np.array([0.025, 0.0295, 0.00094])
    with make_engine(pkg, data, n) as eng:
        eng.upload_particles(pkg.SMC_SET_PRED, th)
        eng.loglik(pkg.SMC_SET_PRED)
        eng.upload_particles(pkg.SMC_SET_FILT, th)
        eng.upload_lk(pkg.SMC_SET_FILT, eng.download_lk(pkg.SMC_SET_PRED))
        cur = th
        for j in range(4):
            out = eng.mh_iteration_device_rng(1.0, 1.0, w_cov, 5, (9 << 16) | j, 0)
            ref = np.cov(cur.T, bias=True) * w_cov
            assert np.abs(out["cov_m"] - ref).max() <= 1e-11 * np.abs(ref).max(), j
            new = eng.download_particles(pkg.SMC_SET_FILT)
            moved = np.any(new != cur, axis=1).sum()
            assert moved == out["accepted_now"] > 0.2 * n
            cur = new
        shifted = cur + np.array([0.5, -0.1, 0.001])
        eng.upload_particles(pkg.SMC_SET_FILT, shifted)             # invalidates the carried moments
        out = eng.mh_iteration_device_rng(1.0, 1.0, w_cov, 5, (9 << 16) | 9, 0)
        ref = np.cov(shifted.T, bias=True) * w_cov
        assert np.abs(out["cov_m"] - ref).max() <= 1e-11 * np.abs(ref).max()


def test_fused_iterations_survive_interleaved_small_result_calls(pkg, data):
    """the moments an accept kernel leaves for the NEXT fused iteration used to sit in d_small[0..12),
    the region every small-result entry point writes (smc_max_lk_*, smc_ess_partials*, smc_moment_sums/centered_local, the
    generic all-reduce helpers); a caller that called one of them between two fused iterations got a proposal covariance
    built from garbage and no error.  They now live in the iteration's own buffer: with such calls interleaved, cov_m of
    every iteration still equals np.cov of the population it starts from, and the whole sequence is bit-identical to the
    undisturbed one."""
    n = 30000
    s = pkg.SMCSettings(n_particle=n)
    w_cov = s.w_cov()
    rs = np.random.RandomState(22)
    th = np.array([1.2254, 0.5218, 0.02048]) + rs.standard_normal((n, 3)) * np.array([0.025, 0.0295, 0.00094])
    finals = {}
    for disturb in (False, True):
        with make_engine(pkg, data, n) as eng:
            eng.upload_particles(pkg.SMC_SET_PRED, th)
            eng.loglik(pkg.SMC_SET_PRED)
            eng.upload_particles(pkg.SMC_SET_FILT, th)
            eng.upload_lk(pkg.SMC_SET_FILT, eng.download_lk(pkg.SMC_SET_PRED))
            cur = th
            for j in range(4):
                out = eng.mh_iteration_device_rng(1.0, 1.0, w_cov, 5, (3 << 16) | j, 0)
                ref = np.cov(cur.T, bias=True) * w_cov
                assert np.abs(out["cov_m"] - ref).max() <= 1e-11 * np.abs(ref).max(), (disturb, j)
                cur = eng.download_particles(pkg.SMC_SET_FILT)
                if disturb:                                            # each of these writes d_small[0..]
                    m = eng.max_lk_local()
                    eng.ess_partials(m, np.array([0.1, 0.2, 0.3]))
                    mean = eng.moment_sums_local() / n
                    eng.moment_centered_local(mean)
                    eng.max_lk_global()
                    eng.ess_partials_global(m, np.array([0.5]))
                    eng.comm_allreduce_sum_f64(np.arange(12.0) * 1e9)
            finals[disturb] = (cur, eng.download_lk(pkg.SMC_SET_FILT))
    assert np.array_equal(finals[True][0], finals[False][0]) and np.array_equal(finals[True][1], finals[False][1])


@pytest.mark.parametrize("d", [2, 5, 8])
def test_device_mvn_factor_any_dimension(pkg, d):
    """smc_proposal_factor_device for d != 3 (the methanation model has d = 5; SMC_MAX_DIM = 8) on a badly scaled, correlated
    cloud: cov_m against NumPy, the Jacobi factor against NumPy's SVD factor row for row (up to the sign of a row)."""
    n = 4096
    rs = np.random.RandomState(d)
    scale = 10.0 ** rs.uniform(-3, 2, d)
    mix = rs.standard_normal((d, d))
    x = rs.standard_normal((n, d)) @ mix * scale
    w = np.full((d, d), 0.5)
    with pkg.HipEngine(n, d, device=0) as eng:
        eng.upload_particles(pkg.SMC_SET_FILT, x)
        cov, xf = eng.proposal_factor_device(w)
    ref_cov = np.cov(x.T, bias=True) * w
    assert np.abs(cov - ref_cov).max() <= 1e-11 * np.abs(ref_cov).max()
    ref = pkg.mvn_transform(cov)
    assert np.abs(xf.T @ xf - cov).max() <= 1e-12 * np.abs(cov).max()
    assert np.abs(_row_signs(xf) - _row_signs(ref)).max() <= 1e-9 * np.abs(ref).max()


def test_global_reductions_on_a_one_rank_communicator(pkg, data, monkeypatch):
    """The *_global entry points (partial on the device -> ncclAllReduce / ncclAllGather in place -> one read-back) on a
    one-rank RCCL communicator must return exactly what the *_local ones do, and a complete run through them
    (run_smc with RcclComm: on-device reductions, smc_resample_global, fused Metropolis iterations) must equal the run with
    no communicator at all, bit for bit."""
    n = 30000
    th = mixed_particles(n, seed=12)
    runs = []
    for force in (False, True):
        if force:
            monkeypatch.setenv("SMC_FORCE_RCCL", "1")
        with make_engine(pkg, data, n) as eng:
            comm = pkg.RcclComm(eng, 0, 1, lambda uid: uid) if force else pkg.SingleComm()
            eng.upload_particles(pkg.SMC_SET_PRED, th)
            eng.loglik(pkg.SMC_SET_PRED)
            m_loc, m_glob = eng.max_lk_local(), eng.max_lk_global()
            gms = [1.0, 0.5, 0.1, 0.01, 1e-3]
            sw_l, sw2_l = eng.ess_partials(m_loc, gms)
            sw_g, sw2_g = eng.ess_partials_global(m_glob, gms)
            assert m_loc == m_glob and np.array_equal(sw_l, sw_g) and np.array_equal(sw2_l, sw2_g)
            out = pkg.run_smc(eng, pkg.SMCSettings(n_particle=n), comm=comm, rng="device", verbose=False, seed_device=3)
            runs.append(out)
    a, b = runs
    assert a["gamma"] == b["gamma"] == 1.0
    assert [r["gamma_new"] for r in a["records"]] == [r["gamma_new"] for r in b["records"]]
    assert [r["n_accept"] for r in a["records"]] == [r["n_accept"] for r in b["records"]]
    assert all(r["n_offspring"] == n for r in b["records"])
    assert np.array_equal(a["p_pred"], b["p_pred"]) and np.array_equal(a["lk"], b["lk"]) and a["logZ"] == b["logZ"]


# ---------------------------------------------------------------------------------------------------
# multi-rank path with the real kernels on one GPU (loopback exchange instead of RCCL send/recv)
# ---------------------------------------------------------------------------------------------------
def _run_ranks(pkg, data, n, world, rng, seed, **settings):
    import threading
    from _thread_comm import ThreadWorld
    tw = ThreadWorld(world)
    nl = n // world
    engines = [pkg.HipEngine(nl, 3, device=0, n_global=n) for _ in range(world)]
    s = pkg.SMCSettings(n_particle=n, seed=seed, **settings)
    for r, e in enumerate(engines):
        e.set_model_mm(data.t, data.P_obs, data.S0)
        e.set_prior(s.priors)
        if world > 1:
            e.debug_set_local_peers(engines, r, tw.barrier.wait)
    p0 = None
    if rng == "numpy":
        np.random.seed(seed)
        p0 = pkg.sample_prior(s.priors, n)
    outs, errs = [None] * world, []
    lock = threading.Lock()

    def work(r):
        try:
            if rng == "numpy":
                # every rank must see the same global stream: serialise the host draws of a rank by replaying
                # the stream from a per-rank copy of the generator state
                raise RuntimeError("numpy mode is run single-threaded")
            outs[r] = pkg.run_smc(engines[r], s, comm=tw.comm(r), rng="device", verbose=False, seed_device=seed)
        except Exception as ex:  # noqa: BLE001
            with lock:
                errs.append(ex)
            tw.barrier.abort()
    ths = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for e in engines:
        e.close()
    if errs:
        raise errs[0]
    return outs


@pytest.mark.parametrize("world,n", [(2, 8192), (4, 8192), (2, 65536)])
def test_multi_rank_loopback_equals_single_rank(pkg, data, world, n):
    """Device-RNG mode is keyed by GLOBAL particle index: a run sharded over W ranks must reproduce the
    single-rank run - same schedule, same offspring, same particles (FP tolerance for the cross-rank sums).  The larger case
    has enough particles per rank for the cost-ordered hand-out (>= 16 384), which every rank decides and sorts for itself."""
    seed = 77
    ref = _run_ranks(pkg, data, n, 1, "device", seed)[0]
    outs = _run_ranks(pkg, data, n, world, "device", seed)
    for o in outs:
        assert [r["gamma_new"] for r in o["records"]] == [r["gamma_new"] for r in ref["records"]]
        assert [r["n_accept"] for r in o["records"]] == [r["n_accept"] for r in ref["records"]]
        assert [r["last_j"] for r in o["records"]] == [r["last_j"] for r in ref["records"]]
        assert all(r["n_offspring"] == n for r in o["records"])
        assert abs(o["logZ"] - ref["logZ"]) < 1e-9 * abs(ref["logZ"])
    p = np.concatenate([o["p_pred"] for o in outs])
    lk = np.concatenate([o["lk"] for o in outs])
    assert np.abs(p - ref["p_pred"]).max() < 1e-9
    assert relerr(lk, ref["lk"]).max() < TOL_LOGL


@pytest.mark.parametrize("scheme", ["systematic", "multinomial"])
def test_multi_rank_loopback_optional_schemes(pkg, data, scheme):
    """The resampling options and the ESS bisection sharded over 3 ranks: thresholds, cumulative weights and the
    bisection brackets are global quantities, so the sharded run must follow the single-rank one (a threshold within
    rounding of a rank boundary may move one offspring; the schedule and evidence must still agree closely)."""
    n, seed = 6144, 31
    ref = _run_ranks(pkg, data, n, 1, "device", seed, resampling=scheme, ess_search="bisection")[0]
    outs = _run_ranks(pkg, data, n, 3, "device", seed, resampling=scheme, ess_search="bisection")
    assert ref["gamma"] == 1.0
    for o in outs:
        assert o["gamma"] == 1.0 and o["step"] == ref["step"]
        assert np.allclose([r["gamma_new"] for r in o["records"]], [r["gamma_new"] for r in ref["records"]], rtol=1e-6, atol=0)
        assert all(r["n_offspring"] in (n - 1, n) for r in o["records"])
        assert abs(o["logZ"] - ref["logZ"]) < 1e-6 * abs(ref["logZ"])
    p = np.concatenate([o["p_pred"] for o in outs])
    same = np.all(p == ref["p_pred"], axis=1).mean()
    assert same > 0.99 or np.abs(p.mean(axis=0) - ref["p_pred"].mean(axis=0)).max() < 3 * ref["p_pred"].std(axis=0).max() / np.sqrt(n)


def test_multi_rank_resample_exchange_exact(pkg, O, data):
    """One resampling step over 3 ranks with strongly skewed weights (most offspring come from rank 0 and
    must travel): offspring counts and the redistributed particles equal the sequential oracle exactly."""
    import threading
    from _thread_comm import ThreadWorld
    world, n = 3, 3 * 1500
    nl = n // world
    rs = np.random.RandomState(12)
    lk = rs.standard_normal(n) * 3
    lk[:nl] += 25          # rank 0 holds nearly all the weight
    p_pred = rs.standard_normal((n, 3))
    mx, gm, u = lk.max(), 0.3, 0.6180339887
    w = np.exp((lk - mx) * gm)
    sum_w = float(np.sum(w))
    p_filt, lk1 = np.zeros((n, 3)), np.zeros(n)
    p_is, _, _ = O.resample(w / sum_w, u, p_pred, lk, p_filt, lk1)
    tw = ThreadWorld(world)
    engines = [pkg.HipEngine(nl, 3, device=0, n_global=n) for _ in range(world)]
    s = pkg.SMCSettings(n_particle=n)
    res = [None] * world
    for r, e in enumerate(engines):
        e.set_model_mm(data.t, data.P_obs, data.S0)
        e.set_prior(s.priors)
        e.debug_set_local_peers(engines, r, tw.barrier.wait)
        e.upload_particles(pkg.SMC_SET_PRED, p_pred[r * nl:(r + 1) * nl])
        e.upload_lk(pkg.SMC_SET_PRED, lk[r * nl:(r + 1) * nl])

    def work(r):
        pkg.resample(engines[r], tw.comm(r), {"max_lk": mx, "gm": gm, "sum_weight": sum_w}, u, s, True)
        res[r] = (engines[r].download_offspring(), engines[r].download_particles(pkg.SMC_SET_FILT),
                  engines[r].download_lk(pkg.SMC_SET_FILT))
    ths = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for e in engines:
        e.close()
    assert all(x is not None for x in res)
    assert np.array_equal(np.concatenate([x[0] for x in res]), p_is)
    assert np.array_equal(np.concatenate([x[1] for x in res]), p_filt)
    assert np.array_equal(np.concatenate([x[2] for x in res]), lk1)
    assert p_is[:nl].sum() > 0.9 * n      # the exchange really moved particles across ranks


# ---------------------------------------------------------------------------------------------------
# "next" rows: per-step dumps (N3) and the pseudo-data generator (N4)
# ---------------------------------------------------------------------------------------------------
def test_per_step_dumps_match_reference_format(pkg, data, golden_run, tmp_path):
    g = golden_run
    with make_engine(pkg, data, 1000) as eng:
        out = pkg.run_smc(eng, pkg.SMCSettings(), rng="numpy", verbose=False, dump_dir=str(tmp_path))
    first = np.loadtxt(tmp_path / "pred" / "first_p_pred.csv", delimiter=",")
    assert np.array_equal(first, g["sweeps_theta"][0])                     # the prior sample, full precision
    steps = int(g["final_step"])
    for k in range(1, steps):                                               # the terminating step is not dumped (:259-266)
        assert (tmp_path / "pred" / f"{k}_p_pred.csv").exists()
    assert not (tmp_path / "pred" / f"{steps}_p_pred.csv").exists()
    last = np.loadtxt(tmp_path / "pred" / "last_p_pred.csv", delimiter=",")
    assert np.array_equal(last, out["p_pred"]) and np.abs(last - g["final_p_pred"]).max() < 1e-9
    import pandas as pd
    post = pd.read_csv(tmp_path / "Posterior_Distribution.csv", float_precision="round_trip")
    assert list(post.columns) == ["Vmax", "Km", "sigma"] and np.array_equal(post.values, last)
    # a dump is a valid restart point
    with make_engine(pkg, data, 1000) as eng:
        eng.upload_particles(pkg.SMC_SET_PRED, np.loadtxt(tmp_path / "pred" / "5_p_pred.csv", delimiter=","))
        assert eng.loglik(pkg.SMC_SET_PRED)["n_failed"] == 0


def test_pseudo_data_generator_reproduces_reference_csvs(pkg, data, tmp_path):
    """The committed data files mm_pseudo_data_1..5 were produced by the reference's generator with seeds
    20250205+i; the GPU generator reproduces them (same NumPy noise, P_true from the HIP RK45: <= 1e-9)."""
    frames = pkg.datagen.make_pseudo_data(out_dir=str(tmp_path))
    for i, df in enumerate(frames, start=1):
        assert np.allclose(df["t"].values, data.t[i], rtol=1e-15, atol=0)   # the reference re-reads t from CSV (1-ulp parser error)
        assert np.abs(df["P_obs"].values - data.P_obs[i]).max() < 1e-9
        assert df["S_true"].iloc[0] == data.S0[i]
        assert (tmp_path / f"mm_pseudo_data_{i}.csv").exists()


# ---------------------------------------------------------------------------------------------------
# exact early rejection (smc_set_early_reject): nothing observable may change
# ---------------------------------------------------------------------------------------------------
def test_early_rejection_changes_nothing_but_the_attempt_count(pkg, O, data):
    """A Metropolis iteration on a prior-like population (where the long solves live) with and without early rejection,
    host-RNG mode against the oracle's decisions, and a complete device-RNG run: p_filt, lk1, accept flags, accept counts,
    the tempering schedule and the evidence must be bit-identical; only rk_attempts may (and must) shrink."""
    n = 60000
    s = pkg.SMCSettings(n_particle=n)
    rs = np.random.RandomState(3)
    th = rs.uniform(0, 10, (n, 3))
    th[: n // 50, 1] = 10.0 ** rs.uniform(-3.3, -2, n // 50)          # stiff band: Vmax / Km in the thousands
    noise = rs.standard_normal((n, 3)) * np.array([0.5, 0.002, 0.5])
    rr = rs.uniform(0, 1, n)
    lk_ref = O.mm_loglik_batch(th, data)[0]
    res = {}
    for on in (True, False):
        with make_engine(pkg, data, n) as eng:
            eng.set_early_reject(on)
            eng.upload_particles(pkg.SMC_SET_FILT, th)
            eng.upload_lk(pkg.SMC_SET_FILT, lk_ref)
            out = eng.mh_step_host_rng(0.004, 1.0, noise, rr)
            res[on] = (out, eng.download_particles(pkg.SMC_SET_FILT), eng.download_lk(pkg.SMC_SET_FILT),
                       eng.download_accept_flags())
    a, b = res[True], res[False]
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    assert a[0]["accepted_now"] == b[0]["accepted_now"] and a[0]["accepted_ever"] == b[0]["accepted_ever"]
    assert a[0]["n_failed"] == b[0]["n_failed"] == 0
    assert a[0]["rk_attempts"] < 0.95 * b[0]["rk_attempts"], (a[0]["rk_attempts"], b[0]["rk_attempts"])
    # ... and both equal the reference's statements evaluated by the oracle
    p_pred, p0, lk2, r, f_ref, l_ref = _mh_reference(O, data, s.priors, th, lk_ref, noise, rr, 0.004, 1.0)
    assert np.array_equal(a[1], f_ref) and a[0]["accepted_now"] == int(r.sum())
    runs = {}
    for on in (True, False):
        with make_engine(pkg, data, n) as eng:
            runs[on] = pkg.run_smc(eng, pkg.SMCSettings(n_particle=n, early_reject=on), rng="device", verbose=False, seed_device=17)
    a, b = runs[True], runs[False]
    assert [r_["gamma_new"] for r_ in a["records"]] == [r_["gamma_new"] for r_ in b["records"]]
    assert [r_["n_accept"] for r_ in a["records"]] == [r_["n_accept"] for r_ in b["records"]]
    assert np.array_equal(a["p_pred"], b["p_pred"]) and np.array_equal(a["lk"], b["lk"]) and a["logZ"] == b["logZ"]
    assert a["stats"]["rk_attempts_mh"] < b["stats"]["rk_attempts_mh"]


@pytest.mark.gpu
def test_early_rejection_at_one_million_particles_is_bit_identical(pkg, data):
    """BASELINE.json's size: a complete device-RNG run over 10^6 particles with and without early rejection ends in the same
    particles, likelihoods, schedule and evidence, bit for bit, with fewer Metropolis-phase attempts (the time shrinks far more than the count: what is cancelled are the
    long serial chains)."""
    n = 1_000_000
    runs = {}
    for on in (True, False):
        with make_engine(pkg, data, n) as eng:
            runs[on] = pkg.run_smc(eng, pkg.SMCSettings(n_particle=n, early_reject=on), rng="device", verbose=False, seed_device=5)
    a, b = runs[True], runs[False]
    assert [r_["gamma_new"] for r_ in a["records"]] == [r_["gamma_new"] for r_ in b["records"]]
    assert [r_["n_accept"] for r_ in a["records"]] == [r_["n_accept"] for r_ in b["records"]]
    assert np.array_equal(a["p_pred"], b["p_pred"]) and np.array_equal(a["lk"], b["lk"]) and a["logZ"] == b["logZ"]
    assert a["stats"]["rk_attempts_mh"] < b["stats"]["rk_attempts_mh"]      # 2.6 % fewer attempts, a third of the time


# ---------------------------------------------------------------------------------------------------
# stiff-first hand-out (smc_set_stiff_first): the order in which independent solves run changes nothing
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,stiff_share", [(1, 1.0), (63, 0.5), (65, 0.0), (200, 1.0), (4099, 0.02), (50000, 0.3)])
def test_stiff_first_handout_is_bit_identical_to_index_order(pkg, O, data, n, stiff_share):
    """Likelihood sweep (with predictions: the WRITE_PRED instantiation, and without) and one host-RNG Metropolis
    iteration with a third of the proposals outside the prior box, with the stiff list on and off: logL, predictions,
    device-counted RK45 attempts, selected particles and accept flags must be bit-identical, for list sizes from empty to
    every particle (n_stiff = n: the index-ordered pass then skips everything), particle counts around the 64-lane group
    size, and against the oracle."""
    rs = np.random.RandomState(n)
    th = rs.uniform(0.05, 10, (n, 3))
    k = int(round(n * stiff_share))
    th[:k, 1] = th[:k, 0] / 10.0 ** rs.uniform(2.0, 3.2, k)             # Vmax / Km = 100 ... 1600: on the list
    th[k:, 1] = np.maximum(th[k:, 1], th[k:, 0] / 50.0)                # below the threshold of 60
    rs.shuffle(th)
    lk_ref, _, info_ref = O.mm_loglik_batch(th, data)
    noise = rs.standard_normal((n, 3)) * np.array([0.3, 0.001, 0.3])
    noise[::3, 2] = 1e3                                                 # sigma far outside [0, 10]: masked proposals
    rr = rs.uniform(0, 1, n)
    res = {}
    for on in (True, False):
        with make_engine(pkg, data, n) as eng:
            eng.set_stiff_first(on)
            eng.set_early_reject(False)                                 # so that the attempt counts are deterministic too
            lk_h, pred_h, info_h = eng.loglik_host(th, want_pred=True)
            eng.upload_particles(pkg.SMC_SET_PRED, th)
            info = eng.loglik(pkg.SMC_SET_PRED)
            lk = eng.download_lk(pkg.SMC_SET_PRED)
            eng.upload_particles(pkg.SMC_SET_FILT, th)
            eng.upload_lk(pkg.SMC_SET_FILT, lk)
            out = eng.mh_step_host_rng(0.01, 1.0, noise, rr)
            res[on] = (lk_h, pred_h, info_h["rk_attempts"], lk, info["rk_attempts"], out["rk_attempts"], out["accepted_now"],
                       eng.download_particles(pkg.SMC_SET_FILT), eng.download_lk(pkg.SMC_SET_FILT), eng.download_accept_flags())
    a, b = res[True], res[False]
    for x, y in zip(a, b):
        assert np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True)
    assert a[2] == a[4] and abs(a[2] - info_ref["n_attempts"]) <= 1e-4 * info_ref["n_attempts"]
    assert np.max(relerr(a[3], lk_ref)) < 1e-6 and np.array_equal(a[0], a[3])    # stiff band: DESIGN.md "Parity in the stiff band"


def test_stiff_first_full_run_is_bit_identical(pkg, data):
    """A complete device-RNG run (early rejection on, as in the benchmark) with and without the stiff list."""
    n = 200_000
    runs = {}
    for on in (True, False):
        with make_engine(pkg, data, n) as eng:
            runs[on] = pkg.run_smc(eng, pkg.SMCSettings(n_particle=n, stiff_first=on), rng="device", verbose=False, seed_device=23)
    a, b = runs[True], runs[False]
    assert [r_["gamma_new"] for r_ in a["records"]] == [r_["gamma_new"] for r_ in b["records"]]
    assert [r_["n_accept"] for r_ in a["records"]] == [r_["n_accept"] for r_ in b["records"]]
    assert np.array_equal(a["p_pred"], b["p_pred"]) and np.array_equal(a["lk"], b["lk"]) and a["logZ"] == b["logZ"]


# ---------------------------------------------------------------------------------------------------
# the hand-written lone-chain loop (smc_set_fast_tail; csrc/mm_rk45.h: mm_fast_uniform_attempts) against the compiled step function
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,seed", [(1, 0), (7, 1), (64, 2), (700, 3), (20000, 4)])
def test_hand_written_lone_chain_loop_is_bit_identical_to_the_compiled_step(pkg, O, data, n, seed):
    """The inline-asm attempt loop that solo solves and last survivors run must produce the bits of mm_item_attempt: same
    operations, same order, same fused multiply-adds.  Populations in which every particle is stiff (Vmax / Km = 60 ... 2e4:
    stiff list and solo list in use; with n = 1, 7, 64 every solve is a solo solve, i.e. runs in the block from its first
    attempt to its last apart from the attempts that produce outputs), likelihood sweep with and without predictions, one
    Metropolis sweep; logL, predictions, device-counted attempts, selected particles, accept flags: all equal with the block
    on and off - and both within the stiff band's tolerance of the checker."""
    rs = np.random.RandomState(seed)
    th = rs.uniform(0.05, 10, (n, 3))
    th[:, 1] = th[:, 0] / 10.0 ** rs.uniform(1.8, 4.3, n)
    th[0] = (10.0, 3e-3, 1.0)                                           # tools/tail_latency.py's 12 562-attempt solve
    noise = rs.standard_normal((n, 3)) * np.array([0.3, 1e-4, 0.3])
    rr = rs.uniform(0, 1, n)
    res = {}
    for on in (True, False):
        with make_engine(pkg, data, n) as eng:
            eng.set_fast_tail(on)
            eng.set_early_reject(False)                                 # so that the attempt counts are deterministic too
            lk_h, pred_h, info_h = eng.loglik_host(th, want_pred=True)
            eng.upload_particles(pkg.SMC_SET_PRED, th)
            info = eng.loglik(pkg.SMC_SET_PRED)
            lk = eng.download_lk(pkg.SMC_SET_PRED)
            eng.upload_particles(pkg.SMC_SET_FILT, th)
            eng.upload_lk(pkg.SMC_SET_FILT, lk)
            out = eng.mh_step_host_rng(0.01, 1.0, noise, rr)
            res[on] = (lk_h, pred_h, info_h["rk_attempts"], lk, info["rk_attempts"], out["rk_attempts"], out["accepted_now"],
                       eng.download_particles(pkg.SMC_SET_FILT), eng.download_lk(pkg.SMC_SET_FILT), eng.download_accept_flags())
    a, b = res[True], res[False]
    for k, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True), k
    assert a[2] == a[4] and np.array_equal(a[0], a[3])
    if n <= 700:
        lk_ref, _, info_ref = O.mm_loglik_batch(th, data)
        assert np.max(relerr(a[3], lk_ref)) < 1e-6 and abs(a[2] - info_ref["n_attempts"]) <= 1e-3 * info_ref["n_attempts"]


def test_hand_written_lone_chain_loop_full_run_is_bit_identical(pkg, data):
    """A complete device-RNG run (early rejection on, as in the benchmark) with the block on and off."""
    n = 200_000
    runs = {}
    for on in (True, False):
        with make_engine(pkg, data, n) as eng:
            eng.set_fast_tail(on)
            runs[on] = pkg.run_smc(eng, pkg.SMCSettings(n_particle=n), rng="device", verbose=False, seed_device=37)
    a, b = runs[True], runs[False]
    assert [r_["gamma_new"] for r_ in a["records"]] == [r_["gamma_new"] for r_ in b["records"]]
    assert [r_["n_accept"] for r_ in a["records"]] == [r_["n_accept"] for r_ in b["records"]]
    assert np.array_equal(a["p_pred"], b["p_pred"]) and np.array_equal(a["lk"], b["lk"]) and a["logZ"] == b["logZ"]


# ---------------------------------------------------------------------------------------------------
# cost order (smc_set_cost_order): heterogeneous Metropolis sweeps hand their solves out by cost class, in phase
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [16384, 70001, 300000])
def test_cost_ordered_handout_changes_no_result(pkg, data, n):
    """Four fused iterations on a prior-like population (proposals from 2^-12 to beyond 2^19 in Vmax / Km: every class of the
    counting sort, out-of-support proposals in the last one, stiff and solo lists in use) with the cost order on and off:
    accept counts, particles and likelihoods bit-identical; and every proposal was handed out exactly once (the sweep's
    items all carry a result: no NaN likelihood, no failure)."""
    s = pkg.SMCSettings(n_particle=n)
    w_cov = s.w_cov()
    rs = np.random.RandomState(n)
    th = rs.uniform(0.0, 10.0, (n, 3))
    th[: n // 8, 1] = 10.0 ** rs.uniform(-5, 1, n // 8)                   # very stiff down to Vmax / Km = 1e5 ... and below 1e-3 through Vmax
    th[n // 8: n // 4, 0] = 10.0 ** rs.uniform(-5, 1, n // 8)
    res = {}
    for on in (True, False):
        with make_engine(pkg, data, n) as eng:
            eng.set_cost_order(on)
            eng.upload_particles(pkg.SMC_SET_PRED, th)
            eng.loglik(pkg.SMC_SET_PRED)
            eng.upload_particles(pkg.SMC_SET_FILT, th)
            eng.upload_lk(pkg.SMC_SET_FILT, eng.download_lk(pkg.SMC_SET_PRED))
            outs = [eng.mh_iteration_device_rng(0.01, 1.0, w_cov, 11, (3 << 16) | j, 0) for j in range(4)]
            res[on] = ([o["accepted_now"] for o in outs], [o["n_failed"] for o in outs], eng.download_particles(pkg.SMC_SET_FILT),
                       eng.download_lk(pkg.SMC_SET_FILT))
    assert res[True][0] == res[False][0] and res[True][1] == res[False][1] == [0, 0, 0, 0]
    assert np.array_equal(res[True][2], res[False][2]) and np.array_equal(res[True][3], res[False][3])
    assert np.isfinite(res[True][3]).all()


def test_cost_ordered_handout_solves_in_support_proposals_without_a_cost_ratio(pkg, data):
    """under a normal prior on Km a proposal with Km <= 0 is in the support - but has no Vmax / Km cost ratio.  It used
    to share the counting sort's last class with the out-of-support proposals, which the solve kernel never visits: nobody wrote
    its sums (early rejection on: the pending NaN became lk2 and NaN * 0 + lk1 poisoned lk1; off: the previous sweep's stale sums
    fed the accept test).  With the fix it sits in a real class and is solved like any other.
    The case is made deterministic with Km == 0 exactly (dS/dt = -Vmax: a benign ODE that every path of the kernel integrates
    alike; for Km < 0 the solution runs into the pole S = -Km and what a solver returns there depends on its last bits): an
    eighth of the particles has Km = 0 and w_cov carries no Km variance, so their proposals keep it.  Four fused iterations with
    the cost order on and off: accept counts, failure counts, particles and likelihoods identical, nothing poisoned."""
    n = 32768
    priors = {"Vmax": {"dist": "uniform", "low": 0, "high": 10}, "Km": {"dist": "normal", "mu": 0.05, "sigma": 0.2},
              "sigma": {"dist": "uniform", "low": 0, "high": 10}}
    s = pkg.SMCSettings(n_particle=n, priors=priors)
    w_cov = s.w_cov()
    w_cov[1, :] = 0.0
    w_cov[:, 1] = 0.0
    rs = np.random.RandomState(21)
    th = np.column_stack([rs.uniform(0.5, 3.0, n), rs.uniform(0.01, 0.12, n), rs.uniform(0.5, 3.0, n)])
    th[: n // 8, 1] = 0.0
    res = {}
    for on in (True, False):
        with make_engine(pkg, data, n, priors=priors) as eng:
            eng.set_cost_order(on)
            eng.upload_particles(pkg.SMC_SET_PRED, th)
            assert eng.loglik(pkg.SMC_SET_PRED)["n_failed"] == 0
            lk0 = eng.download_lk(pkg.SMC_SET_PRED)
            assert np.isfinite(lk0).all()
            eng.upload_particles(pkg.SMC_SET_FILT, th)
            eng.upload_lk(pkg.SMC_SET_FILT, lk0)
            outs, props = [], []
            for j in range(4):
                outs.append(eng.mh_iteration_device_rng(0.05, 1.0, w_cov, 13, (5 << 16) | j, 0))
                props.append(eng.download_particles(pkg.SMC_SET_PRED))
            res[on] = ([o["accepted_now"] for o in outs], [o["n_failed"] for o in outs], eng.download_particles(pkg.SMC_SET_FILT),
                       eng.download_lk(pkg.SMC_SET_FILT), props)
    zero_km = res[True][4][0][:, 1] == 0.0
    assert zero_km.sum() == n // 8                                 # the case is really there: in-support proposals with Km == 0 ...
    assert (res[True][4][0][zero_km, 0] != th[zero_km, 0]).mean() > 0.5     # ... most of which moved (were not reset to the current point)
    assert res[True][0] == res[False][0] and res[True][1] == res[False][1] == [0, 0, 0, 0]
    assert np.array_equal(res[True][2], res[False][2]) and np.array_equal(res[True][3], res[False][3])
    assert np.isfinite(res[True][3]).all()                         # ... none of which ever poisoned a stored likelihood
    assert (res[True][2][: n // 8, 0] != th[: n // 8, 0]).sum() > n // 100      # and some of them were accepted


def test_item_records_of_a_sweep_add_up_to_its_counters(pkg, data):
    """smc_download_item_info (diagnostics, tools/sweep_tail_census.py): the per-(experiment, particle) records of a likelihood
    sweep - attempts | cancelled << 29 | failed << 30 - sum to the sweep's device-counted attempts; after a Metropolis sweep in
    cost order the records of out-of-support proposals (published by the propose kernel) are zero and cancelled solves are
    flagged, none failed."""
    n = 30000
    rs = np.random.RandomState(5)
    th = rs.uniform(0.05, 10, (n, 3))
    with make_engine(pkg, data, n) as eng:
        eng.upload_particles(pkg.SMC_SET_PRED, th)
        info = eng.loglik(pkg.SMC_SET_PRED)
        rec = eng.download_item_info()
        assert rec.shape == (data.t.shape[0], n) and int((rec & 0x1fffffff).sum()) == info["rk_attempts"]
        assert not (rec >> 29).any()
        eng.upload_particles(pkg.SMC_SET_FILT, th)
        eng.upload_lk(pkg.SMC_SET_FILT, eng.download_lk(pkg.SMC_SET_PRED))
        out = eng.mh_iteration_device_rng(0.01, 1.0, pkg.SMCSettings(n_particle=n).w_cov(), 3, 1, 0)
        rec = eng.download_item_info()
        prop = eng.download_particles(pkg.SMC_SET_PRED)
    att = rec & 0x1fffffff
    assert int(att.sum()) == out["rk_attempts"] and not ((rec >> 30) & 1).any()
    untouched = (prop == th).all(axis=1)                      # proposals reset to the current point: outside the prior's support
    assert untouched.sum() > n // 10 and not att[:, untouched].any()
    assert ((rec >> 29) & 1).sum() > 0                        # early rejection cancelled some solves of this prior-like population


def test_cost_ordered_handout_full_run_is_bit_identical(pkg, data):
    """A complete device-RNG run (early rejection on, as in the benchmark) with the cost order on and off."""
    n = 200_000
    runs = {}
    for on in (True, False):
        with make_engine(pkg, data, n) as eng:
            runs[on] = pkg.run_smc(eng, pkg.SMCSettings(n_particle=n, cost_order=on), rng="device", verbose=False, seed_device=41)
    a, b = runs[True], runs[False]
    assert [r_["gamma_new"] for r_ in a["records"]] == [r_["gamma_new"] for r_ in b["records"]]
    assert [r_["n_accept"] for r_ in a["records"]] == [r_["n_accept"] for r_ in b["records"]]
    assert np.array_equal(a["p_pred"], b["p_pred"]) and np.array_equal(a["lk"], b["lk"]) and a["logZ"] == b["logZ"]


@pytest.mark.parametrize("n_cand", [1, 5, 16, 17, 32])
def test_fused_ess_search_equals_max_plus_partials(pkg, data, n_cand):
    """smc_ess_search_global (maximum + up to 32 candidates, ONE synchronisation; the ESS passes read max(lk) from device
    memory) returns bit for bit what smc_max_lk_global followed by smc_ess_partials_global per 16 candidates returns."""
    n = 300_000
    rs = np.random.RandomState(n_cand)
    lk = -np.abs(rs.standard_normal(n)) * 300 + 250
    gms = 0.7 ** np.arange(n_cand)
    with make_engine(pkg, data, n) as eng:
        eng.upload_lk(pkg.SMC_SET_PRED, lk)
        m, sw, sw2 = eng.ess_search_global(gms)
        m_ref = eng.max_lk_global()
        ref = [eng.ess_partials_global(m_ref, gms[k:k + 16]) for k in range(0, n_cand, 16)]
        _, sw_again, _ = eng.ess_search_global(gms[:3], with_max=False)       # the maximum stays in place between calls
    assert m == m_ref == lk.max()
    assert np.array_equal(sw, np.concatenate([r[0] for r in ref])) and np.array_equal(sw2, np.concatenate([r[1] for r in ref]))
    assert np.array_equal(sw_again, sw[:3])
    w = np.exp((lk - lk.max())[None, :] * gms[:, None])
    assert np.allclose(sw, w.sum(axis=1), rtol=1e-12) and np.allclose(sw2, (w * w).sum(axis=1), rtol=1e-12)


def test_in_phase_waves_change_no_result(pkg, data):
    """smc_set_in_phase: homogeneous Metropolis sweeps let a wave wait for all of its lanes before the next hand-out
    (csrc/solve_sched.h: patience).  Scheduling only: four fused iterations on a posterior-like population (the second and
    later ones qualify: the previous sweep had no long item) and a complete run are bit-identical with the feature on and off."""
    n = 150_000
    s = pkg.SMCSettings(n_particle=n)
    w_cov = s.w_cov()
    rs = np.random.RandomState(31)
    th = np.array([1.2254, 0.5218, 0.02048]) + rs.standard_normal((n, 3)) * np.array([0.025, 0.0295, 0.00094])
    res = {}
    for on in (True, False):
        with make_engine(pkg, data, n) as eng:
            eng.set_in_phase(on)
            eng.upload_particles(pkg.SMC_SET_PRED, th)
            eng.loglik(pkg.SMC_SET_PRED)
            eng.upload_particles(pkg.SMC_SET_FILT, th)
            eng.upload_lk(pkg.SMC_SET_FILT, eng.download_lk(pkg.SMC_SET_PRED))
            outs = [eng.mh_iteration_device_rng(1.0, 1.0, w_cov, 9, (2 << 16) | j, 0) for j in range(4)]
            res[on] = ([o["accepted_now"] for o in outs], eng.download_particles(pkg.SMC_SET_FILT),     # (rk_attempts depends on timing: early rejection)
                       eng.download_lk(pkg.SMC_SET_FILT))
    assert res[True][0] == res[False][0]
    assert np.array_equal(res[True][1], res[False][1]) and np.array_equal(res[True][2], res[False][2])
    runs = {}
    for on in (True, False):
        with make_engine(pkg, data, 100_000) as eng:
            runs[on] = pkg.run_smc(eng, pkg.SMCSettings(n_particle=100_000, in_phase=on), rng="device", verbose=False, seed_device=29)
    a, b = runs[True], runs[False]
    assert [r_["gamma_new"] for r_ in a["records"]] == [r_["gamma_new"] for r_ in b["records"]]
    assert np.array_equal(a["p_pred"], b["p_pred"]) and np.array_equal(a["lk"], b["lk"]) and a["logZ"] == b["logZ"]


# ---------------------------------------------------------------------------------------------------
# A10 on the device: batches of Metropolis iterations with one host synchronisation (smc_mh_sweeps_device_rng)
# ---------------------------------------------------------------------------------------------------
def _same_run(a, b):
    assert a["gamma"] == b["gamma"] == 1.0 and a["step"] == b["step"]
    for ra, rb in zip(a["records"], b["records"]):
        assert ra["gamma_new"] == rb["gamma_new"] and ra["last_j"] == rb["last_j"] and ra["n_accept"] == rb["n_accept"]
        assert len(ra["mh"]) == len(rb["mh"]) == ra["last_j"] + 1
        for ma, mb in zip(ra["mh"], rb["mh"]):
            assert ma["mhstep_ratio"] == mb["mhstep_ratio"] and ma["accepted_now"] == mb["accepted_now"]
            assert ma["accepted_ever"] == mb["accepted_ever"] and np.array_equal(ma["cov_m"], mb["cov_m"])
    assert np.array_equal(a["p_pred"], b["p_pred"]) and np.array_equal(a["lk"], b["lk"]) and a["logZ"] == b["logZ"]
    assert a["stats"]["mutation_sweeps"] == b["stats"]["mutation_sweeps"]


@pytest.mark.parametrize("n,wide", [(20000, False), (200000, False), (20000, True)])
def test_device_side_mh_loop_control_is_bit_identical(pkg, data, n, wide):
    """VERDICT r3 item 3.  The Metropolis loop of a tempering step (Micmem_SMC_main.py:209-249) with its control on the device
    - iterations enqueued back to back, break / halving decided by the control kernel, one synchronisation per batch - against
    the loop with one call and one Python decision per iteration (mh_batch = 0): schedule, loop lengths, every iteration's
    cov_m, mhstep_ratio and accept counts, final particles, likelihoods and evidence bit-identical, whatever the batch size
    (1: a decision per synchronisation as before but taken on the device; 3: batches that end mid-loop and are continued;
    "auto"; 32: the whole loop speculated, most of its sweeps finding the loop ended).  `wide`: proposals far too wide
    (mhstep_factor 40), so that r_ac.sum() < r_threshold_min * N and the halving branch (:247-249) runs on the device too."""
    kw = dict(mhstep_factor=40.0, mhstep_factor_cov=40.0) if wide else {}
    runs = {}
    for mb in (0, 1, 3, "auto", 32):
        with make_engine(pkg, data, n) as eng:
            runs[mb] = pkg.run_smc(eng, pkg.SMCSettings(n_particle=n, mh_batch=mb, **kw), rng="device", verbose=False, seed_device=7)
    ref = runs[0]
    for mb in (1, 3, "auto", 32):
        _same_run(ref, runs[mb])
    steps, sweeps = ref["step"], ref["stats"]["mutation_sweeps"]
    assert ref["stats"]["mh_syncs"] == sweeps and runs[1]["stats"]["mh_syncs"] == sweeps
    assert runs[32]["stats"]["mh_syncs"] == steps                       # ONE synchronisation per tempering step
    assert runs[32]["stats"]["mh_noop_sweeps"] > 0 == ref["stats"]["mh_noop_sweeps"]
    assert steps <= runs["auto"]["stats"]["mh_syncs"] < sweeps
    if wide:
        ratios = [m["mhstep_ratio"] for r in ref["records"] for m in r["mh"]]
        assert min(ratios) < 1.0                                        # the halving branch was taken


def test_mh_batch_entry_point_logs_and_limits(pkg, data):
    """smc_mh_sweeps_device_rng on its own: thresholds that can never trigger -> all n_iter iterations run, not stopped,
    ratio unchanged; a stop threshold of 0 -> exactly one iteration runs, the rest find the loop ended and leave p_filt,
    lk1, the accept flags and the carried moments alone (a following per-iteration call continues as if they had never been
    enqueued); bad arguments are refused."""
    n = 30000
    s = pkg.SMCSettings(n_particle=n)
    w_cov = s.w_cov()
    th = mixed_particles(n, seed=4)

    def fresh():
        eng = make_engine(pkg, data, n)
        eng.upload_particles(pkg.SMC_SET_PRED, th)
        eng.loglik(pkg.SMC_SET_PRED)
        eng.upload_particles(pkg.SMC_SET_FILT, th)
        eng.upload_lk(pkg.SMC_SET_FILT, eng.download_lk(pkg.SMC_SET_PRED))
        return eng
    with fresh() as a, fresh() as b:
        ref = [a.mh_iteration_device_rng(0.3, 1.0, w_cov, 5, (2 << 16) | j, 0) for j in range(4)]
        out = b.mh_sweeps_device_rng(0.3, 1.0, w_cov, 5, 2 << 16, 4, 1e300, -1.0, 0)
        assert out["n_done"] == 4 and not out["stopped"] and out["ratio_next"] == 1.0
        for r, it in zip(ref, out["iterations"]):
            assert r["accepted_now"] == it["accepted_now"] and r["accepted_ever"] == it["accepted_ever"]
            assert np.array_equal(r["cov_m"], it["cov_m"]) and it["mhstep_ratio"] == 1.0 and it["n_failed"] == 0
        assert np.array_equal(a.download_particles(pkg.SMC_SET_FILT), b.download_particles(pkg.SMC_SET_FILT))
        assert np.array_equal(a.download_lk(pkg.SMC_SET_FILT), b.download_lk(pkg.SMC_SET_FILT))
        assert np.array_equal(a.download_accept_flags(), b.download_accept_flags())
    with fresh() as a, fresh() as b:
        ref = [a.mh_iteration_device_rng(0.3, 1.0, w_cov, 5, (2 << 16) | j, 0) for j in range(2)]
        out = b.mh_sweeps_device_rng(0.3, 1.0, w_cov, 5, 2 << 16, 6, 0.0, -1.0, 0)      # any accepted particle ends the loop
        assert out["n_done"] == 1 and out["stopped"] and len(out["iterations"]) == 1
        assert out["iterations"][0]["accepted_ever"] == ref[0]["accepted_ever"] > 0
        nxt = b.mh_iteration_device_rng(0.3, 1.0, w_cov, 5, (2 << 16) | 1, 0)            # the carried moments are intact
        assert nxt["accepted_now"] == ref[1]["accepted_now"] and np.array_equal(nxt["cov_m"], ref[1]["cov_m"])
        assert np.array_equal(a.download_particles(pkg.SMC_SET_FILT), b.download_particles(pkg.SMC_SET_FILT))
        halved = b.mh_sweeps_device_rng(0.3, 1.0, w_cov, 5, (2 << 16) | 2, 3, 1e300, 1e300, 0)   # every count is "too small"
        assert halved["n_done"] == 3 and [it["mhstep_ratio"] for it in halved["iterations"]] == [1.0, 0.5, 0.25]
        assert halved["ratio_next"] == 0.125
        with pytest.raises(pkg.SmcError):
            b.mh_sweeps_device_rng(0.3, 1.0, w_cov, 5, 0, 33, 0.0, 0.0, 0)
        with pytest.raises(pkg.SmcError):
            b.mh_sweeps_device_rng(0.3, 1.0, w_cov, 5, 0, 0, 0.0, 0.0, 0)


# ---------------------------------------------------------------------------------------------------
# round 4: resampling without a host synchronisation (one rank), page-locked result arrays
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("scheme", ["residual_systematic", "systematic", "multinomial"])
@pytest.mark.parametrize("pattern,first", [("spread", True), ("spread", False), ("underflow_tail", True), ("underflow_tail", False),
                                           ("one_hot", False)])
def test_enqueued_resampling_equals_the_synchronous_call(pkg, data, scheme, pattern, first):
    """smc_resample_enqueue (one rank: the offspring total stays on the device, one gather launch covers every output slot, no
    synchronisation) against smc_resample_global (two synchronisations, host-side plan) on the same weights: offspring counts,
    the logged totals and the WHOLE p_filt / lk1 block - gathered rows and the rows nobody writes (zeros in the first step, the
    previous p_pred rows later, Micmem_SMC_main.py:178-184) - identical, for the three schemes."""
    n = 5000
    rs = np.random.RandomState(17)
    lk = rs.standard_normal(n) * 4.0
    if pattern == "underflow_tail":
        lk = -np.arange(n, dtype=float) * 30.0      # almost every weight underflows: fewer offspring than particles is possible
    elif pattern == "one_hot":
        lk[:] = -1e6
        lk[n // 3] = 0.0
    p_pred = rs.standard_normal((n, 3))
    mx, gm, u = float(lk.max()), 0.7, 0.318
    sum_w = float(np.sum(np.exp((lk - mx) * gm)))
    s = pkg.SMCSettings(n_particle=n, resampling=scheme)
    es = {"max_lk": mx, "gm": gm, "sum_weight": sum_w}
    got = {}
    for defer in (False, True):
        with make_engine(pkg, data, n) as eng:
            eng.set_resampling(scheme)
            eng.upload_particles(pkg.SMC_SET_PRED, p_pred)
            eng.upload_lk(pkg.SMC_SET_PRED, lk)
            eng.upload_particles(pkg.SMC_SET_FILT, p_pred[::-1].copy())      # something recognisable where nobody writes
            eng.upload_lk(pkg.SMC_SET_FILT, lk[::-1].copy())
            out = pkg.resample(eng, pkg.SingleComm(), es, u, s, first_step=first, defer=defer)
            if defer:
                assert callable(out)
                out = out()
            got[defer] = (out, eng.download_offspring(), eng.download_particles(pkg.SMC_SET_FILT), eng.download_lk(pkg.SMC_SET_FILT))
    a, b = got[False], got[True]
    assert a[0] == b[0] and np.array_equal(a[1], b[1])
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    assert a[1].sum() == a[0]["n_offspring"] <= n


def test_pinned_result_arrays_outlive_their_engine_and_are_reused(pkg, data):
    """Final particles and likelihoods of run_smc arrive in page-locked host memory (one DMA transfer instead of the runtime's
    staged copy).  The buffer belongs to the array: it is valid after the engine is gone, equals an ordinary download, and goes
    back to the pool - to be reused by the next download of that size - only when the array and its views are garbage."""
    import gc
    from smc_lt_amd import engine as E
    n = 20000
    th = mixed_particles(n, seed=9)
    with make_engine(pkg, data, n) as eng:
        eng.upload_particles(pkg.SMC_SET_PRED, th)
        a = eng.download_particles(pkg.SMC_SET_PRED, pinned=True)
        b = eng.download_particles(pkg.SMC_SET_PRED)
        view = a[100:200]
        addr = a.__array_interface__["data"][0]
        c = eng.download_particles(pkg.SMC_SET_PRED, pinned=True)            # a is alive: another buffer
        assert c.__array_interface__["data"][0] != addr
    assert np.array_equal(a, th) and np.array_equal(b, th) and np.array_equal(c, th)      # after smc_destroy
    del a, c
    gc.collect()
    assert np.array_equal(view, th[100:200])                                  # the view keeps its buffer
    assert addr not in E._PINNED.free.get(n * 3 * 8, [])
    del view
    gc.collect()
    assert addr in E._PINNED.free.get(n * 3 * 8, [])
    with make_engine(pkg, data, n) as eng:
        eng.upload_particles(pkg.SMC_SET_PRED, th)
        d = eng.download_particles(pkg.SMC_SET_PRED, pinned=True)
        assert d.__array_interface__["data"][0] in (addr, ) or len(E._PINNED.free.get(n * 3 * 8, [])) >= 1
        assert np.array_equal(d, th)
    out = None
    with make_engine(pkg, data, n) as eng:
        out = pkg.run_smc(eng, pkg.SMCSettings(n_particle=n), rng="device", verbose=False, seed_device=2)
    assert np.isfinite(out["p_pred"]).all() and out["p_pred"].shape == (n, 3) and out["lk"].shape == (n,)
    out["p_pred"][0, 0] = 1.0                                                 # ordinary writable NumPy arrays


def test_pinned_pool_is_thread_safe_capped_and_releasable(pkg):
    """the pool of page-locked result buffers is shared by every engine of the process - several rank threads
    take and give concurrently - must not keep an unbounded amount of memory pinned, and can be emptied."""
    import gc
    import threading
    from smc_lt_amd import engine as E
    pkg.release_pinned_pool()
    assert E._PINNED.pooled == 0 and not any(E._PINNED.free.values())
    errs = []

    def work(seed):
        try:
            rs = np.random.RandomState(seed)
            for _ in range(200):
                a = E.pinned_empty((int(rs.choice([64, 256, 1024])), 3))
                a[:] = seed
                assert (a == seed).all()          # nobody else holds this buffer
                del a
        except Exception as ex:  # noqa: BLE001
            errs.append(ex)
    ths = [threading.Thread(target=work, args=(k,)) for k in range(6)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    gc.collect()
    assert not errs, errs[:1]
    assert 0 < E._PINNED.pooled <= E._PINNED.cap_bytes
    old_cap = E._PINNED.cap_bytes
    try:
        E._PINNED.cap_bytes = E._PINNED.pooled          # the pool is "full": a returned buffer is unpinned at once
        before = E._PINNED.pooled
        b = E.pinned_empty((5000, 3))
        del b
        gc.collect()
        assert E._PINNED.pooled == before
    finally:
        E._PINNED.cap_bytes = old_cap
    pkg.release_pinned_pool()
    assert E._PINNED.pooled == 0
