"""user-written models compiled at run time (hiprtc) into the likelihood kernel.
CPU part: the sources compile for gfx950 (no device needed); GPU part: the Michaelis-Menten model written as a user
model reproduces the built-in kernel, and a two-state model follows SciPy's solve_ivp(RK45)."""
import ctypes

import numpy as np
import pytest


def _check(pkg, src, ns, dim):
    log = ctypes.create_string_buffer(8192)
    rc = pkg.lib().smc_user_model_check(src.encode(), ns, dim, log, 8192)
    return rc, log.value.decode(errors="replace")


def test_example_sources_compile_for_gfx950(pkg):
    for src, ns in ((pkg.user_models.MICHAELIS_MENTEN, 1), (pkg.user_models.MICHAELIS_MENTEN_PLAIN, 1),
                    (pkg.user_models.CONSECUTIVE_REACTIONS, 2)):
        rc, log = _check(pkg, src, ns, 3)
        assert rc == 0, log
    assert "smc_user_cost" in pkg.user_models.MICHAELIS_MENTEN and "smc_user_cost" not in pkg.user_models.MICHAELIS_MENTEN_PLAIN
    # a source that names the cost hint must define it (include/smc_hip.h)
    rc, log = _check(pkg, pkg.user_models.MICHAELIS_MENTEN_PLAIN + "// smc_user_cost: to do\n", 1, 3)
    assert rc == 1 and "smc_user_cost" in log


def test_broken_source_reports_the_compiler_log(pkg):
    rc, log = _check(pkg, "__device__ void smc_user_y0(const double*, const double*, double* y) { y[0] = undefined_name; }", 1, 3)
    assert rc == 1 and "undefined_name" in log
    rc, _ = _check(pkg, pkg.user_models.MICHAELIS_MENTEN, 99, 3)          # more states than SMC_USER_MAX_STATES
    assert rc == 2


@pytest.mark.gpu
def test_mm_as_user_model_equals_builtin_kernel(pkg, data):
    """Same particles through the built-in MM kernel (pinned to the reference) and through the user-model path:
    the generic RK45 is written in the same operation order for one state, so logL agrees to rounding."""
    n = 4096
    rs = np.random.RandomState(2)
    th = np.array([1.2254, 0.5218, 0.02048]) + rs.standard_normal((n, 3)) * np.array([0.05, 0.05, 0.002])
    th[: n // 2] = rs.uniform(0.05, 10, size=(n // 2, 3))
    with pkg.HipEngine(n, 3, device=0) as eng:
        eng.set_prior(pkg.SMCSettings().priors)
        eng.set_model_mm(data.t, data.P_obs, data.S0)
        eng.upload_particles(pkg.SMC_SET_PRED, th)
        info_a = eng.loglik(pkg.SMC_SET_PRED)
        lk_a = eng.download_lk(pkg.SMC_SET_PRED)
        eng.set_model_user(pkg.user_models.MICHAELIS_MENTEN, 1, data.t, data.P_obs, cond=np.asarray(data.S0)[:, None])
        info_b = eng.loglik(pkg.SMC_SET_PRED)
        lk_b = eng.download_lk(pkg.SMC_SET_PRED)
    assert info_a["n_failed"] == 0 and info_b["n_failed"] == 0
    assert info_a["rk_attempts"] == info_b["rk_attempts"]                  # the same step sequence, attempt for attempt
    assert np.max(np.abs(lk_a - lk_b) / np.maximum(1.0, np.abs(lk_a))) < 1e-10   # sum of squares: sequential here, NumPy-pairwise in the built-in kernel


@pytest.mark.gpu
@pytest.mark.parametrize("n", [5, 64, 3000, 20000])
def test_cost_hint_changes_the_order_of_the_solves_and_no_result(pkg, data, n):
    """smc_user_cost (include/smc_hip.h): the hinted model hands its long solves out first and runs the longest one per wave;
    likelihoods, attempt totals, and a Metropolis sweep with exact early rejection (accept flags, particles) must equal the
    plain model's bit for bit - for a grid of a few waves as for a full one.  The population reaches Vmax/Km = 5000, so both
    lists are in use; half of the proposals leave the prior's support (masked, never listed).  From 16 384 particles on the
    Metropolis sweeps of the hinted model are also cost-ordered and in phase (smc_set_cost_order: classes from the hint,
    out-of-support proposals published by the scan kernel): the largest case runs that, and once more with it switched off."""
    rs = np.random.RandomState(n)
    th = np.column_stack([rs.uniform(0.5, 10, n), 10.0 ** rs.uniform(-3.3, 1, n), rs.uniform(0.01, 1, n)])
    th[0] = (9.0, 0.002, 0.5)                                              # 4500: certainly a solo solve
    out = {}
    for name, src in (("plain", pkg.user_models.MICHAELIS_MENTEN_PLAIN), ("hint", pkg.user_models.MICHAELIS_MENTEN),
                      ("hint, lists off", pkg.user_models.MICHAELIS_MENTEN), ("hint, cost order off", pkg.user_models.MICHAELIS_MENTEN)):
        with pkg.HipEngine(n, 3, device=0) as eng:
            eng.set_prior(pkg.SMCSettings().priors)
            eng.set_model_user(src, 1, data.t, data.P_obs, cond=np.asarray(data.S0)[:, None])
            if name == "hint, lists off":
                eng.set_stiff_first(False)
            if name == "hint, cost order off":
                eng.set_cost_order(False)
            eng.upload_particles(pkg.SMC_SET_PRED, th)
            info = eng.loglik(pkg.SMC_SET_PRED)
            lk = eng.download_lk(pkg.SMC_SET_PRED)
            eng.upload_particles(pkg.SMC_SET_FILT, th)
            eng.upload_lk(pkg.SMC_SET_FILT, lk)
            eng.set_early_reject(False)
            mh0 = eng.mh_step_device_rng(0.05, 1.0, np.diag([2.0, 0.5, 0.1]), 7, 3)
            p0, l0 = eng.download_particles(pkg.SMC_SET_FILT), eng.download_lk(pkg.SMC_SET_FILT)
            eng.upload_particles(pkg.SMC_SET_FILT, th)
            eng.upload_lk(pkg.SMC_SET_FILT, lk)
            eng.set_early_reject(True)
            mh1 = eng.mh_step_device_rng(0.05, 1.0, np.diag([2.0, 0.5, 0.1]), 7, 3)
            p1, l1 = eng.download_particles(pkg.SMC_SET_FILT), eng.download_lk(pkg.SMC_SET_FILT)
        assert info["n_failed"] == 0 and mh0["n_failed"] == 0
        assert np.array_equal(p0, p1) and np.array_equal(l0, l1) and mh0["accepted_now"] == mh1["accepted_now"]
        assert mh1["rk_attempts"] <= mh0["rk_attempts"]
        out[name] = (info["rk_attempts"], lk, mh0["accepted_now"], mh0["rk_attempts"], p0, l0)
    for name in ("hint", "hint, lists off", "hint, cost order off"):
        assert out[name][0] == out["plain"][0] and out[name][2] == out["plain"][2] and out[name][3] == out["plain"][3], name
        assert np.array_equal(out[name][1], out["plain"][1]) and np.array_equal(out[name][4], out["plain"][4]), name
        assert np.array_equal(out[name][5], out["plain"][5]), name


@pytest.mark.gpu
def test_two_state_user_model_follows_scipy_and_runs_the_loop(pkg):
    """A -> B -> C observed through B: logL against scipy.integrate.solve_ivp(RK45) + the reference's Gaussian
    likelihood on the host, then a complete tempering run on the device recovers the generating constants."""
    from scipy.integrate import solve_ivp
    rs = np.random.RandomState(0)
    n_ex, n_t = 4, 30
    t = np.tile(np.linspace(0.0, 10.0, n_t), (n_ex, 1))
    A0 = np.array([1.0, 2.0, 0.5, 1.5])
    k_true, sig_true = (0.8, 0.3), 0.01

    def model(k1, k2, a0, tt):
        sol = solve_ivp(lambda _t, y: [-k1 * y[0], k1 * y[0] - k2 * y[1]], [tt[0], tt[-1]], [a0, 0.0], method="RK45",
                        t_eval=tt, rtol=1e-3, atol=1e-6)
        return sol.y[1]
    obs = np.array([model(*k_true, A0[e], t[e]) for e in range(n_ex)]) + sig_true * rs.standard_normal((n_ex, n_t))
    n = 256
    th = np.column_stack([rs.uniform(0.1, 2, n), rs.uniform(0.05, 1, n), rs.uniform(0.005, 0.05, n)])
    ref = np.empty(n)
    for i, (k1, k2, sg) in enumerate(th):
        r2 = sum(np.sum((obs[e] - model(k1, k2, A0[e], t[e])) ** 2) for e in range(n_ex))
        ref[i] = n_ex * (-0.5 * n_t) * np.log(2 * np.pi * sg * sg) - r2 / (2 * sg * sg)
    priors = {"k1": {"dist": "uniform", "low": 0, "high": 3}, "k2": {"dist": "uniform", "low": 0, "high": 3},
              "sigma": {"dist": "uniform", "low": 0, "high": 1}}
    with pkg.HipEngine(n, 3, device=0) as eng:
        eng.set_prior(priors)
        eng.set_model_user(pkg.user_models.CONSECUTIVE_REACTIONS, 2, t, obs, cond=A0[:, None])
        eng.upload_particles(pkg.SMC_SET_PRED, th)
        eng.loglik(pkg.SMC_SET_PRED)
        lk = eng.download_lk(pkg.SMC_SET_PRED)
    assert np.max(np.abs(lk - ref) / np.maximum(1.0, np.abs(ref))) < 1e-6
    n = 8192
    s = pkg.SMCSettings(n_particle=n, priors=priors)
    with pkg.HipEngine(n, 3, device=0) as eng:
        eng.set_prior(priors)
        eng.set_model_user(pkg.user_models.CONSECUTIVE_REACTIONS, 2, t, obs, cond=A0[:, None])
        out = pkg.run_smc(eng, s, rng="device", verbose=False, seed_device=3)
    assert out["gamma"] == 1.0
    m, sd = out["p_pred"].mean(axis=0), out["p_pred"].std(axis=0)
    assert np.all(np.abs(m[:2] - np.array(k_true)) < 5 * sd[:2] + 0.02) and abs(m[2] - sig_true) < 0.004


@pytest.mark.gpu
def test_user_model_compile_error_surfaces_through_the_engine(pkg):
    with pkg.HipEngine(64, 3, device=0) as eng:
        with pytest.raises(pkg.SmcError, match="does not compile"):
            eng.set_model_user("this is not HIP", 1, np.zeros((1, 4)), np.zeros((1, 4)))


CHAIN8 = r"""
// eight first-order steps in a row, y0 -> y1 -> ... -> y7, rate constants theta[0] (even links) / theta[1] (odd links)
__device__ void smc_user_y0(const double *theta, const double *cond, double *y) {
    y[0] = cond[0];
    for (int i = 1; i < 8; ++i) y[i] = 0.0;
}
__device__ void smc_user_rhs(double t, const double *y, const double *theta, const double *cond, double *dydt) {
    for (int i = 0; i < 8; ++i) {
        const double k_in = theta[(i + 1) & 1], k_out = theta[i & 1];
        dydt[i] = ((i > 0) ? k_in * y[i - 1] : 0.0) - ((i < 7) ? k_out * y[i] : 0.0);
    }
}
__device__ double smc_user_obs(double t, const double *y, const double *theta, const double *cond) { return y[7]; }
"""

DIVERGING = r"""
// the right-hand side turns NaN after t = 1: the step size collapses and the solve must be reported as failed
__device__ void smc_user_y0(const double *theta, const double *cond, double *y) { y[0] = 1.0; }
__device__ void smc_user_rhs(double t, const double *y, const double *theta, const double *cond, double *dydt) {
    dydt[0] = (t > 1.0) ? sqrt(-1.0 - y[0] * y[0]) : -theta[0] * y[0];
}
__device__ double smc_user_obs(double t, const double *y, const double *theta, const double *cond) { return y[0]; }
"""


@pytest.mark.gpu
def test_eight_state_user_model_follows_scipy(pkg):
    """SMC_USER_MAX_STATES = 8 states: the generic kernel against solve_ivp(RK45) on the host."""
    from scipy.integrate import solve_ivp
    rs = np.random.RandomState(1)
    n_ex, n_t, n = 2, 25, 64
    t = np.tile(np.linspace(0.0, 20.0, n_t), (n_ex, 1))
    A0 = np.array([1.0, 3.0])
    obs = rs.uniform(0, 1, (n_ex, n_t))
    th = np.column_stack([rs.uniform(0.2, 2, n), rs.uniform(0.2, 2, n), rs.uniform(0.05, 0.5, n)])

    def rhs(_t, y, ka, kb):
        d = np.zeros(8)
        for i in range(8):
            k_in, k_out = (kb, ka) if i % 2 == 0 else (ka, kb)
            d[i] = (k_in * y[i - 1] if i > 0 else 0.0) - (k_out * y[i] if i < 7 else 0.0)
        return d
    ref = np.empty(n)
    for i, (ka, kb, sg) in enumerate(th):
        r2 = 0.0
        for e in range(n_ex):
            y0 = np.zeros(8)
            y0[0] = A0[e]
            sol = solve_ivp(rhs, [t[e, 0], t[e, -1]], y0, method="RK45", t_eval=t[e], rtol=1e-3, atol=1e-6, args=(ka, kb))
            r2 += np.sum((obs[e] - sol.y[7]) ** 2)
        ref[i] = n_ex * (-0.5 * n_t) * np.log(2 * np.pi * sg * sg) - r2 / (2 * sg * sg)
    priors = {"ka": {"dist": "uniform", "low": 0, "high": 3}, "kb": {"dist": "uniform", "low": 0, "high": 3},
              "sigma": {"dist": "uniform", "low": 0, "high": 1}}
    with pkg.HipEngine(n, 3, device=0) as eng:
        eng.set_prior(priors)
        eng.set_model_user(CHAIN8, 8, t, obs, cond=A0[:, None])
        eng.upload_particles(pkg.SMC_SET_PRED, th)
        info = eng.loglik(pkg.SMC_SET_PRED)
        lk = eng.download_lk(pkg.SMC_SET_PRED)
    assert info["n_failed"] == 0
    assert np.max(np.abs(lk - ref) / np.maximum(1.0, np.abs(ref))) < 1e-6


@pytest.mark.gpu
def test_user_model_failure_is_counted_and_the_kernel_returns(pkg):
    n = 256
    t = np.linspace(0.0, 2.0, 10)[None, :]
    th = np.column_stack([np.full(n, 0.5), np.full(n, 1.0), np.full(n, 0.1)])
    with pkg.HipEngine(n, 3, device=0) as eng:
        eng.set_prior(pkg.SMCSettings().priors)
        eng.set_model_user(DIVERGING, 1, t, np.zeros_like(t))
        eng.upload_particles(pkg.SMC_SET_PRED, th)
        info = eng.loglik(pkg.SMC_SET_PRED)
    assert info["n_failed"] == n
