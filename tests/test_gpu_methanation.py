"""GPU parity (-m gpu): the methanation kernels built so far, through the C ABI, against the reference's
own function values (golden fixture) and the pinned oracle.  FP tolerance 1e-11 relative: device exp /
sqrt / reciprocal-square vs libm pow/exp, FMA contraction."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-11


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "methanation_golden.npz"))


@pytest.fixture(scope="module")
def M():
    import __graft_entry__ as g
    g.load_oracle()
    from oracle import methanation
    return methanation


def test_residual_vs_reference_values(pkg, gold):
    res = pkg.methanation.reaction(gold["res_X"], gold["res_dX"], gold["res_p"])
    ref = gold["res_out"]
    assert (np.abs(res - ref) / np.maximum(1.0, np.abs(ref))).max() < TOL
    assert np.array_equal(res[:, [0, 51, 102, 153, 204, 255, 306]], ref[:, [0, 51, 102, 153, 204, 255, 306]])
    assert np.array_equal(res[:, [50, 101, 152, 203, 254, 305, 356]], ref[:, [50, 101, 152, 203, 254, 305, 356]])


def test_residual_large_batch_vs_oracle(pkg, M, gold):
    rs = np.random.RandomState(0)
    n = 3000
    idx = rs.randint(0, len(gold["res_X"]), n)
    X = gold["res_X"][idx] * (1 + 0.05 * rs.uniform(-1, 1, (n, 357)))
    dX = rs.standard_normal((n, 357)) * np.abs(X) * 0.01
    p = gold["res_p"][idx] * (1 + 0.05 * rs.uniform(-1, 1, (n, 18)))
    ref = M.reaction(X, dX, p)
    res = pkg.methanation.reaction(X, dX, p)
    # each equation is a sum of convection / diffusion / reaction terms of magnitude up to ~1e9 that largely
    # cancel for these random (non-solution) states: the error is measured against the largest residual of
    # the same field in the same state (the term scale), not against the cancelled result
    err = np.abs(res - ref).reshape(n, 7, 51)
    scale = np.maximum(1.0, np.abs(ref).reshape(n, 7, 51).max(axis=2, keepdims=True))
    worst = (err / scale).max()
    assert worst < TOL, worst


def test_rate_law(pkg, gold):
    i = gold["rc_in"]
    got = pkg.methanation.func_rCH4(i[:, 0], i[:, 1], i[:, 2], i[:, 3], i[:, 4], gold["rc_par"])
    assert np.allclose(got, gold["rc_out"], rtol=TOL, atol=0)


def test_loglike(pkg, gold):
    nd = int(gold["n_data"])
    for j in range(8):
        got = pkg.methanation.my_loglike(gold["ll_y"][j], gold["ll_d"][j], gold["ll_s"][j], nd)
        assert abs(got - gold["ll_out"][j]) <= TOL * abs(gold["ll_out"][j])
    y = np.arange(150.).reshape(5, 30)
    assert abs(pkg.methanation.my_loglike(y, y + 1, 5.0, 30) - (-244.41568686511505)) < 1e-12
    # batch: one data set, many particles with their own sigma
    rs = np.random.RandomState(1)
    ys = gold["ll_y"][0][None] + rs.standard_normal((500, 5, nd))
    sig = rs.uniform(1, 9, 500)
    got = pkg.methanation.my_loglike(ys, gold["ll_d"][0], sig, nd)
    ref = np.array([np.sum([-(0.5 / s ** 2) * np.sum((y[i] - gold["ll_d"][0][i]) ** 2) - nd * np.log(s) for i in range(5)])
                    for y, s in zip(ys, sig)])
    assert np.allclose(got, ref, rtol=1e-12, atol=0)


# ---------------------------------------------------------------------------------------------------
# K8: DAE time integration (PARITY UNPINNED against the reference's IDA; compared with the oracle's
# implementation of the same algorithm and checked for physical consistency)
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def cond_guess(M):
    cond = M.load_conditions(os.path.join(GOLD, "methanation_information.csv"))
    return cond, M.initial_guess(cond)


def test_dae_batch_vs_oracle(pkg, M, cond_guess):
    """30 experiments x 3 parameter vectors.  Tolerance: the oracle builds its iteration matrix by finite
    differences and solves with a pivoted banded LU, the kernel uses the analytic matrix and a block LU, so the
    Newton iterates differ below the Newton tolerance; both integrate to rtol = atol = 1e-6.  Outlet values
    must agree within 20 tolerance units atol + rtol*|y| (observed: < 1e-3 units at the base parameters)."""
    cond, guess = cond_guess
    lo, hi, pos = M.prior_box()
    rs = np.random.RandomState(3)
    prs = [M.BASEPARAMS.copy()]
    for _ in range(2):
        pr = M.BASEPARAMS.copy()
        pr[:4] = (lo[pos] + (hi[pos] - lo[pos]) * rs.uniform(0.2, 0.8, 5))[:4]
        prs.append(pr)
    p0 = np.array([M.p0_tuple(cond, i, pr) for pr in prs for i in range(30)])
    y0 = np.array([guess[i] for pr in prs for i in range(30)])
    flows, status, states, info = pkg.methanation.dae_solve_batch(p0, y0, want_states=True)
    assert np.all(status == 0)
    outlet = [50, 101, 152, 203, 254, 305, 356]
    worst = 0.0
    for k in range(len(p0)):
        yo, rc, st = M.dae_solve(y0[k], p0[k])
        assert rc == 0
        units = np.abs(states[k] - yo)[outlet] / (1e-6 + 1e-6 * np.abs(yo[outlet]))
        worst = max(worst, units.max())
    assert worst < 20, worst
    # flows follow from the outlet state by my_model's mapping (:204-208)
    fo, _, _ = M.my_model(prs[0], cond, guess)
    assert np.allclose(flows[:30].T, fo, rtol=1e-4, atol=1e-3)
    assert 150 * len(p0) < info["steps"] < 1000 * len(p0)


def test_dae_kernel_converges_under_tolerance_refinement(pkg, M, cond_guess):
    """K8 is parity-unpinned (no IDA): what can be shown without it is that the kernel's answer is the DAE's - the
    outlet state at rtol = atol = 1e-6 lies within a few tolerance units of the one at 1e-8 and the one at 1e-7 several
    times closer (the global error scales with the tolerance); 60 solves,
    base parameters and one parameter vector from inside the prior box."""
    cond, guess = cond_guess
    lo, hi, pos = M.prior_box()
    pr = M.BASEPARAMS.copy()
    pr[:4] = (lo[pos] + (hi[pos] - lo[pos]) * np.random.RandomState(3).uniform(0.2, 0.8, 5))[:4]
    p0 = np.concatenate([pkg.methanation.p0_rows(cond, M.BASEPARAMS), pkg.methanation.p0_rows(cond, pr)])
    y0 = np.concatenate([guess[:30], guess[:30]])
    outlet = [50, 101, 152, 203, 254, 305, 356]
    sol, ok = {}, np.ones(len(p0), dtype=bool)
    for tol in (1e-6, 1e-7, 1e-8):
        _, status, states, _ = pkg.methanation.dae_solve_batch(p0, y0, rtol=tol, atol=tol, want_states=True)
        ok &= status == 0              # the attempt budget (3000) is fixed: the most dynamic solves exceed it at 1e-8
        sol[tol] = states[:, outlet]
    assert ok[:30].all() and ok.sum() >= 40
    scale = 1.0 + np.abs(sol[1e-8][ok])
    e6 = np.abs(sol[1e-6][ok] - sol[1e-8][ok]) / scale
    e7 = np.abs(sol[1e-7][ok] - sol[1e-8][ok]) / scale
    assert e6.max() < 2e-5 and e7.max() < 3e-6, (e6.max(), e7.max())
    assert e7.max() < 0.35 * e6.max()


def test_dae_physics_and_failure_sentinel(pkg, M, cond_guess):
    cond, guess = cond_guess
    p0 = np.array([M.p0_tuple(cond, i, M.BASEPARAMS) for i in range(30)])
    flows, status, _, info = pkg.methanation.dae_solve_batch(p0, guess)
    info_tab = np.loadtxt(os.path.join(GOLD, "methanation_information.csv"), delimiter=",", skiprows=1)[:30]
    assert np.all(status == 0) and np.all(flows > 0)
    assert np.allclose(flows[:, 1] + flows[:, 2], info_tab[:, 11], rtol=2e-2)   # carbon balance
    assert np.allclose(flows[:, 4], info_tab[:, 15], rtol=2e-2)                 # argon is inert
    # an unsolvable case (NaN kinetic parameter) must come back as the reference's sentinel -10000
    bad = p0[:1].copy()
    bad[0, 10] = np.nan
    f2, st2, _, _ = pkg.methanation.dae_solve_batch(bad, guess[:1])
    assert st2[0] != 0 and np.all(f2 == -10000.0)


# ---------------------------------------------------------------------------------------------------
# the methanation model inside the SMC loop (config 4 at reduced N)
# ---------------------------------------------------------------------------------------------------
def _meth_settings(pkg, M, n):
    lo, hi, pos = M.prior_box()
    names = ["Af", "Eaf", "Ar", "Ear", "sigma"]
    priors = {nm: {"dist": "uniform", "low": float(lo[i]), "high": float(hi[i])} for nm, i in zip(names, pos)}
    return pkg.SMCSettings(n_particle=n, priors=priors, seed=20250205), pos


def test_two_wave_kernel_equals_the_one_wave_kernel_bit_for_bit(pkg, M, cond_guess, monkeypatch):
    """K8 v4 (csrc/meth_dae_split.h: integrator wave + chain-server wave per solve, the default) against K8 v3 (one wave per
    solve, SMC_K8_SPLIT=0).  The split changes who computes what and when, not one floating-point operation: states, flows,
    status and every counter of 16 prior-box parameter vectors x 30 experiments - some of them failing solves that run
    their whole attempt budget - must be IDENTICAL; so must a likelihood sweep through the resident path."""
    cond, guess = cond_guess
    lo, hi, pos = M.prior_box()
    rs = np.random.RandomState(11)
    prs = np.tile(M.BASEPARAMS, (16, 1))
    prs[:, :4] = (lo[pos] + (hi[pos] - lo[pos]) * rs.uniform(0, 1, (16, 5)))[:, :4]
    p0 = np.array([M.p0_tuple(cond, i, pr) for pr in prs for i in range(30)])
    y0 = np.array([guess[i] for pr in prs for i in range(30)])
    out = {}
    obs = None
    for split in ("1", "0"):
        monkeypatch.setenv("SMC_K8_SPLIT", split)      # read by the library at every launch
        flows, status, states, info = pkg.methanation.dae_solve_batch(p0, y0, want_states=True)
        info.pop("kernel_ms")
        if obs is None:      # observations: the flows of the base parameters (any fixed 5 x 30 table would do)
            base = np.array([M.p0_tuple(cond, i, M.BASEPARAMS) for i in range(30)])
            obs = pkg.methanation.dae_solve_batch(base, np.array([guess[i] for i in range(30)]))[0].T.copy()
        eng, s = _meth_engine(pkg, M, cond, guess, obs, 16)
        with eng:
            theta = lo[pos] + (hi[pos] - lo[pos]) * np.random.RandomState(12).uniform(0, 1, (16, 5))
            eng.upload_particles(pkg.SMC_SET_PRED, theta)
            eng.loglik(pkg.SMC_SET_PRED)
            lk = eng.download_lk(pkg.SMC_SET_PRED)
        out[split] = (flows, status, states, info, lk)
    a, b = out["1"], out["0"]
    assert (a[1] != 0).sum() == (b[1] != 0).sum() and np.array_equal(a[1], b[1])
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2], equal_nan=True)
    assert a[3] == b[3] and a[3]["steps"] > 150 * len(p0)
    assert np.array_equal(a[4], b[4], equal_nan=True) and np.isfinite(a[4]).sum() >= 8


def _meth_engine(pkg, M, cond, guess, obs, n):
    s, pos = _meth_settings(pkg, M, n)
    eng = pkg.HipEngine(n, 5, device=0)
    eng.set_model_methanation(cond, guess, obs, np.append(M.BASEPARAMS, M.SIGMA_TRUE), pos)
    eng.set_prior(s.priors)
    return eng, s


def test_methanation_loglik_sweep_and_mh_step_vs_oracle(pkg, M, cond_guess):
    """sim_particle on the resident set and one Metropolis iteration (host-RNG mode) against the oracle's
    my_model + my_loglike + cal_prior.  logL tolerance 1e-3 relative: K8 is compared with the oracle's
    implementation of the same integrator (tolerance-level differences of the flows, sigma ~ 5)."""
    cond, guess = cond_guess
    rs = np.random.RandomState(4)
    flows0, _, _ = M.my_model(M.BASEPARAMS, cond, guess)
    obs = flows0 + 5.0 * rs.standard_normal(flows0.shape)
    n = 6
    eng, s = _meth_engine(pkg, M, cond, guess, obs, n)
    lo, hi, pos = M.prior_box()
    theta = lo[pos] + (hi[pos] - lo[pos]) * rs.uniform(0.25, 0.75, (n, 5))
    theta[0] = np.append(M.BASEPARAMS, 5.0)[pos]

    def oracle_lk(th):
        out = np.empty(len(th))
        for k, row in enumerate(th):
            full = np.append(M.BASEPARAMS, M.SIGMA_TRUE).copy()
            full[pos] = row
            f, _, _ = M.my_model(full[:8], cond, guess)
            out[k] = M.loglike(f, obs, full[8], 30)
        return out
    with eng:
        eng.upload_particles(pkg.SMC_SET_PRED, theta)
        eng.upload_particles(pkg.SMC_SET_FILT, theta)
        info = eng.loglik(pkg.SMC_SET_PRED)
        lk = eng.download_lk(pkg.SMC_SET_PRED)
        ref = oracle_lk(theta)
        assert np.all(np.abs(lk - ref) <= 1e-3 * np.maximum(1.0, np.abs(ref))), (lk, ref)
        assert info["rk_attempts"] > 150 * n * 30
        k8 = eng.meth_sweep_counters()                                       # device-counted work of that sweep
        assert k8["bdf_steps"] == info["rk_attempts"] and k8["failed_solves"] == 0
        # round 5, IDA's control policy (csrc/meth_dae_elem.h: SMC_K8_POLICY 1): the carried convergence rate lets most steps
        # converge in ONE Newton iteration (bdf.py needed >= 2), the matrix survives a drift of cj (one factorisation per 5-8 steps)
        assert k8["bdf_steps"] <= k8["newton_iters"] < 3 * k8["bdf_steps"]
        assert 0.03 * k8["bdf_steps"] < k8["factorisations"] < 0.5 * k8["bdf_steps"]
        # one MH iteration with host-drawn noise; proposals 2 and 4 are pushed out of the prior box
        eng.set_debug_capture(True)
        eng.upload_lk(pkg.SMC_SET_FILT, lk)
        noise = rs.standard_normal((n, 5)) * (hi[pos] - lo[pos]) * 0.02
        noise[2, 0] = 1e9
        noise[4, 4] = -1e9
        rr = rs.uniform(0, 1, n)
        gamma = 0.05
        out = eng.mh_step_host_rng(gamma, 1.0, noise, rr)
        prop, lk2, p0, r = eng.download_debug_proposals()
        f = eng.download_particles(pkg.SMC_SET_FILT)
        p_ref = theta + noise * 1.0
        p0_ref = np.int32(M.cal_prior(p_ref) > 0)
        p_ref = p_ref * p0_ref[:, None] + theta * (1.0 - p0_ref[:, None])
        assert np.array_equal(p0, p0_ref) and p0_ref.tolist() == [1, 1, 0, 1, 0, 1]
        assert np.array_equal(prop, p_ref)
        lk2_ref = oracle_lk(p_ref)
        live = p0_ref == 1
        assert np.all(np.abs(lk2[live] - lk2_ref[live]) <= 1e-3 * np.maximum(1.0, np.abs(lk2_ref[live])))
        assert np.array_equal(lk2[~live], lk[~live])                      # masked: lk2 is the stored lk1
        with np.errstate(over="ignore"):
            r_ref = np.int32(np.exp((lk2 - lk) * gamma) * p0_ref >= rr)   # decisions from the device's own lk2
        assert np.array_equal(r, r_ref) and out["accepted_now"] == int(r_ref.sum())
        assert np.array_equal(f, p_ref * r_ref[:, None] + theta * (1.0 - r_ref[:, None]))


def test_complete_run_is_the_same_with_either_k8_kernel(pkg, M, cond_guess, monkeypatch):
    """A complete adaptive-tempering run on the device (N = 96, exact early rejection on, experiments in misfit order) with the
    two-wave K8 kernel and with the one-wave kernel: tempering schedule, accept counts and Metropolis lengths per step, final
    particles, likelihoods and log-evidence are identical to the last bit.  (Which solves the early rejection cancels depends
    on the order in which waves finish; the accepted proposals do not - so the solved / cancelled counts may differ.)"""
    cond, guess = cond_guess
    base = np.array([M.p0_tuple(cond, i, M.BASEPARAMS) for i in range(30)])
    flows0 = pkg.methanation.dae_solve_batch(base, np.array([guess[i] for i in range(30)]))[0].T.copy()
    obs = flows0 + 5.0 * np.random.RandomState(77).standard_normal(flows0.shape)
    outs = {}
    for split in ("1", "0"):
        monkeypatch.setenv("SMC_K8_SPLIT", split)
        eng, s = _meth_engine(pkg, M, cond, guess, obs, 96)
        with eng:
            outs[split] = pkg.run_smc(eng, s, rng="device", verbose=False, seed_device=9)
    a, b = outs["1"], outs["0"]
    assert a["gamma"] == 1.0 and a["step"] == b["step"] >= 3
    for key in ("gamma_new", "n_accept", "last_j"):
        assert [r[key] for r in a["records"]] == [r[key] for r in b["records"]], key
    assert np.array_equal(a["p_pred"], b["p_pred"]) and np.array_equal(a["lk"], b["lk"]) and a["logZ"] == b["logZ"]
    assert a["stats"]["dae_solves"] + a["stats"]["dae_solves_cancelled"] == b["stats"]["dae_solves"] + b["stats"]["dae_solves_cancelled"]


def test_methanation_full_smc_run_recovers_parameters(pkg, M, cond_guess):
    """Config 4 at reduced size: synthetic observations = model at baseparams + sigma = 5 noise
    (SMC_methanation_main.py:89-101), N = 192 particles, adaptive tempering to gamma = 1 on the device.
    The posterior must concentrate around the generating parameters (a statistical check: the integrator is
    parity-unpinned and N is small)."""
    cond, guess = cond_guess
    np.random.seed(20250205)
    flows0, _, _ = M.my_model(M.BASEPARAMS, cond, guess)
    obs = flows0.copy()
    for i in range(5):
        obs[i, :] = 1.0 * 5 * np.random.standard_normal(30) + obs[i, :]     # :94-95
    n = 192
    eng, s = _meth_engine(pkg, M, cond, guess, obs, n)
    with eng:
        out = pkg.run_smc(eng, s, rng="device", verbose=False, seed_device=5)
    assert out["gamma"] == 1.0 and 3 <= out["step"] <= 40
    post = out["p_pred"]
    lo, hi, pos = M.prior_box()
    truth = np.append(M.BASEPARAMS, 5.0)[pos]
    width = (hi - lo)[pos]
    # the kinetic parameters are individually weakly identified (the reference's own saved N=1000 run shows the
    # same); the noise level is well identified and none of the marginals is wider than the prior
    assert np.all(post.std(axis=0) < 1.05 * width / np.sqrt(12))
    assert abs(post[:, 4].mean() - 5.0) < 1.5 and post[:, 4].std() < 1.0   # sigma recovered
    assert np.all(post >= lo[pos]) and np.all(post <= hi[pos])             # support mask respected
    assert abs(post[:, 1].mean() - truth[1]) < 0.35 * width[1]              # activation energy of the forward step
    # the likelihood at the posterior mean is close to the likelihood at the truth
    assert out["lk"].max() > -0.5 * 150 - 150 * np.log(5.0) - 60
    assert np.isfinite(out["logZ"])


def test_config4_full_size_sweeps_complete(pkg, M, cond_guess):
    """BASELINE.json configs[3] at its stated size: 1e5 particles x 30 experiments = 3e6 DAE solves per sweep on one GPU
    (sim_particle, methanation_functions.py:70-92, then one Metropolis iteration, SMC_methanation_main.py:295-391).
    K8 is parity-unpinned (no IDA here), so the assertions are the size-independent properties of a sweep:
      * every (particle, experiment) item the sweep asks for is solved exactly once: the device counts finished solves
        and the host compares them with the number asked for (smc_meth_sweep_check; the library fails the sweep itself
        when they differ); no status is left at the pre-sweep poison value, no wave was split at a dequeue;
      * a failed solve carries the reference's -10000 sentinel in all five flows (methanation_set_likelihood.py:244-249),
        a solved one finite positive flows; the failed share over the prior box stays at the ~1.3 % measured in round 1;
      * in the Metropolis iteration the proposals outside the prior box are not solved (work list = 30 x live proposals)
        and keep p_filt / lk1 (:336, :384-386), the others follow the accept rule recomputed from the device's own lk2.
    """
    cond, guess = cond_guess
    n = 100_000
    rs = np.random.RandomState(11)
    flows0, _, _ = M.my_model(M.BASEPARAMS, cond, guess)
    obs = flows0 + 5.0 * rs.standard_normal(flows0.shape)
    eng, s = _meth_engine(pkg, M, cond, guess, obs, n)
    lo, hi, pos = M.prior_box()
    with eng:
        eng.sample_prior_device(4242, 0)
        theta = eng.download_particles(pkg.SMC_SET_PRED)
        assert np.all(theta >= lo[pos]) and np.all(theta <= hi[pos])
        info = eng.loglik(pkg.SMC_SET_PRED)                                  # 3e6 solves
        chk = eng.meth_sweep_check()
        assert chk == {"expected_solves": n * 30, "completed_solves": n * 30, "unsolved_items": 0, "wave_split": 0,
                       "cancelled_solves": 0}, chk
        k8 = eng.meth_sweep_counters()
        flows, status = eng.meth_download_solves()
        assert set(np.unique(status).tolist()) <= {0, 1}
        failed = status != 0
        assert int(failed.sum()) == k8["failed_solves"]
        assert 0.002 < failed.mean() < 0.04, failed.mean()                    # 1.3 % in round 1's N = 300 sample
        assert np.all(flows[failed] == -10000.0)
        ok_flows = flows[~failed]
        assert np.all(np.isfinite(ok_flows))
        # The model itself allows negative concentrations (the reverse rate of func_rCH4 is not clipped, so in the corners of
        # the prior box the reverse reaction "consumes" CH4 and H2O that are not there): a solved item need not be physical,
        # and IDA would integrate the same equations.  What every solved item must obey are the conservation laws of the
        # reactor at the outlet: the inert (Ar, component 4) passes through, carbon (CO2 + CH4) and the 2:1 H2O:CH4
        # stoichiometry are preserved.  Flows: [H2, CO2, CH4, H2O, Ar].
        okm = ~failed                                                          # (n, 30)
        ar_in = np.median(flows[:, :, 4], axis=0)                              # per experiment
        c_in = np.median(flows[:, :, 1] + flows[:, :, 2], axis=0)
        ar_err = np.abs(flows[:, :, 4] - ar_in[None]) / ar_in[None]
        c_err = np.abs(flows[:, :, 1] + flows[:, :, 2] - c_in[None]) / c_in[None]
        st_err = np.abs(flows[:, :, 3] - 2 * flows[:, :, 2]) / np.maximum(1.0, np.abs(flows[:, :, 3]))
        odd = ((ok_flows < -1.0) | (ok_flows > 1e4)).any(axis=1).mean()
        print(f"config 4 initial sweep: failed share {failed.mean():.4f}, solved-but-unphysical share {odd:.5f}; "
              f"Ar balance err q50/q99/max {np.quantile(ar_err[okm], [0.5, 0.99]).tolist()} {ar_err[okm].max():.3g}; "
              f"carbon balance err q50/q99/max {np.quantile(c_err[okm], [0.5, 0.99]).tolist()} {c_err[okm].max():.3g}; "
              f"H2O = 2 CH4 err q99/max {np.quantile(st_err[okm], 0.99):.3g} {st_err[okm].max():.3g}")
        assert np.quantile(ar_err[okm], 0.99) < 0.05 and np.quantile(c_err[okm], 0.99) < 0.05
        lk = eng.download_lk(pkg.SMC_SET_PRED)
        assert np.all(np.isfinite(lk))
        # my_loglike recomputed on the host from the downloaded flows (methanation_set_likelihood.py:280-300)
        sig = theta[:, 4]
        sq = ((flows.transpose(0, 2, 1) - obs[None]) ** 2).sum(axis=2)        # (n, 5)
        lk_ref = (-(0.5 / sig ** 2)[:, None] * sq - 30 * np.log(sig)[:, None]).sum(axis=1)
        assert np.max(np.abs(lk - lk_ref) / np.maximum(1.0, np.abs(lk_ref))) < 1e-12
        assert info["rk_attempts"] == k8["bdf_steps"] > 100 * n * 30 * 0.9

        # one Metropolis iteration; every third proposal is pushed out of the prior box
        eng.upload_particles(pkg.SMC_SET_FILT, theta)
        eng.upload_lk(pkg.SMC_SET_FILT, lk)
        eng.set_debug_capture(True)
        noise = rs.standard_normal((n, 5)) * (hi[pos] - lo[pos]) * 0.01
        noise[::3, 1] = 1e12
        rr = rs.uniform(0, 1, n)
        gamma = 0.01
        out = eng.mh_step_host_rng(gamma, 1.0, noise, rr)
        prop, lk2, p0, r = eng.download_debug_proposals()
        p_try = theta + noise * 1.0
        p0_ref = ((p_try >= lo[pos]) & (p_try <= hi[pos])).all(axis=1)
        assert np.array_equal(p0.astype(bool), p0_ref) and not p0_ref[::3].any()
        n_live = int(p0_ref.sum())
        chk = eng.meth_sweep_check()
        assert chk == {"expected_solves": n_live * 30, "completed_solves": n_live * 30, "unsolved_items": 0,
                       "wave_split": 0, "cancelled_solves": 0}, chk      # debug capture switches early rejection off
        _, status2 = eng.meth_download_solves()
        assert np.all(status2[~p0_ref] == -1) and np.all(status2[p0_ref] >= 0)   # masked proposals were not solved
        f = eng.download_particles(pkg.SMC_SET_FILT)
        lk1 = eng.download_lk(pkg.SMC_SET_FILT)
        assert np.array_equal(f[~p0_ref], theta[~p0_ref]) and np.array_equal(lk1[~p0_ref], lk[~p0_ref])
        with np.errstate(over="ignore"):
            r_ref = (np.exp((lk2 - lk) * gamma) * p0_ref >= rr)
        assert np.array_equal(r.astype(bool), r_ref) and out["accepted_now"] == int(r_ref.sum())
        assert np.array_equal(f[r_ref], prop[r_ref]) and np.array_equal(f[~r_ref], theta[~r_ref])
        assert np.array_equal(lk1[r_ref], lk2[r_ref]) and np.array_equal(lk1[~r_ref], lk[~r_ref])
        flags_off = eng.download_accept_flags()

        # the SAME Metropolis iteration with exact early rejection (VERDICT r2 item 3, config 4's size): nothing observable
        # may change, every item is either solved or cancelled, and a good share of the 2e6 solves is not started
        eng.set_debug_capture(False)
        eng.set_early_reject(True)
        eng.upload_particles(pkg.SMC_SET_FILT, theta)
        eng.upload_lk(pkg.SMC_SET_FILT, lk)
        eng.reset_accept_flags()
        out_on = eng.mh_step_host_rng(gamma, 1.0, noise, rr)
        chk_on = eng.meth_sweep_check()
        assert chk_on["expected_solves"] == n_live * 30 == chk_on["completed_solves"] + chk_on["cancelled_solves"]
        assert chk_on["unsolved_items"] == 0 and chk_on["wave_split"] == 0
        assert chk_on["cancelled_solves"] > 0.01 * n_live * 30, chk_on      # small steps (1 % of the box), gamma = 0.01: 3 %
        print(f"config 4 Metropolis sweep with early rejection: {chk_on['cancelled_solves']} of {n_live * 30} solves not started "
              f"({100.0 * chk_on['cancelled_solves'] / (n_live * 30):.1f} %)")
        assert out_on["accepted_now"] == out["accepted_now"] and out_on["accepted_ever"] == out["accepted_ever"]
        assert np.array_equal(eng.download_particles(pkg.SMC_SET_FILT), f)
        assert np.array_equal(eng.download_lk(pkg.SMC_SET_FILT), lk1)
        assert np.array_equal(eng.download_accept_flags(), flags_off)
        _, status3 = eng.meth_download_solves()
        cancelled_particles = (status3 == -2).any(axis=1)
        assert not r_ref[cancelled_particles].any()                             # only rejected proposals lose solves
        assert np.all(status3[~p0_ref] == -1) and np.all(status3[p0_ref] != -1)


def test_methanation_early_rejection_changes_nothing_but_the_solve_count(pkg, M, cond_guess):
    """Exact early rejection at N = 192: one host-RNG Metropolis iteration on a prior-drawn population
    at three tempering exponents, and a complete device-RNG run, each with the feature on and off: selected particles,
    likelihoods, accept flags and counts, the tempering schedule and the evidence are bit-identical; the number of DAE
    solves is smaller and completed + cancelled == asked for in every sweep (the library checks that itself and fails the
    sweep otherwise: smc_meth_sweep_check)."""
    cond, guess = cond_guess
    np.random.seed(20250205)
    flows0, _, _ = M.my_model(M.BASEPARAMS, cond, guess)
    obs = flows0 + 5.0 * np.random.standard_normal(flows0.shape)
    n = 192
    lo, hi, pos = M.prior_box()
    rs = np.random.RandomState(8)
    theta = lo[pos] + (hi[pos] - lo[pos]) * rs.uniform(0, 1, (n, 5))
    noise = rs.standard_normal((n, 5)) * (hi[pos] - lo[pos]) * 0.05
    rr = rs.uniform(0, 1, n)
    eng, s = _meth_engine(pkg, M, cond, guess, obs, n)
    with eng:
        eng.upload_particles(pkg.SMC_SET_PRED, theta)
        eng.loglik(pkg.SMC_SET_PRED)
        lk = eng.download_lk(pkg.SMC_SET_PRED)
        for gamma in (1e-4, 1e-2, 1.0):
            res = {}
            for on in (False, True):
                eng.set_early_reject(on)
                eng.upload_particles(pkg.SMC_SET_FILT, theta)
                eng.upload_lk(pkg.SMC_SET_FILT, lk)
                eng.reset_accept_flags()
                out = eng.mh_step_host_rng(gamma, 1.0, noise, rr)
                chk = eng.meth_sweep_check()
                assert chk["completed_solves"] + chk["cancelled_solves"] == chk["expected_solves"] and chk["unsolved_items"] == 0
                res[on] = (out["accepted_now"], out["accepted_ever"], eng.download_particles(pkg.SMC_SET_FILT),
                           eng.download_lk(pkg.SMC_SET_FILT), eng.download_accept_flags(), chk)
            a, b = res[True], res[False]
            assert a[0] == b[0] and a[1] == b[1]
            assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
            assert b[5]["cancelled_solves"] == 0 and a[5]["expected_solves"] == b[5]["expected_solves"]
            print(f"gamma {gamma}: {a[5]['cancelled_solves']} of {a[5]['expected_solves']} solves cancelled, {a[0]} accepted")
            if gamma >= 1e-2:
                assert a[5]["cancelled_solves"] > 0
    runs = {}
    for on in (True, False):
        eng, s = _meth_engine(pkg, M, cond, guess, obs, n)
        s.early_reject = on
        with eng:
            runs[on] = pkg.run_smc(eng, s, rng="device", verbose=False, seed_device=5)
    a, b = runs[True], runs[False]
    assert [r_["gamma_new"] for r_ in a["records"]] == [r_["gamma_new"] for r_ in b["records"]]
    assert [r_["n_accept"] for r_ in a["records"]] == [r_["n_accept"] for r_ in b["records"]]
    assert np.array_equal(a["p_pred"], b["p_pred"]) and np.array_equal(a["lk"], b["lk"]) and a["logZ"] == b["logZ"]
    assert b["stats"]["dae_solves_cancelled"] == 0 < a["stats"]["dae_solves_cancelled"]
    assert a["stats"]["dae_solves"] + a["stats"]["dae_solves_cancelled"] == b["stats"]["dae_solves"]
    print(f"full run, N = 192: {a['stats']['dae_solves']} solves with early rejection, {b['stats']['dae_solves']} without")


# ---------------------------------------------------------------------------------------------------
# config 5: the methanation model SHARDED over ranks (SMC_methanation_main.py:295-391 per rank + the exchange of
# 6-double rows after resampling), rehearsed with W engine contexts on one GPU (loopback exchange instead of RCCL,
# which refuses two ranks on one device; tests/_thread_comm.py)
# ---------------------------------------------------------------------------------------------------
def _run_meth_ranks(pkg, M, cond, guess, obs, n, world, seed):
    import threading
    from _thread_comm import ThreadWorld
    tw = ThreadWorld(world)
    nl = n // world
    s, pos = _meth_settings(pkg, M, n)
    s.seed = seed
    engines = [pkg.HipEngine(nl, 5, device=0, n_global=n) for _ in range(world)]
    for r, e in enumerate(engines):
        e.set_model_methanation(cond, guess, obs, np.append(M.BASEPARAMS, M.SIGMA_TRUE), pos)
        e.set_prior(s.priors)
        if world > 1:
            e.debug_set_local_peers(engines, r, tw.barrier.wait)
    outs, errs = [None] * world, []
    lock = threading.Lock()

    def work(r):
        try:
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


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_methanation_run_equals_single_rank(pkg, M, cond_guess, world):
    """Config 5 in miniature (missing 2): d = 5, n = 96 particles in total, device RNG, early rejection on, W = 2
    and 3 ranks.  What only several ranks exercise: the d = 5 moments reduction (5 + 15 values), the exchange of 6-double
    rows (theta + lk) after resampling, the GLOBAL particle index in the Philox keys of generic_propose_kernel /
    generic_accept_kernel / meth_certainly_rejected (global_offset = rank * n_local), and the per-rank misfit order of the
    experiments (every rank orders by its own block's statistics - any order must give the same results).
    Device-RNG mode is keyed by the global index, so the sharded run must reproduce the one-rank run: tempering schedule,
    Metropolis lengths, accept counts and offspring counts exactly; particles to 1e-9, likelihoods to 1e-5, evidence to 1e-6 (the
    cross-rank moment sums round differently from one block's tree, so proposals may differ in the last bits)."""
    cond, guess = cond_guess
    np.random.seed(20250205)
    flows0, _, _ = M.my_model(M.BASEPARAMS, cond, guess)
    obs = flows0 + 5.0 * np.random.standard_normal(flows0.shape)
    n, seed = 96, 19
    ref = _run_meth_ranks(pkg, M, cond, guess, obs, n, 1, seed)[0]
    outs = _run_meth_ranks(pkg, M, cond, guess, obs, n, world, seed)
    assert ref["gamma"] == 1.0 and ref["stats"]["dae_solves_cancelled"] > 0          # early rejection was at work
    for o in outs:
        assert o["gamma"] == 1.0
        assert [r["gamma_new"] for r in o["records"]] == [r["gamma_new"] for r in ref["records"]]
        assert [r["last_j"] for r in o["records"]] == [r["last_j"] for r in ref["records"]]
        assert [r["n_accept"] for r in o["records"]] == [r["n_accept"] for r in ref["records"]]
        assert [r["n_offspring"] for r in o["records"]] == [r["n_offspring"] for r in ref["records"]]
        assert abs(o["logZ"] - ref["logZ"]) <= 1e-6 * abs(ref["logZ"])          # a sum over likelihoods that agree to ~1e-6 (below)
    p = np.concatenate([o["p_pred"] for o in outs])
    lk = np.concatenate([o["lk"] for o in outs])
    assert p.shape == ref["p_pred"].shape
    assert (np.abs(p - ref["p_pred"]) / np.maximum(1.0, np.abs(ref["p_pred"]))).max() < 1e-9
    # likelihoods: 1e-5 relative = the integrator's own tolerance seen through the likelihood.  A proposal that differs in its last
    # bit can change one accept / reject decision of the BDF step controller or one Newton iteration count, which moves the outlet
    # flows by a fraction of the integrator's tolerance (rtol = atol = 1e-6 on flows of O(100) with sigma ~ 5: up to ~3e-3 on logL
    # ~ -320).  Observed: 1.1e-7 relative with round 4's control policy (Newton converged to 1e-3 tolerance units), 1.0e-6 with
    # round 5's (IDA's 0.33); none of these differences flipped a Metropolis decision (the counts above are exact).  K8's results
    # are defined up to the integrator's tolerance, and the number of ranks enters through the rounding of the moment sums only
    # (DESIGN.md 5).
    assert (np.abs(lk - ref["lk"]) / np.maximum(1.0, np.abs(ref["lk"]))).max() < 1e-5
    # every rank solved (or cancelled) exactly its own block's items in every sweep: the library's own check passed in each
    # sweep (smc_meth_sweep_check), and the blocks' solve counts add up to the one-rank run's
    tot = sum(o["stats"]["dae_solves"] + o["stats"]["dae_solves_cancelled"] for o in outs)
    assert tot == ref["stats"]["dae_solves"] + ref["stats"]["dae_solves_cancelled"]
