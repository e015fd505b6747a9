"""The shadow modules with the reference's names (drop-in boundary)."""
import importlib
import os
import sys

import numpy as np
import pytest

import __graft_entry__ as g

# every global the reference's settings module defines (Micmem_settings.py:15-127)
SETTINGS_NAMES = """n_cores n_particle inv_Np ess_limit mhstep_factor mhstep_factor_cov ad_mhstep_num mhstep_num
mhstep_ratio r_threshold r_threshold_f r_threshold_min d_gamma_max gm_reduction_itr gm_reduction_rate coefficent
coefficent_uni sigma_true num_est_params num_model_params est_params_list est_sigma priors sample_prior samples p_pred
itr_max n_hist fig_dimen w_cov dataset n_ex base_path datapoint obs_data p_filt p_weight p_is y_cal d_lk lk1""".split()


@pytest.fixture()
def dropin_cwd(tmp_path, monkeypatch):
    """A working directory laid out like SMC_example/: data/mm_pseudo_data_{i}.csv."""
    import pandas as pd
    z = np.load(os.path.join(g.ROOT, "tests", "golden", "mm_data.npz"))
    (tmp_path / "data").mkdir()
    for i in range(6):
        s_true = np.full(40, np.nan)
        s_true[0] = z["S0"][i]
        pd.DataFrame({"t": z["t"][i], "S_true": s_true, "P_true": np.nan, "P_obs": z["P_obs"][i]}).to_csv(
            tmp_path / "data" / f"mm_pseudo_data_{i}.csv", index=False)
    monkeypatch.chdir(tmp_path)
    g.load_package()
    monkeypatch.syspath_prepend(os.path.join(g.PKG_DIR, "dropin"))
    for m in ("Micmem_settings", "Micmem_likelihood"):
        sys.modules.pop(m, None)
    yield tmp_path
    for m in ("Micmem_settings", "Micmem_likelihood"):
        sys.modules.pop(m, None)


def test_settings_names_defaults_and_prior_draw(dropin_cwd, golden_run):
    S = importlib.import_module("Micmem_settings")
    missing = [n for n in SETTINGS_NAMES if not hasattr(S, n)]
    assert not missing, missing
    assert S.n_particle == 1000 and S.ess_limit == 0.5 and S.gm_reduction_itr == 80 and S.gm_reduction_rate == 0.7
    assert S.ad_mhstep_num == 20 and S.mhstep_num == 5 and S.itr_max == 50 and S.d_gamma_max == 1
    assert (S.r_threshold, S.r_threshold_f, S.r_threshold_min) == (0.5, 0.7, 0.1)
    assert np.array_equal(S.w_cov, np.full((3, 3), 0.5))
    # import-time side effects in the reference's order: identical prior sample on the reference's seed
    assert np.array_equal(S.p_pred, golden_run["sweeps_theta"][0])
    assert S.datapoint == 40 and S.n_ex == 6 and len(S.dataset) == 6
    assert S.p_is.dtype.kind == "i" and S.p_weight.sum() == pytest.approx(1.0)
    assert np.random.rand() != 0  # the global stream continues from where the reference's would


@pytest.mark.gpu
def test_sim_particle_matches_reference_first_sweep(dropin_cwd, golden_run, known_answers):
    L = importlib.import_module("Micmem_likelihood")
    llk, C_l_ = L.sim_particle(L.p_pred)
    ref = golden_run["sweeps_llk"][0]
    assert len(llk) == 1000
    assert np.max(np.abs(np.asarray(llk) - ref) / np.maximum(1, np.abs(ref))) < 1e-9
    # numpy coercions the reference driver relies on (Micmem_SMC_main.py:118,231,240)
    assert (llk - np.max(llk)).shape == (1000,)
    # lazily computed predictions, indexable per particle and experiment
    assert len(C_l_) == 1000 and len(C_l_[3]) == 6 and len(C_l_[3][0]) == 40
    # single-particle surface
    th = known_answers["theta"][1]
    out = L.log_likelihood_mm_multi(th)
    assert abs(out[0] - known_answers["logL"][1]) < 1e-9 * abs(known_answers["logL"][1])
    assert np.abs(np.array(out[1]) - known_answers["pred"][1]).max() < 1e-9
    assert L.log_likelihood_mm_multi.remote(th)[0] == out[0]
    assert L.log_likelihood_mm_multi(np.array([1.0, 1.0, 0.0])) == -np.inf
    P = L.simulate_mm_on_grid(th[0], th[1], L.dataset[0]["S0"], L.dataset[0]["t"])
    assert np.abs(P - known_answers["pred"][1][0]).max() < 1e-9
    assert L.mm_ode(0.0, 2.0, 1.0, 0.5) == -0.8


# ---------------------------------------------------------------------------------------------------
# methanation shadow modules
# ---------------------------------------------------------------------------------------------------
@pytest.fixture()
def meth_cwd(tmp_path, monkeypatch):
    """A working directory laid out like SMC_methanation/: methanation_data/information.csv (synthetic)."""
    import shutil
    (tmp_path / "methanation_data").mkdir()
    shutil.copy(os.path.join(g.ROOT, "tests", "golden", "methanation_information.csv"),
                tmp_path / "methanation_data" / "information.csv")
    monkeypatch.chdir(tmp_path)
    g.load_package()
    monkeypatch.syspath_prepend(os.path.join(g.PKG_DIR, "dropin"))
    mods = ("methanation_set_conditon", "methanation_set_likelihood", "methanation_functions")
    for m in mods:
        sys.modules.pop(m, None)
    yield tmp_path
    for m in mods:
        sys.modules.pop(m, None)


def test_methanation_settings_against_reference_values(meth_cwd):
    C = importlib.import_module("methanation_set_conditon")
    gold = np.load(os.path.join(g.ROOT, "tests", "golden", "methanation_golden.npz"))
    for k in ["Ca_in", "Cb_in", "Cc_in", "Cd_in", "Ce_in", "void", "T_in", "T_jacket", "u_in", "reactorlength",
              "low_limit", "high_limit", "w_cov", "baseparams"]:
        assert np.array_equal(np.asarray(getattr(C, k), dtype=np.float64), gold[k]), k
    assert C.est_position == list(gold["est_position"]) and C.li == list(gold["algvar"])
    assert C.n_data == 30 and C.NX == 51 and C.num_est_params == 5 and C.n_particle == 1000
    assert (C.ess_limit, C.gm_reduction_itr, C.gm_reduction_rate, C.itr_max) == (0.5, 80, 0.7, 50)
    # the global stream continues exactly where the reference's would: the synthetic-data noise of
    # SMC_methanation_main.py:94-95 is the first 5 x 30 standard normals after seed(20250205)
    z = np.random.standard_normal(30)
    np.random.seed(20250205)
    assert np.array_equal(z, np.random.standard_normal(30))


@pytest.mark.gpu
def test_methanation_sim_particle(meth_cwd):
    """sim_particle for 3 particles against the oracle's my_model + my_loglike (K8 is parity-unpinned vs IDA:
    agreement is asserted with the oracle's implementation of the same integrator, within tolerance units)."""
    F = importlib.import_module("methanation_functions")
    g.load_oracle()
    from oracle import methanation as M
    cond = M.load_conditions("methanation_data/information.csv")
    guess = M.initial_guess(cond)
    rs = np.random.RandomState(2)
    theta = F.low_limit_array + (F.high_limit_array - F.low_limit_array) * rs.uniform(0.3, 0.7, (3, 5))
    theta[0] = [F.baseparams[0], F.baseparams[1], F.baseparams[2], F.baseparams[3], 5.0]
    bases = np.tile(np.append(F.baseparams, F.sigma_true), (3, 1))
    flows0, _, _ = M.my_model(F.baseparams, cond, guess)
    obs = flows0 + 5.0 * rs.standard_normal(flows0.shape)
    llk, C_l_ = F.sim_particle(theta, guess, obs, bases)
    assert len(llk) == 3 and len(C_l_) == 3 and C_l_[0].shape == (5, 30)
    for k in range(3):
        fo, _, _ = M.my_model(bases[k, :8], cond, guess)
        ref = M.loglike(fo, obs, bases[k, 8], 30)
        assert abs(llk[k] - ref) < 1e-3 * max(1.0, abs(ref)), (k, llk[k], ref)
    assert np.allclose(C_l_[0].sum(axis=0), 1.0)
    pri = F.cal_prior(theta)
    assert np.all(pri > 0) and F.cal_prior(theta + 1e9)[0] == 0
