"""CPU tests: the methanation oracle (oracle/methanation_oracle.c, oracle/methanation.py) against golden
vectors produced by the reference's own functions (tests/golden/make_methanation_golden.py).
Pinned: residual, rate law, density, likelihood, prior, inlet conversions.  NOT pinned: time integration."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def M():
    import __graft_entry__ as g
    g.load_oracle()
    from oracle import methanation
    return methanation


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "methanation_golden.npz"))


def test_rate_law_and_density(M, gold):
    got = np.array([M.rCH4(*gold["rc_in"][j], gold["rc_par"][j]) for j in range(len(gold["rc_out"]))])
    assert np.allclose(got, gold["rc_out"], rtol=1e-14, atol=0)
    got = np.array([M.rohg(*gold["rg_in"][j]) for j in range(len(gold["rg_out"]))])
    assert np.array_equal(got, gold["rg_out"])


def test_residual(M, gold):
    res = M.reaction(gold["res_X"], gold["res_dX"], gold["res_p"])
    ref = gold["res_out"]
    scale = np.maximum(1.0, np.abs(ref))
    assert (np.abs(res - ref) / scale).max() < 1e-12
    # the structural rows are exact
    assert np.array_equal(res[:, 0], ref[:, 0]) and np.array_equal(res[:, 306], ref[:, 306])
    assert np.array_equal(res[:, [50, 101, 152, 203, 254, 305, 356]], ref[:, [50, 101, 152, 203, 254, 305, 356]])


def test_loglike(M, gold):
    got = np.array([M.loglike(gold["ll_y"][j], gold["ll_d"][j], gold["ll_s"][j], int(gold["n_data"])) for j in range(8)])
    assert np.allclose(got, gold["ll_out"], rtol=1e-14, atol=0)
    assert M.loglike(np.arange(150.).reshape(5, 30), np.arange(150.).reshape(5, 30) + 1, 5.0, 30) == -244.41568686511505


def test_settings_layer(M, gold):
    cond = M.load_conditions(os.path.join(GOLD, "methanation_information.csv"))
    for k in ["Ca_in", "Cb_in", "Cc_in", "Cd_in", "Ce_in", "void"]:
        assert np.array_equal(cond[k], gold[k]), k
    for k in ["T_in", "T_jacket", "u_in", "reactorlength"]:
        assert np.array_equal(cond[k], gold[k]), k
    assert np.array_equal(M.initial_guess(cond), gold["guess"])
    lo, hi, pos = M.prior_box()
    assert np.array_equal(lo, gold["low_limit"]) and np.array_equal(hi, gold["high_limit"])
    assert pos == list(gold["est_position"])


def test_prior(M, gold):
    got = M.cal_prior(gold["prior_theta"])
    assert np.array_equal(got, gold["prior_pdf"])
    assert 0 < (got > 0).sum() < len(got)


def test_product_settings_functions_equal_oracle(pkg, M):
    """methanation.load_conditions / initial_guess / prior_box / p0_rows (the product's functional form of
    methanation_set_conditon.py, used by bench.py and the tools) against the oracle's restatement, itself pinned by
    the reference's own conversions (tests/golden/methanation_golden.npz)."""
    import os
    csv = os.path.join(os.path.dirname(__file__), "golden", "methanation_information.csv")
    a, b = pkg.methanation.load_conditions(csv), M.load_conditions(csv)
    for k in b:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
    assert np.array_equal(pkg.methanation.initial_guess(a), M.initial_guess(b))
    (la, ha, pa), (lb, hb, pb) = pkg.methanation.prior_box(), M.prior_box()
    assert np.array_equal(la, lb) and np.array_equal(ha, hb) and pa == pb
    assert np.array_equal(pkg.methanation.p0_rows(a, M.BASEPARAMS), np.array([M.p0_tuple(b, i, M.BASEPARAMS) for i in range(30)]))
