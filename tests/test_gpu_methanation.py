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
