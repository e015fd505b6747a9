"""Host side of the methanation rows built so far (configs 4-5):
batched DAE residual, rate law and log-likelihood from outlet flows on the GPU, through the C ABI.
The time integration that would turn these into `my_model` (SMC_methanation/methanation_set_likelihood.py:
144-277, Assimulo IDA in the reference) is not built yet."""
from __future__ import annotations

import numpy as np

from . import binding as B
from .binding import SmcError, lib

NX, NSTATE, NPAR = 51, 357, 18


def _dp(a):
    return a.ctypes.data_as(B.c_dp)


def _ck(st, what):
    if st != 0:
        raise SmcError(f"{what}: {lib().smc_meth_last_error().decode()}")


def reaction(X, dX, params, device=0):
    """res = reaction(t, X, dX, params) for a batch (n, 357), (n, 357), (n, 18)."""
    X = np.ascontiguousarray(np.atleast_2d(X), dtype=np.float64)
    dX = np.ascontiguousarray(np.atleast_2d(dX), dtype=np.float64)
    params = np.ascontiguousarray(np.atleast_2d(params), dtype=np.float64)
    assert X.shape[1] == NSTATE and dX.shape == X.shape and params.shape == (X.shape[0], NPAR)
    res = np.empty_like(X)
    _ck(lib().smc_meth_residual_host(device, _dp(X), _dp(dX), _dp(params), X.shape[0], _dp(res)), "smc_meth_residual_host")
    return res


def func_rCH4(T, Ca, Cb, Cc, Cd, params, device=0):
    inp = np.ascontiguousarray(np.column_stack([np.atleast_1d(v) for v in (T, Ca, Cb, Cc, Cd)]), dtype=np.float64)
    kin = np.ascontiguousarray(np.atleast_2d(params), dtype=np.float64)
    assert kin.shape == (inp.shape[0], 8)
    out = np.empty(inp.shape[0])
    _ck(lib().smc_meth_rate_host(device, _dp(inp), _dp(kin), inp.shape[0], _dp(out)), "smc_meth_rate_host")
    return out


def my_loglike(y, data, sigma, n_data, device=0):
    """y: (n, 5, n_data) or (5, n_data); data (5, n_data); sigma scalar or (n,)."""
    y = np.ascontiguousarray(y, dtype=np.float64)
    single = y.ndim == 2
    if single:
        y = y[None]
    data = np.ascontiguousarray(data, dtype=np.float64)
    sigma = np.ascontiguousarray(np.broadcast_to(np.asarray(sigma, dtype=np.float64), (y.shape[0],)))
    assert y.shape[1:] == (5, n_data) and data.shape == (5, n_data)
    lk = np.empty(y.shape[0])
    _ck(lib().smc_meth_loglike_host(device, _dp(y), _dp(data), _dp(sigma), y.shape[0], int(n_data), _dp(lk)),
        "smc_meth_loglike_host")
    return lk[0] if single else lk


S_AREA = np.pi * (0.01 / 2) ** 2          # methanation_set_conditon.py:80-81
P_STP = 1.013 * 10 ** 5                    # :89


def dae_solve_batch(p0_all, y0_all, tf=75.0, rtol=1e-6, atol=1e-6, h0=1e-5, want_states=False, device=0):
    """Integrate n independent (particle, experiment) DAE solves on the GPU (K8, parity unpinned).
    p0_all (n, 18), y0_all (n, 357) -> flows (n, 5), status (n,), states (n, 357) | None, info dict."""
    p0_all = np.ascontiguousarray(np.atleast_2d(p0_all), dtype=np.float64)
    y0_all = np.ascontiguousarray(np.atleast_2d(y0_all), dtype=np.float64)
    n = p0_all.shape[0]
    assert p0_all.shape == (n, NPAR) and y0_all.shape == (n, NSTATE)
    flows = np.empty((n, 5))
    status = np.empty(n, dtype=np.int32)
    states = np.empty((n, NSTATE)) if want_states else None
    stats = np.zeros(4, dtype=np.int64)
    ms = B.ctypes.c_double(0)
    _ck(lib().smc_meth_dae_host(device, _dp(p0_all), _dp(y0_all), n, tf, rtol, atol, h0, S_AREA, P_STP, _dp(flows),
                                _dp(states) if want_states else None, status.ctypes.data_as(B.ctypes.POINTER(B.ctypes.c_int32)),
                                stats.ctypes.data_as(B.c_i64p), B.ctypes.byref(ms)), "smc_meth_dae_host")
    info = {"steps": int(stats[0]), "rejects": int(stats[1]), "newton_fail": int(stats[2]), "newton_iters": int(stats[3]),
            "kernel_ms": ms.value}
    return flows, status, states, info
