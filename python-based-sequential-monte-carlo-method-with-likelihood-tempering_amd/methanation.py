"""Host side of the methanation rows (configs 4-5): batched DAE residual, rate law,
log-likelihood from outlet flows and the DAE time integration (`my_model`, K8) on the GPU through the C ABI, plus the
settings-layer conversions of methanation_set_conditon.py as functions of a file path (the drop-in module
dropin/methanation_set_conditon.py keeps the reference's import-time form)."""
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
    stats = np.zeros(5, dtype=np.int64)
    ms = B.ctypes.c_double(0)
    _ck(lib().smc_meth_dae_host(device, _dp(p0_all), _dp(y0_all), n, tf, rtol, atol, h0, S_AREA, P_STP, _dp(flows),
                                _dp(states) if want_states else None, status.ctypes.data_as(B.ctypes.POINTER(B.ctypes.c_int32)),
                                stats.ctypes.data_as(B.c_i64p), B.ctypes.byref(ms)), "smc_meth_dae_host")
    info = {"steps": int(stats[0]), "rejects": int(stats[1]), "newton_fail": int(stats[2]), "newton_iters": int(stats[3]),
            "factorisations": int(stats[4]), "kernel_ms": ms.value}
    return flows, status, states, info


# ---- settings layer as functions (methanation_set_conditon.py) ---------------------------------------------------
DATALIST = [0, 2, 5, 6, 8, 9, 10, 11, 13, 14, 15, 16, 17, 19, 20, 21, 22, 25, 26, 27, 28, 31, 35, 38, 40, 45, 49, 52, 55, 58]  # :48
EST_PARAMS_LIST = [1, 1, 1, 1, 0, 0, 0, 0, 1]                                                                                 # :14
BASEPARAMS = np.array([13.04, 52.2e3, 1.147e5, 96.7e3, 23.34, -6, 0.72, -2.51e3])                                             # :55
SIGMA_TRUE = 5                                                                                                                # :53
GAS_R = 8.3144589
TUBE_AREA = np.pi * (0.01 / 2) ** 2


# information.csv columns the settings module reads (methanation_set_conditon.py:141-186)
_INFO_COLS = {"catag": 2, "reactorlength": 4, "T_jacket": 5, "void_fraction": 6, "T_in": 7, "P_total": 9,
              "in_flow_a": 10, "in_flow_b": 11, "in_flow_c": 12, "in_flow_d": 14, "in_flow_e": 15, "in_flow_total": 16,
              "out_flow_a": 17, "out_flow_b": 18, "out_flow_c": 19, "out_flow_d": 21, "out_flow_e": 22, "out_flow_total": 23,
              "out_molf_a": 24, "out_molf_b": 25, "out_molf_c": 26, "out_molf_d": 28, "out_molf_e": 29}


def settings_arrays(information_csv: str) -> dict:
    """Every array the reference's settings module derives from the information table
    (methanation_set_conditon.py:137-214), under the reference's names - the ONE place where that conversion lives (the
    drop-in module dropin/methanation_set_conditon.py only republishes these under module-level names).
    Column views of the datastart..datafin slice (:139-186); then, for the first n_data rows only, as the reference's
    loop does (:188-212): inlet concentrations C = (P_gauge*1e6 + 101325) / R / T_in * flow / sum(flows), Kelvin
    temperatures, metres and kilograms; u_in over the whole slice (:214).  Element for element the reference's
    arithmetic (same operations in the same order, so the values are bit-equal to the loop's)."""
    import pandas as pd
    information = pd.read_csv(information_csv).fillna(0).iloc[DATALIST[0]:DATALIST[-1] + 1].values
    n_data = len(DATALIST)
    a = {"information": information, "n_data": n_data}
    for name, col in _INFO_COLS.items():
        a[name] = information[:, col]                       # views: the in-place unit conversions below show through
    h = slice(0, n_data)
    a["T_in"][h] = a["T_in"][h] + 273
    tot = a["in_flow_a"][h] + a["in_flow_b"][h] + a["in_flow_c"][h] + a["in_flow_d"][h] + a["in_flow_e"][h]
    for k in "abcde":
        a[f"C{k}_in"] = (a["P_total"][h] * 1e6 + 101325) / GAS_R / a["T_in"][h] * a[f"in_flow_{k}"][h] / tot
        a[f"X{k}_out"] = a[f"out_molf_{k}"][h].copy()
        a[f"F{k}_out"] = a[f"out_flow_{k}"][h].copy()
    a["T_jacket"][h] = a["T_jacket"][h] + 273
    a["catag"][h] = a["catag"][h] / 1000
    a["reactorlength"][h] = a["reactorlength"][h] / 1000
    a["sccm"] = a["in_flow_total"][h].copy()
    a["void"] = a["void_fraction"][h].copy()
    a["u_in"] = a["in_flow_total"] * 1.667e-8 / TUBE_AREA * (101325 * a["T_in"]) / ((a["P_total"] * 1e6 + 101325) * 298)
    return a


def load_conditions(information_csv: str) -> dict:
    """The inlet conditions the DAE needs, from settings_arrays()."""
    a = settings_arrays(information_csv)
    return dict(Ca_in=a["Ca_in"], Cb_in=a["Cb_in"], Cc_in=a["Cc_in"], Cd_in=a["Cd_in"], Ce_in=a["Ce_in"], T_in=a["T_in"],
                T_jacket=a["T_jacket"], u_in=a["u_in"], void=a["void"], reactorlength=a["reactorlength"], n_data=a["n_data"])


def initial_guess(cond: dict) -> np.ndarray:
    """The starting profile of every experiment (SMC_methanation_main.py:47-58): inlet values along the bed, 400 K
    behind the first node."""
    n_data = cond["n_data"]
    guess = np.ones((n_data, 7 * NX))
    for f, key in enumerate(("Ca_in", "Cb_in", "Cc_in", "Cd_in", "Ce_in", "T_in")):
        guess[:, f * NX:(f + 1) * NX] = np.asarray(cond[key])[:n_data, None]
    guess[:, 5 * NX + 1:6 * NX] = 400
    guess[:, 6 * NX:7 * NX] = np.asarray(cond["u_in"])[:n_data, None]
    return guess


HIGH_K = [25, 1, 30, 2, 1, -2, 1, -2, 2]          # :59
LOW_K = [4, 1, 4, 1, 1, -2, 1, -2, 0.9]           # :60


def prior_box():
    """(low_limit, high_limit, est_position) of the uniform priors (methanation_set_conditon.py:22,59-70)."""
    use = np.append(BASEPARAMS, SIGMA_TRUE)
    high = use + use * np.array(HIGH_K)
    low = use - use * np.array(LOW_K)
    return low, high, [i for i, x in enumerate(EST_PARAMS_LIST) if x == 1]


def p0_rows(cond: dict, params) -> np.ndarray:
    """The p0 tuples of my_model for all experiments (methanation_set_likelihood.py:164): (n_data, 18)."""
    n_data = cond["n_data"]
    cols = [np.asarray(cond[k], dtype=np.float64)[:n_data] for k in
            ("Ca_in", "Cb_in", "Cc_in", "Cd_in", "Ce_in", "T_in", "T_jacket", "u_in", "void")]
    cols.append(np.asarray(cond["reactorlength"], dtype=np.float64)[:n_data] / (NX - 1))
    pr = np.asarray(params, dtype=np.float64)[:8]
    return np.column_stack(cols + [np.full(n_data, v) for v in pr])
