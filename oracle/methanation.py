"""oracle/methanation.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the methanation model's pure-arithmetic layer: ctypes access to
oracle/methanation_oracle.c plus the settings-side conversions of the reference
(SMC_methanation/methanation_set_conditon.py) restated with the same NumPy expressions.

Parity status: residual / rate law / density / likelihood / prior / inlet conversions are PINNED by
tests/golden/methanation_golden.npz (reference functions evaluated on a synthetic inlet table, see
tests/golden/make_methanation_golden.py).  The DAE time integration is parity-UNPINNED.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import oracle as _O

NX = 51
_dp = ctypes.POINTER(ctypes.c_double)
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        L = _O.lib()
        L.meth_rCH4.restype = ctypes.c_double
        L.meth_rCH4.argtypes = [ctypes.c_double] * 5 + [_dp]
        L.meth_rohg.restype = ctypes.c_double
        L.meth_rohg.argtypes = [ctypes.c_double] * 7
        L.meth_reaction.restype = None
        L.meth_reaction.argtypes = [_dp, _dp, _dp, _dp]
        L.meth_reaction_batch.restype = None
        L.meth_reaction_batch.argtypes = [_dp, _dp, _dp, ctypes.c_int64, _dp]
        L.meth_loglike.restype = ctypes.c_double
        L.meth_loglike.argtypes = [_dp, _dp, ctypes.c_double, ctypes.c_int]
        L.meth_flows.restype = None
        L.meth_flows.argtypes = [_dp, ctypes.c_double, ctypes.c_double, ctypes.c_double, _dp]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(_dp)


def rCH4(T, Ca, Cb, Cc, Cd, params):
    params = np.ascontiguousarray(params, dtype=np.float64)
    return lib().meth_rCH4(float(T), float(Ca), float(Cb), float(Cc), float(Cd), _p(params))


def rohg(a, b, c, d, e, T, P0):
    return lib().meth_rohg(*[float(x) for x in (a, b, c, d, e, T, P0)])


def reaction(X, dX, params):
    X = np.ascontiguousarray(X, dtype=np.float64)
    dX = np.ascontiguousarray(dX, dtype=np.float64)
    params = np.ascontiguousarray(params, dtype=np.float64)
    if X.ndim == 1:
        res = np.empty(7 * NX)
        lib().meth_reaction(_p(X), _p(dX), _p(params), _p(res))
        return res
    res = np.empty_like(X)
    lib().meth_reaction_batch(_p(X), _p(dX), _p(params), X.shape[0], _p(res))
    return res


def loglike(y, data, sigma, n_data):
    y = np.ascontiguousarray(y, dtype=np.float64)
    data = np.ascontiguousarray(data, dtype=np.float64)
    return lib().meth_loglike(_p(y), _p(data), float(sigma), int(n_data))


# ---- settings layer (methanation_set_conditon.py) ------------------------------------------------
DATALIST = [0, 2, 5, 6, 8, 9, 10, 11, 13, 14, 15, 16, 17, 19, 20, 21, 22, 25, 26, 27, 28, 31, 35, 38, 40, 45, 49, 52, 55, 58]
EST_PARAMS_LIST = [1, 1, 1, 1, 0, 0, 0, 0, 1]
BASEPARAMS = np.array([13.04, 52.2e3, 1.147e5, 96.7e3, 23.34, -6, 0.72, -2.51e3])
SIGMA_TRUE = 5
R = 8.3144589
S_AREA = np.pi * (0.01 / 2) ** 2
P_STP = 1.013 * 10 ** 5


def prior_box():
    """low/high limits (methanation_set_conditon.py:59-70)."""
    use_params = np.append(BASEPARAMS, SIGMA_TRUE)
    high_k = [25, 1, 30, 2, 1, -2, 1, -2, 2]
    low_k = [4, 1, 4, 1, 1, -2, 1, -2, 0.9]
    high_limit = use_params + use_params * np.array(high_k)
    low_limit = use_params - use_params * np.array(low_k)
    est_position = [i for i, x in enumerate(EST_PARAMS_LIST) if x == 1]
    return low_limit, high_limit, est_position


def load_conditions(information_csv: str):
    """Inlet conditions from the information table (methanation_set_conditon.py:137-214)."""
    import pandas as pd
    info_df = pd.read_csv(information_csv).fillna(0)
    datastart, datafin = DATALIST[0], DATALIST[-1]
    information = info_df.iloc[datastart:datafin + 1].values
    n_data = len(DATALIST)
    out = {}
    reactorlength = information[:, 4].copy()
    T_jacket = information[:, 5].copy()
    void_fraction = information[:, 6]
    T_in = information[:, 7].copy()
    P_total = information[:, 9]
    fa, fb, fc, fd, fe = (information[:, k] for k in (10, 11, 12, 14, 15))
    in_flow_total = information[:, 16]
    Ca = np.zeros(n_data); Cb = np.zeros(n_data); Cc = np.zeros(n_data); Cd = np.zeros(n_data); Ce = np.zeros(n_data)
    void = np.zeros(n_data)
    for i in range(n_data):
        T_in[i] = T_in[i] + 273
        tot = fa[i] + fb[i] + fc[i] + fd[i] + fe[i]
        Ca[i] = (P_total[i] * 1e6 + 101325) / R / T_in[i] * fa[i] / tot
        Cb[i] = (P_total[i] * 1e6 + 101325) / R / T_in[i] * fb[i] / tot
        Cc[i] = (P_total[i] * 1e6 + 101325) / R / T_in[i] * fc[i] / tot
        Cd[i] = (P_total[i] * 1e6 + 101325) / R / T_in[i] * fd[i] / tot
        Ce[i] = (P_total[i] * 1e6 + 101325) / R / T_in[i] * fe[i] / tot
        T_jacket[i] = T_jacket[i] + 273
        reactorlength[i] = reactorlength[i] / 1000
        void[i] = void_fraction[i]
    u_in = in_flow_total * 1.667e-8 / S_AREA * (101325 * T_in) / ((P_total * 1e6 + 101325) * 298)   # :214
    # NB: the reference slices rows datastart..datafin (59 rows) but n_data = len(datalist) = 30, so its loops
    # only convert the FIRST 30 rows; arrays keep 59 entries, entries >= 30 unconverted (:139,188-212)
    out.update(Ca_in=Ca, Cb_in=Cb, Cc_in=Cc, Cd_in=Cd, Ce_in=Ce, T_in=T_in, T_jacket=T_jacket, u_in=u_in, void=void,
               reactorlength=reactorlength, n_data=n_data)
    return out


def initial_guess(cond):
    """SMC_methanation_main.py:47-58."""
    n_data = cond["n_data"]
    guess = np.zeros([n_data, 7 * NX])
    for i in range(n_data):
        g0 = np.ones(7 * NX)
        g0[0:NX] = cond["Ca_in"][i]
        g0[NX:2 * NX] = cond["Cb_in"][i]
        g0[2 * NX:3 * NX] = cond["Cc_in"][i]
        g0[3 * NX:4 * NX] = cond["Cd_in"][i]
        g0[4 * NX:5 * NX] = cond["Ce_in"][i]
        g0[5 * NX:6 * NX] = cond["T_in"][i]
        g0[5 * NX + 1:6 * NX] = 400
        g0[6 * NX:7 * NX] = cond["u_in"][i]
        guess[i, :] = g0
    return guess


def cal_prior(theta):
    """methanation_functions.py:96-135, uniform branch (normal_pred = False)."""
    import scipy.stats
    low_limit, high_limit, est_position = prior_box()
    lo = np.array([low_limit[i] for i in est_position])
    hi = np.array([high_limit[i] for i in est_position])
    p = scipy.stats.uniform.pdf(theta, lo, hi - lo)
    return np.prod(p.T, axis=0)


# ---- DAE time integration (parity UNPINNED; see oracle/meth_dae_oracle.c) ----------------------------------
class DaeStats(ctypes.Structure):
    _fields_ = [("steps", ctypes.c_int64), ("rejects", ctypes.c_int64), ("nlu", ctypes.c_int64), ("nres", ctypes.c_int64),
                ("newton_fail", ctypes.c_int64), ("status", ctypes.c_int32), ("order_hist", ctypes.c_int32 * 6)]

    def asdict(self):
        return {"steps": self.steps, "rejects": self.rejects, "nlu": self.nlu, "nres": self.nres,
                "newton_fail": self.newton_fail, "status": self.status, "order_hist": list(self.order_hist)}


class DaePolicy(ctypes.Structure):
    """Control policy of the BDF integrator (oracle/meth_dae_oracle.c: dae_policy).  The default of the C side (NULL) is the
    checker of rounds 1-4; K8_POLICY is what the HIP kernel K8 does since round 5 (csrc/meth_dae_elem.h: SMC_K8_POLICY 1)."""
    _fields_ = [("reuse", ctypes.c_int32), ("newton", ctypes.c_int32), ("stepctl", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("epcon", ctypes.c_double), ("xrate", ctypes.c_double)]


class DaeStatsExt(ctypes.Structure):
    _fields_ = [("newton_iters", ctypes.c_int64), ("stale_retries", ctypes.c_int64)]


def k8_policy():
    return DaePolicy(2, 1, 0, 0, 0.33, 0.15)


def dae_solve_policy(y0, p, policy=None, tf=75.0, rtol=1e-6, atol=1e-6, h0=1e-5):
    """dae_solve under a control policy (None: the checker's default); the stats carry Newton iterations and stale retries."""
    y0 = np.ascontiguousarray(y0, dtype=np.float64)
    p = np.ascontiguousarray(p, dtype=np.float64)
    out = np.empty(7 * NX)
    st, ext = DaeStats(), DaeStatsExt()
    rc = _dae_lib().meth_dae_solve_policy(_p(y0), _p(p), tf, rtol, atol, h0, _p(out), ctypes.byref(st),
                                          ctypes.byref(policy) if policy is not None else None, ctypes.byref(ext))
    d = st.asdict()
    d.update(newton_iters=ext.newton_iters, stale_retries=ext.stale_retries)
    return out, rc, d


class IdaStats(ctypes.Structure):
    _fields_ = [("steps", ctypes.c_int64), ("netf", ctypes.c_int64), ("ncfn", ctypes.c_int64), ("nsetups", ctypes.c_int64),
                ("nni", ctypes.c_int64), ("nres", ctypes.c_int64), ("status", ctypes.c_int32), ("order_hist", ctypes.c_int32 * 6)]

    def asdict(self):
        return {"steps": self.steps, "error_test_failures": self.netf, "newton_failures": self.ncfn, "nlu": self.nsetups,
                "newton_iters": self.nni, "nres": self.nres, "status": self.status, "order_hist": list(self.order_hist)}


def dae_solve_ida(y0, p, tf=75.0, ncp=10, rtol=1e-6, atol=1e-6):
    """The same DAE through the restatement of IDA's OWN algorithm (oracle/meth_dae_oracle.c: dae_ida_integrate - fixed-leading-
    coefficient variable-step BDF on divided differences, IDA's step / order / Newton control, interpolation at the output points):
    a cross-check of K8's formulation, itself never compared with SUNDIALS (absent): parity stays unpinned."""
    y0 = np.ascontiguousarray(y0, dtype=np.float64)
    p = np.ascontiguousarray(p, dtype=np.float64)
    out = np.empty(7 * NX)
    st = IdaStats()
    rc = _dae_lib().meth_dae_solve_ida(_p(y0), _p(p), tf, int(ncp), rtol, atol, _p(out), ctypes.byref(st))
    return out, rc, st.asdict()


def ida_test_ode(y0, k, tf, rtol, atol):
    y0 = np.ascontiguousarray(y0, dtype=np.float64)
    k = np.ascontiguousarray(k, dtype=np.float64)
    out = np.empty(3)
    st = IdaStats()
    rc = _dae_lib().dae_ida_test_ode(_p(y0), _p(k), tf, rtol, atol, _p(out), ctypes.byref(st))
    return out, rc, st.asdict()


def _dae_lib():
    L = lib()
    L.meth_dae_solve_ida.restype = ctypes.c_int
    L.meth_dae_solve_ida.argtypes = [_dp, _dp, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_double, _dp, ctypes.POINTER(IdaStats)]
    L.dae_ida_test_ode.restype = ctypes.c_int
    L.dae_ida_test_ode.argtypes = [_dp, _dp, ctypes.c_double, ctypes.c_double, ctypes.c_double, _dp, ctypes.POINTER(IdaStats)]
    L.meth_dae_solve_policy.restype = ctypes.c_int
    L.meth_dae_solve_policy.argtypes = [_dp, _dp, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _dp,
                                        ctypes.POINTER(DaeStats), ctypes.POINTER(DaePolicy), ctypes.POINTER(DaeStatsExt)]
    L.meth_dae_solve.restype = ctypes.c_int
    L.meth_dae_solve.argtypes = [_dp, _dp, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _dp,
                                 ctypes.POINTER(DaeStats)]
    L.meth_model_one.restype = None
    L.meth_model_one.argtypes = [_dp, _dp, ctypes.c_double, ctypes.c_double, _dp, _dp, ctypes.POINTER(DaeStats)]
    L.dae_bdf_test_ode.restype = ctypes.c_int
    L.dae_bdf_test_ode.argtypes = [_dp, _dp, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _dp,
                                   ctypes.POINTER(DaeStats)]
    return L


def dae_solve(y0, p, tf=75.0, rtol=1e-6, atol=1e-6, h0=1e-5):
    y0 = np.ascontiguousarray(y0, dtype=np.float64)
    p = np.ascontiguousarray(p, dtype=np.float64)
    out = np.empty(7 * NX)
    st = DaeStats()
    rc = _dae_lib().meth_dae_solve(_p(y0), _p(p), tf, rtol, atol, h0, _p(out), ctypes.byref(st))
    return out, rc, st.asdict()


def p0_tuple(cond, i, pr):
    """The p0 tuple of my_model (methanation_set_likelihood.py:164)."""
    return np.array([cond["Ca_in"][i], cond["Cb_in"][i], cond["Cc_in"][i], cond["Cd_in"][i], cond["Ce_in"][i],
                     cond["T_in"][i], cond["T_jacket"][i], cond["u_in"][i], cond["void"][i],
                     cond["reactorlength"][i] / (NX - 1), *np.asarray(pr, dtype=np.float64)[:8]], dtype=np.float64)


def my_model(params, cond, guess):
    """Flows (5, n_data) and outlet states for one parameter vector (methanation_set_likelihood.py:144-277)."""
    n_data = cond["n_data"]
    flows = np.empty((5, n_data))
    states = np.empty((n_data, 7 * NX))
    stats = []
    L = _dae_lib()
    for i in range(n_data):
        p = p0_tuple(cond, i, params)
        f = np.empty(5)
        st = DaeStats()
        L.meth_model_one(_p(np.ascontiguousarray(guess[i])), _p(p), S_AREA, P_STP, _p(f), _p(states[i]), ctypes.byref(st))
        flows[:, i] = f
        stats.append(st.asdict())
    return flows, states, stats


def bdf_test_ode(y0, k, tf, rtol, atol, h0=1e-5):
    y0 = np.ascontiguousarray(y0, dtype=np.float64)
    k = np.ascontiguousarray(k, dtype=np.float64)
    out = np.empty(3)
    st = DaeStats()
    rc = _dae_lib().dae_bdf_test_ode(_p(y0), _p(k), tf, rtol, atol, h0, _p(out), ctypes.byref(st))
    return out, rc, st.asdict()
