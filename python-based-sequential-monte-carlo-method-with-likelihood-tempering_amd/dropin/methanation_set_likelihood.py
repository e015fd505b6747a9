"""Shadow of SMC_methanation/methanation_set_likelihood.py backed by the HIP engine.

  func_rCH4(T,Ca,Cb,Cc,Cd,params)     :44-58    GPU rate-law kernel
  func_rohg(a,b,c,d,e,T,P0)           :61-66    one expression (host arithmetic of the caller's scalars)
  reaction(t, X, dX, params)          :69-139   GPU residual kernel
  my_model(params, initial_guess)     :144-277  GPU DAE integration (K8) for the n_data experiments; returns
                                                (Flow, molfraction) like the reference.  PARITY UNPINNED against
                                                the reference's IDA (see csrc/meth_dae.h).
  my_loglike(y, data, sigma, n_data)  :280-300  GPU likelihood kernel
"""
import numpy as np

from methanation_set_conditon import *  # noqa: F401,F403
from smc_lt_amd import methanation as _gpu

errorbox = [['params', 'i']]


def func_rCH4(T, Ca, Cb, Cc, Cd, params):
    return float(_gpu.func_rCH4(T, Ca, Cb, Cc, Cd, np.asarray(params, dtype=np.float64)[None, :8])[0])


def func_rohg(a, b, c, d, e, T, P0):
    return P0 / R / T * (a * 2 + b * 44 + c * 16 + d * 18 + e * 40) / (a + b + c + d + e) * 0.001   # noqa: F405


def reaction(t, X, dX, params):
    return _gpu.reaction(np.asarray(X, dtype=np.float64), np.asarray(dX, dtype=np.float64),
                         np.asarray(params, dtype=np.float64))[0]


def _p0_rows(pr):
    """The p0 tuples of :164 for every experiment."""
    pr = np.asarray(pr, dtype=np.float64)
    rows = np.empty((n_data, 18))   # noqa: F405
    for i in range(n_data):   # noqa: F405
        rows[i] = (Ca_in[i], Cb_in[i], Cc_in[i], Cd_in[i], Ce_in[i], T_in[i], T_jacket[i], u_in[i], void[i],   # noqa: F405
                   reactorlength[i] / (NX - 1), pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], pr[6], pr[7])   # noqa: F405
    return rows


def my_model_batch(params_all, initial_guess):
    """my_model for many parameter vectors at once: (n, >=8) -> Flow (n, 5, n_data), molfraction (n, 5, n_data)."""
    params_all = np.atleast_2d(np.asarray(params_all, dtype=np.float64))
    n = params_all.shape[0]
    p0 = np.concatenate([_p0_rows(pr) for pr in params_all])
    y0 = np.tile(np.asarray(initial_guess, dtype=np.float64)[:n_data], (n, 1))   # noqa: F405
    flows, status, states, _ = _gpu.dae_solve_batch(p0, y0, want_states=True)
    Flow = flows.reshape(n, n_data, 5).transpose(0, 2, 1).copy()   # noqa: F405
    c = states.reshape(n, n_data, 7, NX)[:, :, :5, -1]   # outlet concentrations   # noqa: F405
    mol = (c / c.sum(axis=2, keepdims=True)).transpose(0, 2, 1).copy()   # :220-224
    failed = status.reshape(n, n_data) != 0   # noqa: F405
    mol[np.broadcast_to(failed[:, None, :], mol.shape)] = 0.0   # :250-254
    for k, i in zip(*np.nonzero(failed)):
        errorbox.append([params_all[k], int(i)])   # :239-240
    return Flow, mol


def my_model(params, initial_guess):
    Flow, mol = my_model_batch(np.asarray(params, dtype=np.float64)[None, :], initial_guess)
    return Flow[0], mol[0]


def my_loglike(y, data, sigma, n_data, scale=1.0):
    return float(_gpu.my_loglike(np.asarray(y, dtype=np.float64), np.asarray(data, dtype=np.float64), float(sigma), n_data))
