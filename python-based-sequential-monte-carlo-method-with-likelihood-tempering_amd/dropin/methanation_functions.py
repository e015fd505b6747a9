"""Shadow of the hot-path part of SMC_methanation/methanation_functions.py (:44-135): the likelihood fan-out
(`sim_particle`, `cal_parallel_new`) and the prior density (`cal_prior`).  The plotting / CSV helpers of the
reference (:139-272) are out of scope.  One batched GPU call replaces the one-Ray-task-per-particle fan-out."""
import numpy as np
import scipy.stats

from methanation_set_conditon import *  # noqa: F401,F403
from methanation_set_likelihood import *  # noqa: F401,F403
from methanation_set_likelihood import my_model_batch as _my_model_batch
from smc_lt_amd import methanation as _gpu


def cal_parallel_new(params, obs_data, initial_guess):
    """(lk, molfraction) for one 9-vector (8 kinetic parameters + sigma), :44-65."""
    params = np.asarray(params, dtype=np.float64)
    sigma = params[-1] if est_sigma else sigma_true   # noqa: F405
    ycal1, molfraction = my_model(params, initial_guess)   # noqa: F405
    lk = my_loglike(ycal1, obs_data, sigma, n_data, scale=1.0)   # noqa: F405
    return lk, molfraction


cal_parallel_new.remote = cal_parallel_new


def sim_particle(particle, initial_guess, obs_data, p_pred_bases):
    """:70-92 - scatter the estimated columns into the base rows, evaluate every particle, return (llk, C_l_)."""
    print('sim_particle')
    p_pred_bases[:, est_position] = particle   # noqa: F405
    rows = np.ascontiguousarray(p_pred_bases[:len(particle)], dtype=np.float64)
    Flow, mol = _my_model_batch(rows, initial_guess)
    sigma = rows[:, -1] if est_sigma else np.full(len(rows), float(sigma_true))   # noqa: F405
    llk = _gpu.my_loglike(Flow, np.asarray(obs_data, dtype=np.float64), sigma, n_data)   # noqa: F405
    return llk, list(mol)


def cal_prior(theta):
    """:96-135, uniform branch (normal_pred = False is the reference's setting; the Gaussian branches are dead)."""
    if normal_pred:   # noqa: F405
        raise NotImplementedError("normal_pred=True is dead code in the reference (methanation_set_conditon.py:23)")
    dl = high_limit_array - low_limit_array   # noqa: F405
    p = scipy.stats.uniform.pdf(theta, [low_limit[i] for i in est_position], dl)   # noqa: F405
    return np.prod(p.T, axis=0)
