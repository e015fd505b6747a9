"""Shadow of the reference's settings module for the Michaelis-Menten example.

Drop-in boundary: the reference's driver and likelihood module obtain every
hyper-parameter, the prior sample, the data set and the work buffers by `from Micmem_settings import *`
(SMC_example/Micmem_SMC_main.py:28, Micmem_likelihood.py:11).  This module provides the same NAMES with
the same defaults and the same import-time side effects, in the same order:
  seed the global NumPy RNG (Micmem_settings.py:47) -> draw the prior sample parameter-major
  (:69-87) -> load data/mm_pseudo_data_{i}.csv relative to the working directory (:103-115) ->
  allocate the work buffers (:118-127).
The values come from the engine's SMCSettings so that the two cannot drift apart.

n_particle can be overridden with the environment variable SMC_N_PARTICLE (the reference hard-codes 1000).
"""
import os as _os

import numpy as np
import pandas as pd

from smc_lt_amd.driver import SMCSettings as _SMCSettings
from smc_lt_amd.driver import sample_prior as _sample_prior_impl

_s = _SMCSettings(n_particle=int(_os.environ.get("SMC_N_PARTICLE", "1000")))

# ---- SMC hyper-parameters (Micmem_settings.py:15-31) ----
n_cores = 30
n_particle = _s.n_particle
inv_Np = 1 / n_particle
ess_limit = _s.ess_limit
mhstep_factor = _s.mhstep_factor
mhstep_factor_cov = _s.mhstep_factor_cov
ad_mhstep_num = _s.ad_mhstep_num
mhstep_num = _s.mhstep_num
mhstep_ratio = 1.0
r_threshold = _s.r_threshold
r_threshold_f = _s.r_threshold_f
r_threshold_min = _s.r_threshold_min
d_gamma_max = _s.d_gamma_max
gm_reduction_itr = _s.gm_reduction_itr
gm_reduction_rate = _s.gm_reduction_rate

# ---- global settings (:38-53) ----
coefficent = np.array([0.5, 0.5, 0.5])
coefficent_uni = np.array([0.5] * 8)
sigma_true = _s.sigma_true
np.random.seed(_s.seed)
num_est_params = 3
num_model_params = 2
est_params_list = [1, 1, 1]
est_sigma = _s.est_sigma
priors = _s.priors


def sample_prior(priors, n_particle):
    """dict name -> N draws, parameter-major on the global RNG (:69-82)."""
    arr = _sample_prior_impl(priors, n_particle)
    return {name: arr[:, j] for j, name in enumerate(priors.keys())}


samples = sample_prior(priors, n_particle)
p_pred = np.zeros((n_particle, num_est_params))
for j, name in enumerate(priors.keys()):      # j, name (and i, df, data_i below) stay behind as module globals in the
    p_pred[:, j] = samples[name]               # reference too, and `import *` hands them on: kept under the same names

itr_max = _s.itr_max
n_hist = 50
fig_dimen = int(num_est_params * 100 + 11)
w_cov = _s.w_cov()

# ---- experimental data (:103-115) ----
dataset = []
n_ex = 6
base_path = "data/mm_pseudo_data"
for i in range(0, n_ex):
    df = pd.read_csv(f"{base_path}_{i}.csv")
    data_i = {"t": df["t"].values, "P_obs": df["P_obs"].values, "S0": df["S_true"].iloc[0]}
    dataset.append(data_i)
datapoint = len(dataset[0]["t"])
obs_data = dataset

# ---- work buffers (:118-127) ----
p_filt = np.zeros((n_particle, num_est_params))
p_weight = np.ones(n_particle) / n_particle
p_is = np.zeros(n_particle, dtype=int)
y_cal = np.zeros((n_particle, n_ex))
d_lk = np.zeros(n_particle)
lk1 = np.zeros(n_particle)
