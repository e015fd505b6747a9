"""Shadow of the reference's methanation settings module (SMC_methanation/methanation_set_conditon.py; the
misspelling is the reference's).  Same names, defaults and import-time side effects: seed the global NumPy RNG
(:15), prior box (:55-70), physical constants (:74-89), DAE variable flags (:94-103), SMC hyper-parameters
(:107-132), and the inlet conditions read from methanation_data/information.csv relative to the working
directory (:137-214).  NB: like the reference, only the FIRST n_data rows of the sliced table are converted
(the loops run over range(n_data) although the slice holds datafin-datastart+1 rows)."""
import numpy as np

from smc_lt_amd import methanation as _M
from smc_lt_amd.driver import SMCSettings as _SMCSettings

import os as _os

np.random.seed(20250205)
# n_particle can be overridden with the environment variable SMC_N_PARTICLE (the reference hard-codes 1000, :108)
_s = _SMCSettings(n_particle=int(_os.environ.get("SMC_N_PARTICLE", "1000")))

n_state = 9
num_model_params = 8
est_params_list = list(_M.EST_PARAMS_LIST)
num_est_params = np.sum(est_params_list)
taylor = False
normal_pred = False
est_sigma = est_params_list[-1] == 1
uni_list = [0, 1, 2, 3, 8]
coefficent = np.array([0.5, 0.5, 0.5, 0.5, 0.3, 0.3, 0.3, 0.3])
coefficent_uni = np.array([0.5] * 8)
est_position = [i for i, x in enumerate(est_params_list) if x == 1]
est_position_set = set(est_position)
uni_list_set = set(uni_list)
def_set = uni_list_set - est_position_set

NX = _M.NX
datalist = list(_M.DATALIST)
datastart = datalist[0]
datafin = datalist[-1]
n_data = len(datalist)

sigma_true = _M.SIGMA_TRUE
baseparams = _M.BASEPARAMS.copy()
baseparams_withsigma = np.append(baseparams, sigma_true)
use_params = np.append(baseparams, sigma_true)
high_k = list(_M.HIGH_K)
low_k = list(_M.LOW_K)
low_limit, high_limit, _ = _M.prior_box()
high_limit_array = np.array([high_limit[i] for i in est_position])
low_limit_array = np.array([low_limit[i] for i in est_position])

pi = np.pi
sc = np.array([-4, -1, 1, 2, 0])
Dz = 0.95e-5
rhos = 5075
Hr = -164940
R = _M.GAS_R
Rr = 0.01 / 2
S = pi * Rr ** 2
Cpg = 2800
Cps = 698
keff = 0.72
dint = 0.005
U = 68.2480
bed = 5.4e-3
ku = 8180
P_stp = 1.013 * 10 ** 5

li = [1] * (6 * NX) + [0] * NX
at = 0.001
atol = [at] * (7 * NX)

# SMC hyper-parameters (:107-132): the engine's own defaults, so that the two cannot drift apart
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
itr_max = _s.itr_max
n_hist = 50
fig_dimen = int(n_state * 100 + 11)
w_cov = np.ones((num_est_params, num_est_params))
for _i in range(num_est_params):
    w_cov[_i, :] = mhstep_factor_cov
    w_cov[_i, _i] = mhstep_factor

# inlet conditions (:137-214): computed in ONE place, smc_lt_amd.methanation.settings_arrays; republished here under the
# reference's module-level names
_A = _M.settings_arrays('methanation_data/information.csv')
information = _A["information"]
(catag, reactorlength, T_jacket, void_fraction, T_in, P_total) = (_A[k] for k in
                                                                  ("catag", "reactorlength", "T_jacket", "void_fraction", "T_in", "P_total"))
in_flow_a, in_flow_b, in_flow_c, in_flow_d, in_flow_e = (_A[f"in_flow_{k}"] for k in "abcde")
in_flow_total = _A["in_flow_total"]
out_flow_a, out_flow_b, out_flow_c, out_flow_d, out_flow_e = (_A[f"out_flow_{k}"] for k in "abcde")
out_flow_total = _A["out_flow_total"]
out_molf_a, out_molf_b, out_molf_c, out_molf_d, out_molf_e = (_A[f"out_molf_{k}"] for k in "abcde")
Ca_in, Cb_in, Cc_in, Cd_in, Ce_in = (_A[f"C{k}_in"] for k in "abcde")
Xa_out, Xb_out, Xc_out, Xd_out, Xe_out = (_A[f"X{k}_out"] for k in "abcde")
Fa_out, Fb_out, Fc_out, Fd_out, Fe_out = (_A[f"F{k}_out"] for k in "abcde")
u_in, sccm, void = _A["u_in"], _A["sccm"], _A["void"]
