"""Shadow of the reference's methanation settings module (SMC_methanation/methanation_set_conditon.py; the
misspelling is the reference's).  Same names, defaults and import-time side effects: seed the global NumPy RNG
(:15), prior box (:55-70), physical constants (:74-89), DAE variable flags (:94-103), SMC hyper-parameters
(:107-132), and the inlet conditions read from methanation_data/information.csv relative to the working
directory (:137-214).  NB: like the reference, only the FIRST n_data rows of the sliced table are converted
(the loops run over range(n_data) although the slice holds datafin-datastart+1 rows)."""
import numpy as np
import pandas as pd

np.random.seed(20250205)

n_state = 9
num_model_params = 8
est_params_list = [1, 1, 1, 1, 0, 0, 0, 0, 1]
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

NX = 51
datalist = [0, 2, 5, 6, 8, 9, 10, 11, 13, 14, 15, 16, 17, 19, 20, 21, 22, 25, 26, 27, 28, 31, 35, 38, 40, 45, 49, 52, 55, 58]
datastart = datalist[0]
datafin = datalist[-1]
n_data = len(datalist)

sigma_true = 5
baseparams = np.array([13.04, 52.2e3, 1.147e5, 96.7e3, 23.34, -6, 0.72, -2.51e3])
baseparams_withsigma = np.append(baseparams, sigma_true)
use_params = np.append(baseparams, sigma_true)
high_k = [25, 1, 30, 2, 1, -2, 1, -2, 2]
low_k = [4, 1, 4, 1, 1, -2, 1, -2, 0.9]
high_limit = use_params + use_params * np.array(high_k)
low_limit = use_params - use_params * np.array(low_k)
high_limit_array = np.array([high_limit[i] for i in est_position])
low_limit_array = np.array([low_limit[i] for i in est_position])

pi = np.pi
sc = np.array([-4, -1, 1, 2, 0])
Dz = 0.95e-5
rhos = 5075
Hr = -164940
R = 8.3144589
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

n_cores = 30
n_particle = 1000
inv_Np = 1 / n_particle
ess_limit = 0.5
mhstep_factor = 0.5
mhstep_factor_cov = 0.5
ad_mhstep_num = 20
mhstep_num = 5
mhstep_ratio = 1.0
r_threshold = 0.5
r_threshold_f = 0.7
r_threshold_min = 0.1
d_gamma_max = 1
gm_reduction_itr = 80
gm_reduction_rate = 0.7
itr_max = 50
n_hist = 50
fig_dimen = int(n_state * 100 + 11)
w_cov = np.ones((num_est_params, num_est_params))
for _i in range(num_est_params):
    w_cov[_i, :] = mhstep_factor_cov
    w_cov[_i, _i] = mhstep_factor

info_df = pd.read_csv('methanation_data/information.csv').fillna(0)
information = info_df.iloc[datastart:datafin + 1].values

catag = information[:, 2]
reactorlength = information[:, 4]
T_jacket = information[:, 5]
void_fraction = information[:, 6]
T_in = information[:, 7]
P_total = information[:, 9]
in_flow_a, in_flow_b, in_flow_c, in_flow_d, in_flow_e = (information[:, _k] for _k in (10, 11, 12, 14, 15))
in_flow_total = information[:, 16]
out_flow_a, out_flow_b, out_flow_c, out_flow_d, out_flow_e = (information[:, _k] for _k in (17, 18, 19, 21, 22))
out_flow_total = information[:, 23]
out_molf_a, out_molf_b, out_molf_c, out_molf_d, out_molf_e = (information[:, _k] for _k in (24, 25, 26, 28, 29))

Ca_in, Cb_in, Cc_in, Cd_in, Ce_in = (np.zeros(n_data) for _ in range(5))
Xa_out, Xb_out, Xc_out, Xd_out, Xe_out = (np.zeros(n_data) for _ in range(5))
u_in = np.zeros(n_data)
sccm = np.zeros(n_data)
void = np.zeros(n_data)
Fa_out, Fb_out, Fc_out, Fd_out, Fe_out = (np.zeros(n_data) for _ in range(5))
for _i in range(0, n_data):
    T_in[_i] = T_in[_i] + 273
    _tot = in_flow_a[_i] + in_flow_b[_i] + in_flow_c[_i] + in_flow_d[_i] + in_flow_e[_i]
    Ca_in[_i] = (P_total[_i] * 1e6 + 101325) / R / T_in[_i] * in_flow_a[_i] / _tot
    Cb_in[_i] = (P_total[_i] * 1e6 + 101325) / R / T_in[_i] * in_flow_b[_i] / _tot
    Cc_in[_i] = (P_total[_i] * 1e6 + 101325) / R / T_in[_i] * in_flow_c[_i] / _tot
    Cd_in[_i] = (P_total[_i] * 1e6 + 101325) / R / T_in[_i] * in_flow_d[_i] / _tot
    Ce_in[_i] = (P_total[_i] * 1e6 + 101325) / R / T_in[_i] * in_flow_e[_i] / _tot
    Xa_out[_i], Xb_out[_i], Xc_out[_i], Xd_out[_i], Xe_out[_i] = (out_molf_a[_i], out_molf_b[_i], out_molf_c[_i],
                                                                     out_molf_d[_i], out_molf_e[_i])
    T_jacket[_i] = T_jacket[_i] + 273
    catag[_i] = catag[_i] / 1000
    reactorlength[_i] = reactorlength[_i] / 1000
    sccm[_i] = in_flow_total[_i]
    void[_i] = void_fraction[_i]
    Fa_out[_i], Fb_out[_i], Fc_out[_i], Fd_out[_i], Fe_out[_i] = (out_flow_a[_i], out_flow_b[_i], out_flow_c[_i],
                                                                     out_flow_d[_i], out_flow_e[_i])
u_in = in_flow_total * 1.667e-8 / S * (101325 * T_in) / ((P_total * 1e6 + 101325) * 298)
