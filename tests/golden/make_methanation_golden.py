#!/usr/bin/env python3
"""Golden vectors for the methanation model, generated from the REFERENCE's own functions.

Runs ONLY in the build container.  The reference's experimental-conditions table
methanation_data/information.csv (read at SMC_methanation/methanation_set_conditon.py:137) is NOT in the
reference repository, so a SYNTHETIC table is generated here (seeded; physically plausible ranges,
SURVEY.md section 8(d)) and committed as tests/golden/methanation_information.csv - it is this build's own
data, not reference content.  With it (in a scratch cwd) and the inert stand-ins of tests/golden/_shims the
reference modules methanation_set_conditon / methanation_set_likelihood / methanation_functions import, and
their pure-arithmetic functions are evaluated:

  reaction(t, X, dX, params)   357-equation DAE residual     methanation_set_likelihood.py:69-139
  func_rCH4, func_rohg         rate law, gas density         :44-66
  my_loglike                   Gaussian log-likelihood       :280-300
  cal_prior                    uniform prior product         methanation_functions.py:96-135
  inlet conversions, prior box, algvar flags                 methanation_set_conditon.py:64-70,94-103,188-214

The time integration (my_model -> Assimulo IDA) cannot run here (no Assimulo/SUNDIALS): it stays
"parity unpinned".  Output: tests/golden/methanation_golden.npz (+ the CSV).
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/SMC_methanation"


def make_information_csv(path):
    """59 rows x 30 numeric columns; only the columns the reference reads carry meaning
    (methanation_set_conditon.py:164-186)."""
    rs = np.random.RandomState(20251124)
    n = 59
    a = np.zeros((n, 30))
    a[:, 0] = np.arange(n)
    a[:, 2] = rs.uniform(50, 300, n).round(1)        # catalyst mass [mg]
    a[:, 4] = rs.uniform(20, 60, n).round(1)         # reactor length [mm]
    a[:, 5] = rs.uniform(220, 320, n).round(0)       # jacket temperature [degC]
    a[:, 6] = 0.4                                    # void fraction
    a[:, 7] = rs.uniform(180, 250, n).round(0)       # inlet temperature [degC]
    a[:, 9] = 0.0                                    # gauge pressure [MPa]
    h2 = rs.uniform(150, 400, n).round(0)
    a[:, 10] = h2                                    # H2 [sccm]
    a[:, 11] = (h2 / 4).round(2)                     # CO2
    a[:, 12] = 0.0                                   # CH4
    a[:, 14] = 0.0                                   # H2O
    a[:, 15] = rs.uniform(20, 60, n).round(0)        # Ar
    a[:, 16] = a[:, 10] + a[:, 11] + a[:, 12] + a[:, 14] + a[:, 15]
    # outlet columns (17-29) are only copied into unused arrays by the reference; plausible numbers
    conv = rs.uniform(0.2, 0.9, n)
    a[:, 17] = a[:, 10] - 4 * conv * a[:, 11]
    a[:, 18] = a[:, 11] * (1 - conv)
    a[:, 19] = a[:, 11] * conv
    a[:, 21] = 2 * a[:, 11] * conv
    a[:, 22] = a[:, 15]
    a[:, 23] = a[:, 17] + a[:, 18] + a[:, 19] + a[:, 21] + a[:, 22]
    for c_out, c_in in [(24, 17), (25, 18), (26, 19), (28, 21), (29, 22)]:
        a[:, c_out] = a[:, c_in] / a[:, 23]
    header = ",".join(f"c{i}" for i in range(30))
    np.savetxt(path, a, delimiter=",", header=header, comments="", fmt="%.10g")


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not present")
    csv_path = os.path.join(HERE, "methanation_information.csv")
    make_information_csv(csv_path)
    scratch = tempfile.mkdtemp(prefix="golden_meth_")
    os.makedirs(os.path.join(scratch, "methanation_data"))
    os.symlink(csv_path, os.path.join(scratch, "methanation_data", "information.csv"))
    os.chdir(scratch)
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path.insert(0, os.path.join(HERE, "_shims"))
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    import methanation_set_conditon as C
    import methanation_set_likelihood as L
    import methanation_functions as F

    NX, n_data = C.NX, C.n_data
    rs = np.random.RandomState(5)
    # initial guess exactly as the driver builds it (SMC_methanation_main.py:47-58)
    guess = np.zeros([n_data, 7 * NX])
    for i in range(n_data):
        g0 = np.ones(7 * NX)
        g0[0:NX] = C.Ca_in[i]; g0[NX:2 * NX] = C.Cb_in[i]; g0[2 * NX:3 * NX] = C.Cc_in[i]
        g0[3 * NX:4 * NX] = C.Cd_in[i]; g0[4 * NX:5 * NX] = C.Ce_in[i]
        g0[5 * NX:6 * NX] = C.T_in[i]; g0[5 * NX + 1:6 * NX] = 400; g0[6 * NX:7 * NX] = C.u_in[i]
        guess[i] = g0

    def p0_of(i, pr):
        return np.array([C.Ca_in[i], C.Cb_in[i], C.Cc_in[i], C.Cd_in[i], C.Ce_in[i], C.T_in[i], C.T_jacket[i], C.u_in[i],
                         C.void[i], C.reactorlength[i] / (NX - 1), *pr[:8]], dtype=np.float64)

    # ---- residual known answers: guess + perturbations, random dX, several parameter vectors ----
    res_X, res_dX, res_p, res_out = [], [], [], []
    for k in range(24):
        i = k % n_data
        pr = C.baseparams * (1 + 0.3 * rs.uniform(-1, 1, 8)) if k >= 4 else C.baseparams.copy()
        X = guess[i].copy()
        if k >= 2:
            X[:5 * NX] *= 1 + 0.2 * rs.uniform(-1, 1, 5 * NX)
            X[2 * NX:4 * NX] += rs.uniform(0, 2, 2 * NX)          # some CH4 / H2O so that both rates are non-zero
            X[5 * NX:6 * NX] = rs.uniform(450, 650, NX)
            X[6 * NX:] *= 1 + 0.2 * rs.uniform(-1, 1, NX)
        dX = np.zeros(7 * NX) if k % 3 == 0 else rs.standard_normal(7 * NX) * np.abs(X) * 0.01
        p = p0_of(i, pr)
        out = L.reaction(0.0, X, dX, p)
        res_X.append(X); res_dX.append(dX); res_p.append(p); res_out.append(np.array(out))
    # ---- rate law / density ----
    rc_in = np.column_stack([rs.uniform(400, 700, 64), rs.uniform(0, 20, 64), rs.uniform(0, 6, 64), rs.uniform(0, 5, 64),
                             rs.uniform(0, 8, 64)])
    rc_in[:4, 1] = [0.0, 1e-9, 1e-4, 5e-4]                       # PH2 clamp max(0.001, PH2) (:54)
    rc_par = np.array([C.baseparams * (1 + 0.3 * rs.uniform(-1, 1, 8)) for _ in range(64)])
    rc_out = np.array([L.func_rCH4(*rc_in[j], rc_par[j]) for j in range(64)])
    rg_in = np.column_stack([rs.uniform(0, 20, 32), rs.uniform(0, 6, 32), rs.uniform(0, 5, 32), rs.uniform(0, 8, 32),
                             rs.uniform(0.1, 4, 32), rs.uniform(400, 700, 32), rs.uniform(9e4, 2e5, 32)])
    rg_out = np.array([L.func_rohg(*rg_in[j]) for j in range(32)])
    # ---- likelihood ----
    ll_y = rs.uniform(0, 300, (8, 5, n_data)); ll_d = ll_y + rs.standard_normal((8, 5, n_data)) * 5
    ll_y[7, :, 3] = -10000.0                                      # the failure sentinel (:244-249)
    ll_s = rs.uniform(0.5, 10, 8)
    ll_out = np.array([L.my_loglike(ll_y[j], ll_d[j], ll_s[j], n_data) for j in range(8)])
    # ---- prior ----
    lo, hi = np.array(C.low_limit_array), np.array(C.high_limit_array)
    th = lo + (hi - lo) * rs.uniform(-0.2, 1.2, (200, len(lo)))
    th[0] = lo; th[1] = hi; th[2] = lo - 1e-12 * np.abs(lo); th[3] = hi * (1 + 1e-15)
    F.n_particle = len(th)
    pri = F.cal_prior(th)

    np.savez_compressed(
        os.path.join(HERE, "methanation_golden.npz"),
        NX=np.int64(NX), n_data=np.int64(n_data), datalist=np.array(C.datalist), est_position=np.array(C.est_position),
        baseparams=C.baseparams, low_limit=C.low_limit, high_limit=C.high_limit, low_limit_array=lo, high_limit_array=hi,
        algvar=np.array(C.li), sc=C.sc.astype(np.float64),
        consts=np.array([C.Dz, C.rhos, C.Hr, C.R, C.Rr, C.S, C.Cpg, C.Cps, C.keff, C.dint, C.U, C.P_stp]),
        Ca_in=C.Ca_in, Cb_in=C.Cb_in, Cc_in=C.Cc_in, Cd_in=C.Cd_in, Ce_in=C.Ce_in, T_in=np.asarray(C.T_in, dtype=np.float64),
        T_jacket=np.asarray(C.T_jacket, dtype=np.float64), u_in=np.asarray(C.u_in, dtype=np.float64),
        void=C.void, reactorlength=np.asarray(C.reactorlength, dtype=np.float64), guess=guess,
        res_X=np.array(res_X), res_dX=np.array(res_dX), res_p=np.array(res_p), res_out=np.array(res_out),
        rc_in=rc_in, rc_par=rc_par, rc_out=rc_out, rg_in=rg_in, rg_out=rg_out,
        ll_y=ll_y, ll_d=ll_d, ll_s=ll_s, ll_out=ll_out, prior_theta=th, prior_pdf=pri,
        w_cov=C.w_cov, n_state=np.int64(C.n_state), sigma_true=np.float64(C.sigma_true),
    )
    print("[golden] methanation: residual cases", len(res_out), "max |res|", np.abs(np.array(res_out)).max())
    print("[golden] my_loglike(arange) check:", L.my_loglike(np.arange(150.).reshape(5, 30), np.arange(150.).reshape(5, 30) + 1, 5.0, 30))


if __name__ == "__main__":
    main()
