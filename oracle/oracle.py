"""oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the reference's likelihood-tempered SMC particle loop, used as
the parity checker.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product package never does.

Structure
---------
* vectorised driver statements are the SAME NumPy calls the reference makes, in the
  same order, on the same global legacy RNG (np.random.*), so that a run with the
  reference's seed consumes the identical random stream
  (reference: SMC_example/Micmem_SMC_main.py:98-262, cited line by line below);
* scalar hot loops (SciPy RK45 per particle x experiment, the pure-Python resampling
  loop) are restated in C in oracle/smc_oracle.c and called through ctypes;
* `loglik="scipy"` swaps the C RK45 for scipy.integrate.solve_ivp itself, i.e. the
  reference's own third-party call (Micmem_likelihood.py:24-30) - this is the
  "NumPy/SciPy counterpart" timed as bench.py's cpu_baseline (kind "port").

Parity status: PINNED by tests/golden/ (fixtures generated from the reference run
here by tests/golden/make_golden.py); see tests/test_oracle_golden.py.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int64_p = ctypes.POINTER(ctypes.c_int64)


class _RkStats(ctypes.Structure):
    _fields_ = [("n_attempts", ctypes.c_int64), ("n_accepted", ctypes.c_int64), ("nfev", ctypes.c_int64),
                ("status", ctypes.c_int32), ("n_out", ctypes.c_int32)]


def build(force: bool = False) -> str:
    """Compile oracle/smc_oracle.c -> oracle/libsmc_oracle.so (gcc, no FMA contraction)."""
    so = os.path.join(_HERE, "libsmc_oracle.so")
    src = os.path.join(_HERE, "smc_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsmc_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.oracle_mm_rk45.restype = _RkStats
        L.oracle_mm_rk45.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_double, c_double_p, ctypes.c_int,
                                     ctypes.c_double, ctypes.c_double, c_double_p]
        L.oracle_np_sum.restype = ctypes.c_double
        L.oracle_np_sum.argtypes = [c_double_p, ctypes.c_int64]
        L.oracle_mm_loglik.restype = ctypes.c_double
        L.oracle_mm_loglik.argtypes = [c_double_p, c_double_p, c_double_p, c_double_p, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_double_p,
                                       ctypes.POINTER(ctypes.c_int), c_int64_p]
        L.oracle_mm_loglik_batch.restype = None
        L.oracle_mm_loglik_batch.argtypes = [c_double_p, ctypes.c_int64, c_double_p, c_double_p, c_double_p,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                             ctypes.c_double, ctypes.c_double, c_double_p, c_double_p, c_int64_p,
                                             c_int64_p, ctypes.c_int]
        L.oracle_resample_residual_systematic.restype = ctypes.c_int64
        L.oracle_resample_residual_systematic.argtypes = [c_double_p, ctypes.c_int64, ctypes.c_double, c_int64_p,
                                                          c_double_p, c_double_p, ctypes.c_int, c_double_p,
                                                          c_double_p, c_int64_p]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(c_double_p)


# --------------------------------------------------------------------------------------
# data + settings (reference: SMC_example/Micmem_settings.py)
# --------------------------------------------------------------------------------------
@dataclass
class MMData:
    """The 6 x 40 pseudo-data set (Micmem_settings.py:103-115): t, P_obs per experiment, S0."""
    t: np.ndarray      # (n_ex, n_t)
    P_obs: np.ndarray  # (n_ex, n_t)
    S0: np.ndarray     # (n_ex,)

    @property
    def n_ex(self):
        return self.t.shape[0]

    @property
    def n_t(self):
        return self.t.shape[1]

    @staticmethod
    def load(path: str | None = None) -> "MMData":
        path = path or os.path.join(os.path.dirname(_HERE), "tests", "golden", "mm_data.npz")
        z = np.load(path)
        return MMData(np.ascontiguousarray(z["t"], dtype=np.float64),
                      np.ascontiguousarray(z["P_obs"], dtype=np.float64),
                      np.ascontiguousarray(z["S0"], dtype=np.float64))


@dataclass
class SMCSettings:
    """Hyper-parameters with the reference's names and defaults (Micmem_settings.py:15-31,90)."""
    n_particle: int = 1000
    ess_limit: float = 0.5
    mhstep_factor: float = 0.5
    mhstep_factor_cov: float = 0.5
    ad_mhstep_num: int = 20
    mhstep_num: int = 5
    r_threshold: float = 0.5
    r_threshold_f: float = 0.7
    r_threshold_min: float = 0.1
    d_gamma_max: float = 1
    gm_reduction_itr: int = 80
    gm_reduction_rate: float = 0.7
    itr_max: int = 50
    est_sigma: bool = True
    sigma_true: float = 5
    rtol: float = 1e-3   # SciPy solve_ivp defaults (Micmem_likelihood.py:24-30 passes none)
    atol: float = 1e-6
    priors: dict = field(default_factory=lambda: {
        "Vmax": {"dist": "uniform", "low": 0, "high": 10},
        "Km": {"dist": "uniform", "low": 0, "high": 10},
        "sigma": {"dist": "uniform", "low": 0, "high": 10},
    })

    @property
    def num_est_params(self):
        return len(self.priors)

    def w_cov(self):
        """Micmem_settings.py:94-97."""
        d = self.num_est_params
        w = np.ones((d, d))
        for i in range(d):
            w[i, :] = self.mhstep_factor_cov
            w[i, i] = self.mhstep_factor
        return w


def sample_prior(priors: dict, n_particle: int) -> np.ndarray:
    """Micmem_settings.py:69-87: one draw of size N per parameter, parameter-major, global RNG."""
    p_pred = np.zeros((n_particle, len(priors)))
    for j, (name, p) in enumerate(priors.items()):
        if p["dist"] in ("normal", "flat"):
            p_pred[:, j] = np.random.normal(loc=p["mu"], scale=p["sigma"], size=n_particle)
        elif p["dist"] == "uniform":
            p_pred[:, j] = np.random.uniform(low=p["low"], high=p["high"], size=n_particle)
        else:
            raise ValueError(f"Unknown distribution: {p['dist']}")
    return p_pred


def cal_prior(theta: np.ndarray, priors: dict) -> np.ndarray:
    """Micmem_SMC_main.py:60-90 (scipy.stats pdf product)."""
    import scipy.stats
    pdf_vals = np.zeros((theta.shape[0], len(priors)))
    for j, (name, cfg) in enumerate(priors.items()):
        x = theta[:, j]
        if cfg["dist"] == "normal":
            pdf_vals[:, j] = scipy.stats.norm.pdf(x, loc=cfg["mu"], scale=cfg["sigma"])
        elif cfg["dist"] == "uniform":
            pdf_vals[:, j] = scipy.stats.uniform.pdf(x, loc=cfg["low"], scale=cfg["high"] - cfg["low"])
        elif cfg["dist"] == "flat":   # no factor: sigma under normal_pred without taylor (methanation_functions.py:132-138)
            pdf_vals[:, j] = 1.0
        else:
            raise ValueError(f"Unknown prior: {cfg['dist']}")
    return np.prod(pdf_vals, axis=1)


# --------------------------------------------------------------------------------------
# likelihood (reference: SMC_example/Micmem_likelihood.py)
# --------------------------------------------------------------------------------------
def rk45_solve(Vmax, Km, S0, t_eval, rtol=1e-3, atol=1e-6):
    """C restatement of solve_ivp(RK45, t_eval=...) for the MM ODE. Returns (S(t_eval), stats dict)."""
    t_eval = np.ascontiguousarray(t_eval, dtype=np.float64)
    y = np.full(t_eval.shape[0], np.nan)
    st = lib().oracle_mm_rk45(float(Vmax), float(Km), float(S0), _p(t_eval), t_eval.shape[0], rtol, atol, _p(y))
    return y, {"n_attempts": st.n_attempts, "n_accepted": st.n_accepted, "nfev": st.nfev, "status": st.status,
               "n_out": st.n_out}


def mm_loglik_batch(particle, data: MMData, est_sigma=True, sigma_fixed=5.0, rtol=1e-3, atol=1e-6,
                    want_pred=False, n_threads=0):
    """sim_particle (Micmem_likelihood.py:79-92) over an (N,3) array -> (lk[N], pred|None, info)."""
    particle = np.ascontiguousarray(particle, dtype=np.float64)
    assert particle.ndim == 2 and particle.shape[1] == 3
    N = particle.shape[0]
    lk = np.empty(N)
    pred = np.empty((N, data.n_ex, data.n_t)) if want_pred else None
    nfail = ctypes.c_int64(0)
    stats = (ctypes.c_int64 * 3)()
    lib().oracle_mm_loglik_batch(_p(particle), N, _p(data.t), _p(data.P_obs), _p(data.S0), data.n_ex, data.n_t,
                                 int(est_sigma), float(sigma_fixed), rtol, atol, _p(lk),
                                 _p(pred) if want_pred else None, ctypes.byref(nfail), stats, int(n_threads))
    info = {"n_failed": nfail.value, "n_attempts": stats[0], "n_accepted": stats[1], "nfev": stats[2]}
    return lk, pred, info


def mm_loglik_scipy_one(params, data: MMData, est_sigma=True, sigma_fixed=5.0):
    """The reference's per-particle statements with scipy.solve_ivp itself
    (Micmem_likelihood.py:17-77): the NumPy/SciPy counterpart timed as cpu_baseline."""
    from scipy.integrate import solve_ivp
    Vmax, Km = params[0], params[1]
    sigma = params[-1] if est_sigma else sigma_fixed
    if sigma <= 0:
        return -np.inf
    logL_total = 0.0
    for i in range(data.n_ex):
        t = data.t[i]
        S0 = data.S0[i]
        sol = solve_ivp(fun=lambda tt, S: -Vmax * S / (Km + S), t_span=(t[0], t[-1]), y0=[S0], t_eval=t,
                        method="RK45")
        P_model = S0 - sol.y[0]
        residual = data.P_obs[i] - P_model
        logL_i = -0.5 * data.n_t * np.log(2 * np.pi * sigma ** 2) - np.sum(residual ** 2) / (2 * sigma ** 2)
        logL_total += logL_i
    return logL_total


_SCIPY_POOL_STATE = {}


def _scipy_chunk(idx):
    particle, data, est_sigma, sigma_fixed = _SCIPY_POOL_STATE["args"]
    return [mm_loglik_scipy_one(particle[i], data, est_sigma, sigma_fixed) for i in idx]


def mm_loglik_batch_scipy(particle, data: MMData, est_sigma=True, sigma_fixed=5.0, n_workers=1):
    """One task per particle on a fork pool: stand-in for the reference's Ray fan-out."""
    particle = np.ascontiguousarray(particle, dtype=np.float64)
    N = particle.shape[0]
    if n_workers <= 1:
        return np.array([mm_loglik_scipy_one(particle[i], data, est_sigma, sigma_fixed) for i in range(N)])
    import multiprocessing as mp
    _SCIPY_POOL_STATE["args"] = (particle, data, est_sigma, sigma_fixed)
    chunks = np.array_split(np.arange(N), min(N, n_workers * 8))
    with mp.get_context("fork").Pool(n_workers) as pool:
        out = pool.map(_scipy_chunk, chunks)
    _SCIPY_POOL_STATE.clear()
    return np.array([x for c in out for x in c])


# --------------------------------------------------------------------------------------
# the stages of the tempering loop (reference: SMC_example/Micmem_SMC_main.py)
# --------------------------------------------------------------------------------------
def ess_search(lk, gamma_old, s: SMCSettings):
    """Micmem_SMC_main.py:111-144. Returns dict(gamma_new, ess, sum_weight, max_lk, p_weight, iters)."""
    n_particle = lk.shape[0]
    gamma_new = gamma_old + s.d_gamma_max
    if gamma_new > 1.0:
        gamma_new = 1.0
    max_lk = np.max(lk)
    d_lk = lk - max_lk
    iters = 0
    for i in range(s.gm_reduction_itr):
        gm = gamma_new - gamma_old
        p_weight = np.exp(d_lk * gm)
        sum_weight = np.sum(p_weight)
        p_weight = p_weight / sum_weight
        ess = np.sum(p_weight ** 2)
        ess = 1.0 / ess / n_particle
        iters += 1
        if ess > s.ess_limit:
            break
        gamma_new = (gamma_new - gamma_old) * s.gm_reduction_rate + gamma_old
    # gm is the increment the final weights were computed with (differs from gamma_new-gamma_old only when
    # all gm_reduction_itr trials failed, :143-144)
    return {"gamma_new": gamma_new, "ess": ess, "sum_weight": sum_weight, "max_lk": max_lk, "p_weight": p_weight,
            "iters": iters, "gm": gm, "d_lk": d_lk}


def ess_candidates(gamma_old, s: SMCSettings):
    """The fixed grid of increments the back-off visits (:121,141), computed with the reference's recurrence."""
    gamma_new = gamma_old + s.d_gamma_max
    if gamma_new > 1.0:
        gamma_new = 1.0
    gms, gammas = [], []
    for i in range(s.gm_reduction_itr):
        gms.append(gamma_new - gamma_old)
        gammas.append(gamma_new)
        gamma_new = (gamma_new - gamma_old) * s.gm_reduction_rate + gamma_old
    return np.array(gms), np.array(gammas), gamma_new


def resample(p_weight, wrand_u, p_pred, lk, p_filt, lk1):
    """Micmem_SMC_main.py:147-184 (C restatement of the pure-Python loop). p_filt/lk1 are updated in place.
    Returns (p_is, n_written, n_tmp)."""
    N, d = p_pred.shape
    w = np.ascontiguousarray(p_weight, dtype=np.float64).copy()
    p_is = np.zeros(N, dtype=np.int64)
    p_pred_c = np.ascontiguousarray(p_pred, dtype=np.float64)
    lk_c = np.ascontiguousarray(lk, dtype=np.float64)
    assert p_filt.flags.c_contiguous and lk1.flags.c_contiguous
    n_tmp = ctypes.c_int64(0)
    n = lib().oracle_resample_residual_systematic(_p(w), N, float(wrand_u), p_is.ctypes.data_as(c_int64_p),
                                                  _p(p_pred_c), _p(lk_c), d, _p(p_filt), _p(lk1),
                                                  ctypes.byref(n_tmp))
    if n > N:
        raise IndexError(f"resampling produced {n} > {N} rows (the reference raises IndexError at :180)")
    return p_is, n, n_tmp.value


def resample_python(p_weight, wrand_u, p_pred, lk, p_filt, lk1):
    """The same loop in pure Python, literally as the reference writes it (small N only; pins the C version)."""
    n_particle = p_pred.shape[0]
    inv_Np = 1 / n_particle
    p_is = np.trunc(p_weight * n_particle).astype(int)
    p_weight = p_weight - p_is * inv_Np
    n_tmp = n_particle - np.sum(p_is)
    wrand = wrand_u * inv_Np
    sum_ = 0.0
    n = 0
    p_pred_copy = p_pred.copy()
    for j in range(n_particle):
        sum_ += p_weight[j]
        if sum_ >= wrand:
            p_is[j] += 1
            wrand += inv_Np
            n_tmp -= 1
        for k in range(p_is[j]):
            p_filt[n, :] = p_pred_copy[j, :]
            lk1[n] = lk[j]
            n += 1
    return p_is, n, n_tmp


@dataclass
class StepRecord:
    step: int
    gamma_old: float
    gamma_new: float
    gm: float
    ess: float
    ess_iters: int
    sum_weight: float
    max_lk: float
    dlogZ: float
    wrand_u: float
    p_is: np.ndarray
    n_tmp: int
    last_j: int
    n_accept: float
    mh: list  # per MH iteration: dict(cov_m, noise, rr, p0, r, lk2, mhstep_ratio, proposals)


def run_smc(data: MMData, s: SMCSettings | None = None, seed: int | None = 20250205, loglik="c", n_threads=0,
            record_mh=True, verbose=False, p_pred0=None, resample_impl="c"):
    """The reference's driver (Micmem_SMC_main.py:95-262) statement by statement on the global NumPy RNG.

    seed=None leaves the global stream as it is.  Returns dict with final particles, lk, per-step records,
    every likelihood sweep (inputs/outputs, in order) and log-evidence.

    resample_impl="python" runs the resampling loop as the reference writes it, in pure Python (main:147-184) - with
    loglik="scipy" that makes the whole run the NumPy/SciPy counterpart bench.py times as `cpu_baseline`;
    `stage_s` in the result holds the wall time of the likelihood sweeps, the ESS searches and the resampling loops.
    """
    import time as _time
    stage_s = {"likelihood": 0.0, "ess_search": 0.0, "resample": 0.0}
    s = s or SMCSettings()
    n_particle = s.n_particle
    d = s.num_est_params
    inv_Np = 1 / n_particle
    w_cov = s.w_cov()
    sweeps = []

    def sim_particle(p):
        t_in = _time.perf_counter()
        try:
            return _sim_particle(p)
        finally:
            stage_s["likelihood"] += _time.perf_counter() - t_in

    def _sim_particle(p):
        if loglik == "c":
            lk_, _, info = mm_loglik_batch(p, data, s.est_sigma, s.sigma_true, s.rtol, s.atol, n_threads=n_threads)
            if info["n_failed"]:
                raise RuntimeError("an RK45 solve failed; the reference raises here (ragged sol.y)")
        elif loglik == "scipy":
            lk_ = mm_loglik_batch_scipy(p, data, s.est_sigma, s.sigma_true, n_workers=max(1, n_threads))
        else:
            lk_ = np.asarray(loglik(p), dtype=np.float64)
        sweeps.append((p.copy(), lk_.copy()))
        return lk_

    if seed is not None:
        np.random.seed(seed)                       # Micmem_settings.py:47
    p_pred = sample_prior(s.priors, n_particle) if p_pred0 is None else np.array(p_pred0, dtype=np.float64)
    p_filt = np.zeros((n_particle, d))             # Micmem_settings.py:118
    lk1 = np.zeros(n_particle)                     # :127
    lk = sim_particle(p_pred)                      # main:98
    gamma_old = 0.0
    gamma_new = 1.0
    records = []
    logZ = 0.0
    n_mutation_sweeps = 0
    n_ess_iters = 0
    step = 0
    for step in range(1, s.itr_max):               # :109
        t_in = _time.perf_counter()
        es = ess_search(lk, gamma_old, s)          # :111-144
        stage_s["ess_search"] += _time.perf_counter() - t_in
        gamma_new, ess, p_weight, max_lk = es["gamma_new"], es["ess"], es["p_weight"], es["max_lk"]
        n_ess_iters += es["iters"]
        # log-evidence increment (not in the reference):
        # weights are uniform on entry, so Z_t/Z_{t-1} = mean(exp(gm*lk)) = exp(gm*max_lk) * sum_weight / N
        dlogZ = es["gm"] * max_lk + math.log(es["sum_weight"] / n_particle)
        logZ += dlogZ
        wrand_u = np.random.rand()                 # :156
        t_in = _time.perf_counter()
        p_is, n_written, n_tmp = (resample_python if resample_impl == "python" else resample)(p_weight, wrand_u, p_pred, lk, p_filt, lk1)   # :147-184
        stage_s["resample"] += _time.perf_counter() - t_in
        r_ac = np.zeros(n_particle)                # :187
        mhstep_ratio = 1.0                         # :190
        if gamma_new >= 1.0:                       # :193-208
            nMH, r_th = s.ad_mhstep_num, s.r_threshold_f
        else:
            nMH, r_th = s.mhstep_num, s.r_threshold
        mh_records = []
        j = 0
        for j in range(nMH):                       # :209
            cov_m = np.cov(p_filt.T, bias=True)    # :212
            cov_m = cov_m * w_cov                  # :215
            noise = np.random.multivariate_normal(np.zeros(d), cov_m, n_particle)   # :220
            ratio_used = mhstep_ratio
            p_pred = p_filt + noise * mhstep_ratio
            p0_2 = cal_prior(p_pred, s.priors)     # :225
            p0 = np.int32(p0_2 > 0)                # :226
            prior_mode = getattr(s, "prior_mode", "mask")
            if prior_mode != "mask":               # SMC_methanation_main.py:323-324 / :359-360
                with np.errstate(divide="ignore", invalid="ignore"):
                    pratio = p0_2 / cal_prior(p_filt, s.priors)
            if prior_mode != "ratio":
                p_pred = p_pred * p0[:, None] + p_filt * (1.0 - p0[:, None])   # :228
            lk2 = sim_particle(p_pred)             # :229
            n_mutation_sweeps += 1
            px = lk2 - lk1                         # :231
            with np.errstate(over="ignore"):
                if prior_mode == "mask":
                    pp = np.exp(px * gamma_new) * p0   # :233
                elif prior_mode == "ratio_mask":
                    pp = np.exp(px * gamma_new) * pratio * p0    # SMC_methanation_main.py:343
                else:
                    pp = np.exp(px * gamma_new) * pratio         # SMC_methanation_main.py:367
            rr = np.random.uniform(0, 1, n_particle)   # :235
            r = np.int32(pp >= rr)                 # :236
            p_filt = p_pred * r[:, None] + p_filt * (1.0 - r[:, None])   # :238
            lk1 = lk2 * r + lk1 * (1.0 - r)        # :240
            r_ac = np.maximum(r_ac, r)             # :241
            if record_mh:
                mh_records.append({"cov_m": cov_m, "noise": noise, "rr": rr, "p0": p0, "r": r, "lk2": lk2,
                                   "mhstep_ratio": ratio_used, "proposals": p_pred})
            if r_ac.sum() > r_th * n_particle:     # :243
                break
            if r_ac.sum() < s.r_threshold_min * n_particle:   # :247
                mhstep_ratio = mhstep_ratio * 0.5
        p_pred = p_filt.copy()                     # :251
        lk = lk1.copy()                            # :252
        if verbose:
            print(f"iteration:{step}, nMH:{j}, ESS:{ess}, Max Likelihood:{max_lk}, New Gamma:{gamma_new}, "
                  f"Number of Adoption:{r_ac.sum()}")
        records.append(StepRecord(step, gamma_old, gamma_new, es["gm"], ess, es["iters"], es["sum_weight"], max_lk,
                                  dlogZ, wrand_u, p_is, n_tmp, j, float(r_ac.sum()), mh_records))
        if gamma_new == 1.0:                       # :259
            break
        gamma_old = gamma_new
    return {"p_pred": p_pred, "lk": lk, "records": records, "sweeps": sweeps, "logZ": logZ, "gamma": gamma_new,
            "step": step, "n_mutation_sweeps": n_mutation_sweeps, "n_ess_iters": n_ess_iters, "stage_s": stage_s}
