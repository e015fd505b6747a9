"""Shadow of the reference's likelihood module: the callback surface of the hot path
(SMC_example/Micmem_likelihood.py), backed by the HIP engine.

  sim_particle(particle) -> (llk, C_l_)        Micmem_likelihood.py:79-92
  log_likelihood_mm_multi(params)               :35-77   (also reachable as .remote(params))
  simulate_mm_on_grid(Vmax, Km, S0, t_array)    :17-33
  mm_ode(t, S, Vmax, Km)                        :14-15

Every likelihood value is computed by libsmc_hip.so on the GPU (one persistent RK45 kernel for the whole
batch instead of one Ray task per particle); nothing here falls back to the CPU.  Like the reference this
module star-imports the settings module, so `dataset`, `n_ex`, `n_particle`, `est_sigma`, `sigma_true`
are the caller's.  C_l_ (the 6 x 40 model predictions per particle, used only by the plot helpers) is
computed lazily on first access.
"""
import numpy as np

from Micmem_settings import *  # noqa: F401,F403  (the reference does the same, :11)
from smc_lt_amd.engine import HipEngine as _HipEngine

_ENGINE = None


def _engine(n):
    """One context per process, sized for the largest batch seen so far."""
    global _ENGINE
    if _ENGINE is None or _ENGINE.n_local < n:
        if _ENGINE is not None:
            _ENGINE.close()
        _ENGINE = _HipEngine(max(int(n), 1), 3, device=0)
        t = np.stack([np.asarray(d["t"], dtype=np.float64) for d in dataset])          # noqa: F405
        P = np.stack([np.asarray(d["P_obs"], dtype=np.float64) for d in dataset])      # noqa: F405
        S0 = np.array([float(d["S0"]) for d in dataset])                               # noqa: F405
        _ENGINE.set_model_mm(t, P, S0, est_sigma=est_sigma, sigma_fixed=sigma_true)    # noqa: F405
        if hasattr(_ENGINE, "set_exact_pow"):
            _ENGINE.set_exact_pow(True)     # the reference's drivers run on NumPy's stream: libm-grade step-controller power
    return _ENGINE


class _LazyPredictions:
    """C_l_: indexable per particle like the reference's tuple of per-experiment prediction lists."""

    def __init__(self, particle):
        self._particle = particle
        self._pred = None

    def _ensure(self):
        if self._pred is None:
            _, self._pred, _ = _engine(len(self._particle)).loglik_host(self._particle, want_pred=True)
        return self._pred

    def __len__(self):
        return len(self._particle)

    def __getitem__(self, i):
        return list(self._ensure()[i])

    def __iter__(self):
        return (list(row) for row in self._ensure())


def mm_ode(t, S, Vmax, Km):
    return - Vmax * S / (Km + S)


def simulate_mm_on_grid(Vmax, Km, S0, t_array):
    """P_model(t_array) for one experiment (:17-33), solved on the GPU with the same RK45."""
    t_array = np.ascontiguousarray(t_array, dtype=np.float64)
    with _HipEngine(1, 3, device=0) as eng:
        eng.set_model_mm(t_array[None, :], np.zeros((1, len(t_array))), np.array([float(S0)]))
        eng.set_exact_pow(True)
        _, pred, info = eng.loglik_host(np.array([[Vmax, Km, 1.0]]), want_pred=True)
    if info["n_failed"]:
        raise RuntimeError("RK45 did not reach t_bound (SciPy: status -1)")
    return pred[0, 0]


def log_likelihood_mm_multi(params):
    """(logL_total, P_model_list) for one parameter vector; bare -inf when sigma <= 0 (:53-54)."""
    params = np.asarray(params, dtype=np.float64)
    sigma = params[-1] if est_sigma else sigma_true    # noqa: F405
    if sigma <= 0:
        return -np.inf
    lk, pred, info = _engine(1).loglik_host(params[None, :3], want_pred=True)
    if info["n_failed"]:
        raise ValueError("RK45 failed before reaching the last data time (the reference raises here too)")
    return float(lk[0]), list(pred[0])


log_likelihood_mm_multi.remote = log_likelihood_mm_multi   # call sites written for Ray keep working eagerly


def sim_particle(particle):
    print('sim_particle')
    particle = np.ascontiguousarray(np.asarray(particle, dtype=np.float64)[:n_particle])   # noqa: F405
    lk, _, info = _engine(len(particle)).loglik_host(particle)
    if info["n_failed"]:
        raise ValueError(f"RK45 failed for {info['n_failed']} particle(s) (the reference raises here too)")
    return lk, _LazyPredictions(particle)
