"""Pseudo-data generator of the Michaelis-Menten example on the GPU.

Mirror of SMC_example/Micmen_generate_data.py:31-66 (`make_pseudo_data`): for the i-th initial concentration S0
(i = 1..) seed the global NumPy RNG with 20250205 + i, solve the MM ODE at the true parameters on
linspace(t_span, num_points) - here with the engine's RK45 instead of scipy.solve_ivp - and add N(0, noise_std)
noise to the product curve.  Writes data/<csv_path>_<i>.csv with the reference's columns t,S_true,P_true,P_obs."""
from __future__ import annotations

import os

import numpy as np

from .engine import HipEngine


def simulate_mm(Vmax, Km, S0_list, t_eval, device=0):
    """P(t_eval) = S0 - S(t_eval) for every S0 (one experiment each), solved by the HIP RK45 kernel."""
    t_eval = np.ascontiguousarray(t_eval, dtype=np.float64)
    S0 = np.ascontiguousarray(S0_list, dtype=np.float64)
    with HipEngine(1, 3, device=device) as eng:
        eng.set_model_mm(np.tile(t_eval, (len(S0), 1)), np.zeros((len(S0), len(t_eval))), S0)
        _, pred, info = eng.loglik_host(np.array([[Vmax, Km, 1.0]]), want_pred=True)
    if info["n_failed"]:
        raise RuntimeError("RK45 did not reach the end of t_span")
    return pred[0]


def make_pseudo_data(Vmax_true=1.2, Km_true=0.5, S0_list=(0.1, 0.25, 0.5, 1.0, 2.0), t_span=(0.0, 10.0), num_points=40,
                     noise_std=0.02, csv_path="mm_pseudo_data", out_dir="data", write=True, device=0):
    import pandas as pd
    t = np.linspace(t_span[0], t_span[1], num_points)
    P = simulate_mm(Vmax_true, Km_true, S0_list, t, device=device)
    frames = []
    for i, S0 in enumerate(S0_list, start=1):
        np.random.seed(20250205 + i)                                       # :48-49
        P_true = P[i - 1]
        P_obs = P_true + np.random.normal(0.0, noise_std, size=len(P_true))   # :54
        df = pd.DataFrame({"t": t, "S_true": S0 - P_true, "P_true": P_true, "P_obs": P_obs})
        if write:
            os.makedirs(out_dir, exist_ok=True)
            df.to_csv(os.path.join(out_dir, f"{csv_path}_{i}.csv"), index=False)
        frames.append(df)
    return frames
