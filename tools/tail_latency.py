"""One stiff (particle, experiment) solve through the PRODUCT kernel, alone on the GPU - the latency-bound tail of an
early-tempering sweep - next to tools/attempt_probe.hip's stand-alone loop on the same operands (Vmax 10, Km 3e-3, S0 0.1:
12 562 RK45 attempts)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
t = np.linspace(0, 10, 40)[None, :]
for n_lanes in (1, 64, 4096):
    with pkg.HipEngine(max(n_lanes, 1), 3) as eng:
        eng.set_model_mm(t, np.zeros((1, 40)), np.array([0.1]))
        th = np.tile(np.array([[10.0, 3e-3, 1.0]]), (n_lanes, 1))
        eng.loglik_host(th)
        eng.timing_enable(True); eng.timing_reset()
        reps = 10
        for _ in range(reps):
            lk, _, info = eng.loglik_host(th)
        tm = eng.timing_get()
        ms = tm["solve"]["ms"] / tm["solve"]["launches"]
        att = info["rk_attempts"] / n_lanes
        print(f"{n_lanes:5d} identical stiff items: solve kernel {ms:.3f} ms, {att:.0f} attempts each -> {ms * 1e3 / att:.4f} us per attempt", flush=True)
