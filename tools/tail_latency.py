"""One stiff (particle, experiment) solve through the PRODUCT kernel, alone on the GPU - the latency-bound tail of an
early-tempering sweep - next to tools/attempt_probe.hip's stand-alone loop on the same operands (Vmax 10, Km 3e-3, S0 0.1:
12 562 RK45 attempts).  `python tools/tail_latency.py user` runs the same solve through the run-time compiled user-model
kernel (with its cost hint: a solo solve; `user-plain`: without, through the scheduler's uniform tail)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
t = np.linspace(0, 10, 40)[None, :]
for n_lanes in (1, 64, 4096):
    with pkg.HipEngine(max(n_lanes, 1), 3) as eng:
        mode = sys.argv[1] if len(sys.argv) > 1 else "built-in"
        th = np.tile(np.array([[10.0, 3e-3, 1.0]]), (n_lanes, 1))
        if mode.startswith("built-in"):
            eng.set_model_mm(t, np.zeros((1, 40)), np.array([0.1]))
            eng.set_fast_tail(mode != "built-in-nofast")          # the hand-written lone-chain loop (smc_set_fast_tail)
            run = lambda: eng.loglik_host(th)[2]                      # noqa: E731
        else:
            src = pkg.user_models.MICHAELIS_MENTEN if mode == "user" else pkg.user_models.MICHAELIS_MENTEN_PLAIN
            eng.set_prior(pkg.SMCSettings().priors)
            eng.set_model_user(src, 1, t, np.zeros((1, 40)), cond=np.array([[0.1]]))
            eng.upload_particles(pkg.SMC_SET_PRED, th)
            run = lambda: eng.loglik(pkg.SMC_SET_PRED)                # noqa: E731
        run()
        eng.timing_enable(True); eng.timing_reset()
        reps = 10
        for _ in range(reps):
            info = run()
        tm = eng.timing_get()
        ms = tm["solve"]["ms"] / tm["solve"]["launches"]
        att = info["rk_attempts"] / n_lanes
        print(f"{n_lanes:5d} identical stiff items: solve kernel {ms:.3f} ms, {att:.0f} attempts each -> {ms * 1e3 / att:.4f} us per attempt", flush=True)
