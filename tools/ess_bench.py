"""ESS-search pass alone: one fused launch for 16 tempering candidates over 1e6 log-likelihoods (HIP-event time of the
SMC_T_ESS class = ess_partial_kernel + sum_rows_final_kernel) and the max pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
n = 1_000_000
rs = np.random.RandomState(0)
lk = -np.abs(rs.standard_normal(n)) * 300
with pkg.HipEngine(n, 3) as eng:
    eng.upload_lk(pkg.SMC_SET_PRED, lk)
    for k in (1, 4, 8, 16):
        gms = list(0.01 * 0.7 ** np.arange(k))
        eng.ess_partials_global(0.0, gms)
        eng.timing_enable(True); eng.timing_reset()
        for _ in range(50):
            eng.ess_partials_global(0.0, gms)
        tm = eng.timing_get()
        ms = tm["ess"]["ms"] / tm["ess"]["launches"]
        print(f"K = {k:2d} candidates: {ms * 1e3:.1f} us per pass -> {k / (ms * 1e-3):.0f} ESS iterations/s, {8 * n / (ms * 1e-3) / 1e9:.0f} GB/s of lk, "
              f"{k * n / (ms * 1e-3) / 1e9:.0f} G exp/s", flush=True)
    eng.timing_reset()
    for _ in range(50):
        eng.max_lk_global()
    tm = eng.timing_get()
    print(f"max pass: {tm['max']['ms'] / tm['max']['launches'] * 1e3:.1f} us")
