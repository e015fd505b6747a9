""""the reference's drivers drop in unchanged".  BUILD-CONTAINER ONLY (skipped where /root/reference is
absent, i.e. on the GPU box): the reference's own driver script, SMC_example/Micmem_SMC_main.py, is executed UNMODIFIED
with runpy, but with the shadow modules of this package (dropin/Micmem_settings.py, dropin/Micmem_likelihood.py) first
on sys.path, so that its `from Micmem_settings import *` / `from Micmem_likelihood import *` (:28-29) resolve to them.

There is no GPU in the build container, so the engine behind the shadow `sim_particle` is swapped for a CPU double that
answers `loglik_host` with the oracle's C restatement - the point of this test is the BOUNDARY (names, import-time side
effects, call signatures, return types the driver's NumPy expressions rely on: `lk - max_lk` :118, `lk2 - lk1` :231,
`lk2 * r` :240, the RNG stream), not the kernels (those are pinned on the GPU by tests/test_gpu_parity.py and
tests/test_dropin.py).  The packages the reference imports but this image lacks (ray, numba, assimulo, memory_profiler,
seaborn) are the inert stand-ins of tests/golden/_shims, as for the golden-vector generator.  Nothing of the reference is
copied or shipped; its files are only read from /root/reference at test time.

Pass criterion: the run finishes, and its tempering schedule (parsed from the driver's own log line, :254), accept counts
and final particles equal those of the all-reference run recorded in tests/golden/mm_ref_run_n1000.npz.
"""
import contextlib
import io
import os
import re
import runpy
import sys

import numpy as np
import pytest

import __graft_entry__ as g

REF = "/root/reference/SMC_example"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference is only present in the build container")
SHIMS = os.path.join(g.ROOT, "tests", "golden", "_shims")
MODS = ("Micmem_settings", "Micmem_likelihood", "ray", "numba", "assimulo", "assimulo.problem", "assimulo.solvers",
        "memory_profiler", "seaborn")


class _CpuEngine:
    """Test double for smc_lt_amd.engine.HipEngine, as far as dropin/Micmem_likelihood.py uses it."""
    O = None

    def __init__(self, n_local, dim=3, device=0, n_global=None):
        self.n_local, self.dim = int(n_local), dim
        self.data = None

    def set_model_mm(self, t, P_obs, S0, est_sigma=True, sigma_fixed=5.0, rtol=1e-3, atol=1e-6):
        assert est_sigma and rtol == 1e-3 and atol == 1e-6
        self.data = self.O.MMData(np.ascontiguousarray(t, dtype=float), np.ascontiguousarray(P_obs, dtype=float),
                                  np.ascontiguousarray(S0, dtype=float))

    def loglik_host(self, particle, want_pred=False):
        lk, pred, info = self.O.mm_loglik_batch(np.ascontiguousarray(particle, dtype=float), self.data, want_pred=want_pred)
        return lk, pred, {"n_failed": info["n_failed"], "rk_attempts": info["n_attempts"]}

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


@pytest.fixture()
def shadow_env(tmp_path, monkeypatch):
    os.symlink(os.path.join(REF, "data"), tmp_path / "data")     # the driver reads data/ and writes a PNG into the cwd
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("MPLBACKEND", "Agg")
    pkg = g.load_package()
    O = g.load_oracle()
    _CpuEngine.O = O
    import smc_lt_amd.engine as engine_mod
    monkeypatch.setattr(engine_mod, "HipEngine", _CpuEngine)
    saved = {m: sys.modules.pop(m, None) for m in MODS}
    monkeypatch.syspath_prepend(SHIMS)
    monkeypatch.syspath_prepend(os.path.join(g.PKG_DIR, "dropin"))     # first: the shadows win over everything else
    monkeypatch.setattr(sys, "dont_write_bytecode", True)
    yield pkg
    for m in MODS:
        sys.modules.pop(m, None)
        if saved[m] is not None:
            sys.modules[m] = saved[m]


def test_unmodified_reference_driver_runs_on_the_shadow_modules(shadow_env, golden_run):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = runpy.run_path(os.path.join(REF, "Micmem_SMC_main.py"), run_name="__main__")
    log = buf.getvalue()
    # the star-imports resolved to this package's shadows, not to the reference's modules
    assert sys.modules["Micmem_likelihood"].__file__.startswith(os.path.join(g.PKG_DIR, "dropin"))
    assert sys.modules["Micmem_settings"].__file__.startswith(os.path.join(g.PKG_DIR, "dropin"))
    assert "Traceback" not in log                                   # the driver swallows exceptions and prints them (:305-314)
    # schedule from the driver's own log line (:254)
    rows = re.findall(r"iteration:(\d+), nMH:(\d+), Calculation time:[^,]+, ESS:([^,]+), Max Likelihood:([^,]+), "
                      r"New Gamma:([^,]+), Number of Adoption:([^\s]+)", log)
    assert len(rows) == len(golden_run["sched_gamma"]) == 14
    gam = np.array([float(r[4]) for r in rows])
    assert np.array_equal(gam, golden_run["sched_gamma"]) and gam[-1] == 1.0
    assert np.array_equal([float(r[5]) for r in rows], golden_run["sched_accept"])
    assert np.array_equal([int(r[1]) for r in rows], golden_run["sched_last_j"])
    assert np.allclose([float(r[2]) for r in rows], golden_run["sched_ess"], rtol=1e-9, atol=0)
    assert log.count("sim_particle") == 34                          # 1 initial + 33 Metropolis sweeps, like the reference run
    # final state the script leaves in its globals
    assert out["p_pred"].shape == (1000, 3) and np.abs(out["p_pred"] - golden_run["final_p_pred"]).max() < 1e-9
    ref_lk = golden_run["final_lk"]
    assert np.max(np.abs(np.asarray(out["lk"]) - ref_lk) / np.maximum(1, np.abs(ref_lk))) < 1e-9
    assert os.path.exists("Posterior_Distributions.png")             # the driver's only output file (:277-295)
    # the global NumPy stream ends where the all-reference run's did
    assert np.random.rand() == float(golden_run["next_rand_after_run"])


def test_shadow_settings_cover_every_name_of_the_reference_settings_module(shadow_env):
    """The names the drop-in must keep ("Settings names") taken from the reference module itself rather
    than from a hand-made list: every public global of SMC_example/Micmem_settings.py exists in the shadow, with an equal
    value where the value is plain data."""
    ref = runpy.run_path(os.path.join(REF, "Micmem_settings.py"))
    sys.modules.pop("Micmem_settings", None)
    import importlib
    S = importlib.import_module("Micmem_settings")
    assert S.__file__.startswith(os.path.join(g.PKG_DIR, "dropin"))
    names = [k for k, v in ref.items() if not k.startswith("_") and not isinstance(v, type(os))]
    assert len(names) >= 40
    missing = [k for k in names if not hasattr(S, k)]
    assert not missing, missing
    checked = 0
    for k in names:
        a, b = ref[k], getattr(S, k)
        if isinstance(a, (int, float, bool, str)):
            assert a == b, k
            checked += 1
        elif isinstance(a, np.ndarray) and a.dtype.kind in "fi":
            assert a.shape == np.asarray(b).shape and np.array_equal(a, b), k     # incl. the prior sample p_pred (same seed)
            checked += 1
        elif isinstance(a, dict) and k == "priors":
            assert a == b
            checked += 1
    assert checked >= 30
