"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/smc_hip.h declares; the product fails loudly without a GPU / without the library."""
import ctypes
import os
import subprocess

import pytest


def _have_gpu():
    return os.path.exists("/dev/kfd")


def test_library_exports_every_declared_symbol(pkg):
    L = pkg.lib()
    declared = pkg.header_symbols()
    assert len(declared) >= 40
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert set(declared) == set(pkg.binding.SIGNATURES), set(declared) ^ set(pkg.binding.SIGNATURES)
    assert L.smc_abi_version() == pkg.binding.SMC_ABI_VERSION == 3      # 3: smc_work_totals, batched Metropolis loop for every model (round 5)
    hdr = open(pkg.binding.HEADER_PATH).read()
    assert "#define SMC_ABI_VERSION 3" in hdr


def test_exported_symbols_are_plain_c(pkg):
    out = subprocess.check_output(["nm", "-D", "--defined-only", pkg.LIB_PATH], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(pkg.header_symbols()) <= exported
    # no torch / python in the link line
    ldd = subprocess.check_output(["ldd", pkg.LIB_PATH], text=True)
    assert "torch" not in ldd and "python" not in ldd
    assert "libamdhip64" in ldd and "librccl" in ldd


def test_no_silent_cpu_fallback(pkg):
    if _have_gpu():
        pytest.skip("GPU present")
    with pytest.raises(pkg.SmcError):
        pkg.HipEngine(16, 3)


def test_product_does_not_import_oracle():
    import __graft_entry__ as g
    for root, _, files in os.walk(g.PKG_DIR):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(root, f)).read()
                assert "oracle" not in txt.replace("test oracle", ""), f"{f} mentions the oracle"


def test_ess_candidates_are_the_reference_recurrence(pkg):
    s = pkg.SMCSettings()
    gms, gammas, after = pkg.ess_candidates(0.25, s)
    assert len(gms) == 80 and gammas[0] == 1.0 and gms[0] == 0.75
    g = 1.0
    for k in range(80):
        assert gammas[k] == g and gms[k] == g - 0.25
        g = (g - 0.25) * 0.7 + 0.25
    assert after == g
