"""The one-wave-per-solve DAE kernels (K8) must have wave-uniform loop control BY CONSTRUCTION - checked on the CPU with
the compiler's own uniformity analysis, no GPU needed.

Background (profiles/r02_k8_dequeue_hang_isa.md): round 1's first dequeue loop hung on the device because its exits went
through a VGPR; the compiler then ran the two sides of `if (lane == 0)` as separate trips through the loop and the
`__shfl` of the dequeued index executed without lane 0.  The kernels now route everything that steers a loop through
v_readfirstlane (csrc/meth_dae_wave.h: wave_dequeue, wave_uniform).  This test compiles the kernels to LLVM IR and asks
`opt -passes=print<uniformity>` (the analysis the AMDGPU backend itself uses to choose between scalar branches and
exec-mask control flow) for
  * cycles with a divergent exit  -> must be none;
  * divergent branches            -> every one must be explained by the lane id alone (`if (node)`, `if (lane == 0)` ...).
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "python-based-sequential-monte-carlo-method-with-likelihood-tempering_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
OPT = "/opt/rocm/lib/llvm/bin/opt"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=on", "-fno-fast-math"]   # csrc/Makefile


def _uniformity(src, tmp_path):
    ll = str(tmp_path / "k.ll")
    subprocess.run([HIPCC, *FLAGS, "-emit-llvm", "-S", "--cuda-device-only", "-o", ll, os.path.join(CSRC, src)],
                   check=True, stderr=subprocess.DEVNULL, timeout=600)
    r = subprocess.run([OPT, "-mtriple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-passes=print<uniformity>", "-disable-output",
                        ll], check=True, capture_output=True, text=True, timeout=900)
    return r.stderr


def _kernel_report(text, kernel_substr):
    """-> (cycles with divergent exit, divergent branches, those not explained by the lane id) of one kernel."""
    blocks = re.split(r"^UniformityInfo for function ", text, flags=re.M)
    body = [b for b in blocks if b.startswith("'") and kernel_substr in b.split("'")[1]]
    assert len(body) == 1, f"kernel {kernel_substr} not found in the uniformity report"
    lines = body[0].split("\n")
    div, branches = {}, []
    for l in lines:
        m = re.match(r"\s*DIVERGENT:\s+(%\d+) = (.*)", l)
        if m:
            div[m.group(1)] = m.group(2)
        m = re.match(r"\s*DIVERGENT:\s+br i1 (%\d+),", l)
        if m:
            branches.append(m.group(1))
    lane_sources = {k for k, v in div.items() if "workitem.id" in v or "mbcnt" in v or "atomicrmw" in v}

    def roots(v):
        seen, stack, out = set(), [v], set()
        while stack:
            x = stack.pop()
            if x in seen or x not in div:
                continue
            seen.add(x)
            ops = [o for o in re.findall(r"%\d+", div[x]) if o in div and o != x]
            if not ops:
                out.add(x)
            stack += ops
        return out
    unexplained = [b for b in branches if not (roots(b) and roots(b) <= lane_sources)]
    cycles = sum(1 for l in lines if re.match(r"\s*depth=\d+: entries", l))
    in_cycle_section = "CYCLES WITH DIVERGENT EXIT" in body[0]
    return (cycles if in_cycle_section else 0), len(branches), unexplained


@pytest.mark.skipif(not (os.path.exists(HIPCC) and os.path.exists(OPT)), reason="needs hipcc and LLVM opt from ROCm")
@pytest.mark.parametrize("src,kernel", [("meth_smc.hip", "meth_particles_dae_kernel"),
                                        ("meth_kernels.hip", "dae_elem_kernel")])
def test_k8_loop_control_is_wave_uniform(src, kernel, tmp_path):
    text = _uniformity(src, tmp_path)
    cycles, n_branches, unexplained = _kernel_report(text, kernel)
    assert cycles == 0, f"{kernel}: {cycles} loop(s) with a divergent exit"
    assert unexplained == [], f"{kernel}: divergent branches that do not come from the lane id: {unexplained[:5]}"
    assert n_branches > 0        # the lane-id branches (if (node) ...) are still there: the parser saw the kernel
    shutil.rmtree(tmp_path, ignore_errors=True)


def test_no_dequeue_behind_a_lane_id_branch():
    """The two hangs of this code base (profiles/r02_k8_dequeue_hang_isa.md) had one source shape: a value produced under
    `if (lane == 0)` (an atomicAdd on a work queue) and then read by every lane through v_readfirstlane / __shfl.  The
    compiler may separate the two sides of such a branch.  No source file may contain that shape again."""
    csrc = CSRC
    bad = []
    for name in sorted(os.listdir(csrc)):
        if not name.endswith((".hip", ".h")):
            continue
        text = open(os.path.join(csrc, name), encoding="utf-8").read()
        text = re.sub(r"//[^\n]*", "", text)
        for m in re.finditer(r"if\s*\(\s*(lane|threadIdx\.x\s*&\s*63|l)\s*==\s*0\s*\)\s*\{?[^;{}]*=\s*atomicAdd", text):
            tail = text[m.end():m.end() + 400]
            if re.search(r"readfirstlane|__shfl\s*\(", tail):
                bad.append(f"{name}: ...{text[m.start():m.end()]}...")
    assert not bad, bad
