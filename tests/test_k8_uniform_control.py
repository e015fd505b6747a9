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


_REPORTS = {}      # source file -> uniformity report (one compile + one analysis per file, shared by its kernels)


def _uniformity(src, tmp_path):
    if src in _REPORTS:
        return _REPORTS[src]
    ll = str(tmp_path / "k.ll")
    subprocess.run([HIPCC, *FLAGS, "-emit-llvm", "-S", "--cuda-device-only", "-o", ll, os.path.join(CSRC, src)],
                   check=True, stderr=subprocess.DEVNULL, timeout=600)
    r = subprocess.run([OPT, "-mtriple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-passes=print<uniformity>", "-disable-output",
                        ll], check=True, capture_output=True, text=True, timeout=900)
    _REPORTS[src] = r.stderr
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
                                        ("meth_kernels.hip", "dae_elem_kernel"),
                                        # two waves per solve (meth_dae_split.h): the s_barriers of the command protocol sit in this
                                        # control flow, so a divergent branch around one of them would be a hang
                                        ("meth_smc.hip", "meth_particles_dae_split_kernel"),
                                        ("meth_kernels.hip", "dae_split_kernel")])
def test_k8_loop_control_is_wave_uniform(src, kernel, tmp_path):
    text = _uniformity(src, tmp_path)
    cycles, n_branches, unexplained = _kernel_report(text, kernel)
    assert cycles == 0, f"{kernel}: {cycles} loop(s) with a divergent exit"
    assert unexplained == [], f"{kernel}: divergent branches that do not come from the lane id: {unexplained[:5]}"
    assert n_branches > 0        # the lane-id branches (if (node) ...) are still there: the parser saw the kernel
    shutil.rmtree(tmp_path, ignore_errors=True)


# <WRITE_PRED, EXACT, FAST> as mm_kernels.hip launches them: FAST (the hand-written lone-chain loop) only in the default mode
MM_SOLVE_INSTANTIATIONS = ("mm_solve_kernelILb0ELb0ELb0", "mm_solve_kernelILb0ELb0ELb1", "mm_solve_kernelILb0ELb1ELb0",
                           "mm_solve_kernelILb1ELb0ELb0", "mm_solve_kernelILb1ELb0ELb1", "mm_solve_kernelILb1ELb1ELb0")


def _load_tool(name):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.skipif(not (os.path.exists(HIPCC) and os.path.exists(OPT)), reason="needs hipcc and LLVM opt from ROCm")
def test_mm_solve_kernel_has_no_cross_lane_operation_in_a_loop_that_lanes_leave_one_by_one(tmp_path):
    """VERDICT r2 item 4(a).  mm_solve_kernel is the most intricate scheduler of the library (item pool, ballots, the
    uniform tail with its v_readlane broadcast).  Its lanes DO diverge by design - every lane runs its own solve - so the
    K8 criterion (no divergent branch at all) does not apply.  What must hold is narrower and is what the two hangs of
    this code base violated: no cycle that the compiler's uniformity analysis reports as having a DIVERGENT EXIT (lanes
    leave it one by one) may contain a cross-lane operation (ballot, readlane, readfirstlane, mbcnt, ds_bpermute, DPP,
    a wave barrier) or an atomic that the atomic optimiser turns into a wave reduction.  The legitimate divergent
    cycles are the per-lane dense-output loops, the strided LDS table fill and the `blk -= n_blk` chain; the hand-out
    loop, the pool, the tail loop and the attempt loops must not appear.  Round 2's kernel failed this: the join of its
    `if (lane == src)` publish was the tail loop's exit block, so the analysis (conservatively) called the whole tail loop
    divergent and compiled the 'uniform' attempt loop with exec-mask control flow (tools/uniformity_report.py)."""
    U = _load_tool("uniformity_report")
    ll, uni = U.compile_ir(os.path.join(CSRC, "mm_kernels.hip"), str(tmp_path))
    for inst in MM_SOLVE_INSTANTIATIONS:
        name, cycles, n_div = U.kernel_cycles(ll, uni, inst)
        assert n_div > 0                                   # the per-lane branches are there: the parser saw the kernel
        bad = [(c["depth"], len(c["blocks"]), c["cross_lane"][:3]) for c in cycles if c["cross_lane"]]
        assert not bad, f"{name}: cross-lane operations inside a cycle with a divergent exit: {bad}"
        assert all(len(c["blocks"]) <= 8 for c in cycles), f"{name}: a large cycle has a divergent exit: " \
            f"{[(c['depth'], len(c['blocks'])) for c in cycles]}"
    shutil.rmtree(tmp_path, ignore_errors=True)


def _innermost_loop(body, mk):
    """(index of the loop's first label, index of the last branch back to it) for the innermost loop around line mk"""
    for lab in range(mk, -1, -1):
        if not re.match(r"^\.LBB\d+_\d+:", body[lab]):
            continue
        label = body[lab].split(":")[0]
        back = [i for i in range(mk, len(body)) if re.search(r"s_c?branch\w*\s+" + re.escape(label) + r"\b", body[i])]
        if back:
            return lab, back[-1]
    raise AssertionError("no loop around the mark")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_mm_chunk_dequeue_in_the_isa(tmp_path):
    """VERDICT r2 item 4(b) / ADVICE r2: the chunk dequeue of mm_solve_kernel.  Source form: the first lane adds kChunk,
    the others add 0 (correct with or without LLVM's atomic optimiser).  In the ISA of both instantiations there must be
    exactly ONE memory atomic on the queue - a 64-bit add that returns its old value - and the uniform tail's attempt
    loop must branch on scalar conditions only (s_cbranch_vcc*, no s_*_saveexec): that is what 'wave-uniform operands'
    buys, and what the early-rejection check silently took away in round 2."""
    asm = str(tmp_path / "mm.s")
    subprocess.run([HIPCC, *FLAGS, "-DSMC_ISA_MARKS", "-S", "--cuda-device-only", "-o", asm, os.path.join(CSRC, "mm_kernels.hip")],
                   check=True, stderr=subprocess.DEVNULL, timeout=600)
    lines = open(asm).read().split("\n")
    for inst in MM_SOLVE_INSTANTIATIONS:
        start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN3smc15" + inst + r"\w*:", l))
        end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
        body = [l.strip() for l in lines[start:end]]
        atomics = [l for l in body if re.match(r"(global|flat|buffer)_atomic", l)]
        assert len(atomics) == 1, (inst, atomics)
        assert re.match(r"global_atomic_add_x2 v\[\d+:\d+\], ", atomics[0]) and ("sc0" in atomics[0] or "glc" in atomics[0]), atomics[0]
        # the uniform attempt loops (compiled step function; hand-written block + step function): the innermost loop around
        # each mark, i.e. from the nearest label before the mark that a later instruction branches back to, to that branch
        marks = [i for i, l in enumerate(body) if "MARK uniform_tail_attempt" in l]
        assert marks
        for mk in marks:
            lab, back = _innermost_loop(body, mk)
            loop = [l for l in body[lab:back + 1] if l and not l.startswith(";")]
            assert not [l for l in loop if re.match(r"s_\w+_saveexec", l)], f"{inst}: exec-mask control flow in the uniform attempt loop"
            assert sum(1 for l in loop if l.startswith("s_cbranch_vcc")) >= 4
        # the hand-written block is there (FAST instantiations only) and is a loop of scalar branches
        fast = [i for i, l in enumerate(body) if re.match(r"^\.Lfast_loop_\d+:", l)]
        assert len(fast) == (2 if inst.endswith("ELb1") else 0), (inst, fast)     # FAST: solo phase + uniform tail
        for f0 in fast:
            f1 = next(i for i in range(f0, len(body)) if re.match(r"^\.Lfast_end_\d+:", body[i]))
            blk = [l for l in body[f0:f1] if l and not l.startswith(";") and not l.startswith(".")]
            assert 130 <= len(blk) <= 160 and not [l for l in blk if "saveexec" in l or re.match(r"s_mov_b32 s\d+, 0x", l)], len(blk)   # no constant is re-materialised


@pytest.mark.skipif(not (os.path.exists(HIPCC) and os.path.exists(OPT)), reason="needs hipcc and LLVM opt from ROCm")
@pytest.mark.parametrize("model,n_states", [("MICHAELIS_MENTEN", 1), ("CONSECUTIVE_REACTIONS", 2)])
def test_user_model_kernel_is_under_the_same_control_flow_checks(pkg, tmp_path, model, n_states):
    """VERDICT r2 item 4(a), second half: the run-time compiled user-model kernel runs on the scheduler of the built-in kernel
    (csrc/solve_sched.h, handed to hiprtc as an in-memory header), so the same must hold for it.  The library writes out
    exactly what hiprtc gets (smc_user_model_dump_source); hipcc compiles that off line and the compiler's uniformity
    analysis is asked the same question: no cross-lane operation inside a cycle that lanes leave one by one.  On the ISA:
    one memory atomic on the queue in the solve kernel (the chunk dequeue), plain per-lane atomics only in the scan kernel
    that builds the lists of a model with a cost hint, scalar branches only in the uniform attempt loop."""
    d = str(tmp_path)
    src = getattr(pkg.user_models, model)
    assert pkg.lib().smc_user_model_dump_source(src.encode(), n_states, 3, d.encode()) == 0
    assert sorted(os.listdir(d)) == ["philox.h", "rk45_math.h", "smc_user_model.hip", "solve_sched.h", "sweep_args.h"]
    U = _load_tool("uniformity_report")
    ll, uni = U.compile_ir(os.path.join(d, "smc_user_model.hip"), d, extra=("-I", d))
    name, cycles, n_div = U.kernel_cycles(ll, uni, "smc_user_solve_kernel")
    assert n_div > 0
    bad = [(c["depth"], len(c["blocks"]), c["cross_lane"][:3]) for c in cycles if c["cross_lane"]]
    assert not bad, f"{name}: cross-lane operations inside a cycle with a divergent exit: {bad}"
    assert all(len(c["blocks"]) <= 8 for c in cycles), f"{name}: a large cycle has a divergent exit: " \
        f"{[(c['depth'], len(c['blocks'])) for c in cycles]}"
    asm = os.path.join(d, "u.s")
    subprocess.run([HIPCC, *FLAGS, "-I", d, "-DSMC_ISA_MARKS", "-S", "--cuda-device-only", "-o", asm, os.path.join(d, "smc_user_model.hip")],
                   check=True, stderr=subprocess.DEVNULL, timeout=600)
    lines = open(asm).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("smc_user_solve_kernel:"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = [l.strip() for l in lines[start:end]]
    atomics = [l for l in body if re.match(r"(global|flat|buffer)_atomic", l)]
    assert len(atomics) == 1 and re.match(r"global_atomic_add_x2 v\[\d+:\d+\], ", atomics[0]), atomics
    for mk in [i for i, l in enumerate(body) if "MARK uniform_tail_attempt" in l]:
        lab, back = _innermost_loop(body, mk)
        loop = [l for l in body[lab:back + 1] if l and not l.startswith(";")]
        assert not [l for l in loop if re.match(r"s_\w+_saveexec", l)], f"{model}: exec-mask control flow in the uniform attempt loop"
    shutil.rmtree(tmp_path, ignore_errors=True)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc from ROCm")
def test_two_wave_k8_kernel_keeps_its_loops_free_of_scratch_stores(tmp_path):
    """The two-wave K8 kernel (csrc/meth_dae_split.h) is compiled for 256 VGPRs - two waves per SIMD - and lives at that limit.
    A value the register allocator parks in scratch INSIDE the step loop is stored once per factorisation or per step by every
    lane: an earlier build wrote 7 GB per launch that way, against 45 MB of algorithmic traffic (profiles/r04_ab_k8_split.log).
    Checked on the CPU from the listing: scratch stores may only sit outside every loop (the kernel's entry), and the scratch
    loads inside loops stay few (reloads of hoisted constants)."""
    asm = str(tmp_path / "k.s")
    subprocess.run([HIPCC, *FLAGS, "-S", "--cuda-device-only", "-o", asm, os.path.join(CSRC, "meth_smc.hip")], check=True,
                   stderr=subprocess.DEVNULL, timeout=900)
    lines = open(asm).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN3smc31meth_particles_dae_split_kernel\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    in_loop, stores_in_loops, loads_in_loops, stores_outside = False, 0, 0, 0
    for l in lines[start + 1:end]:
        m = re.match(r"^\.LBB\d+_\d+:\s*(;.*)?$", l)
        if m:
            in_loop = "Loop" in (m.group(1) or "")
            continue
        t = l.strip()
        if t.startswith("scratch_store"):
            stores_in_loops += in_loop
            stores_outside += not in_loop
        elif t.startswith("scratch_load"):
            loads_in_loops += in_loop
    meta = "\n".join(lines)
    k = meta.index(".name:           _ZN3smc31meth_particles_dae_split_kernel")
    vgprs = int(re.search(r"\.vgpr_count:\s+(\d+)", meta[k:]).group(1))
    assert vgprs <= 256, vgprs                      # two waves per SIMD
    assert stores_in_loops == 0, f"{stores_in_loops} scratch stores inside loops"
    assert loads_in_loops <= 48, loads_in_loops     # static count; 25 at the time of writing
    shutil.rmtree(tmp_path, ignore_errors=True)
