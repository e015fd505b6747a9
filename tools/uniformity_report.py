#!/usr/bin/env python3
"""What LLVM's uniformity analysis (the one the AMDGPU backend uses to choose between scalar branches and exec-mask
control flow) says about the loops of one kernel - on the CPU, from `hipcc -emit-llvm -S --cuda-device-only` plus
`opt -passes='print<uniformity>'`:

  * every cycle (loop) that has a DIVERGENT EXIT, i.e. that some lanes may leave while others stay in it,
  * the branches inside it that cause that (divergent terminators with a successor outside the cycle),
  * the cross-lane operations inside it (ballot, readlane, readfirstlane, mbcnt, ds_bpermute, permlane, dpp, wave barrier,
    atomics that the atomic optimiser turns into a wave reduction): a cross-lane operation inside a loop that lanes leave
    one by one reads lanes that are no longer there - the shape of both hangs of profiles/r02_k8_dequeue_hang_isa.md.

Library use: tests/test_k8_uniform_control.py.  Command line:  tools/uniformity_report.py file.hip kernel_substring
"""
import os
import re
import subprocess
import sys
import tempfile

HIPCC = "/opt/rocm/bin/hipcc"
OPT = "/opt/rocm/lib/llvm/bin/opt"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=on", "-fno-fast-math"]   # csrc/Makefile
CROSS_LANE = ("llvm.amdgcn.ballot", "llvm.amdgcn.readlane", "llvm.amdgcn.readfirstlane", "llvm.amdgcn.mbcnt",
              "llvm.amdgcn.ds.bpermute", "llvm.amdgcn.ds.permute", "llvm.amdgcn.permlane", "llvm.amdgcn.update.dpp",
              "llvm.amdgcn.mov.dpp", "llvm.amdgcn.wave.barrier", "llvm.amdgcn.s.barrier", "llvm.amdgcn.icmp",
              "llvm.amdgcn.fcmp", "llvm.amdgcn.wave.reduce", "llvm.amdgcn.set.inactive", "llvm.amdgcn.writelane")


def compile_ir(src, workdir, extra=()):
    ll = os.path.join(workdir, os.path.basename(src) + ".ll")
    subprocess.run([HIPCC, *FLAGS, *extra, "-emit-llvm", "-S", "--cuda-device-only", "-o", ll, src], check=True,
                   stderr=subprocess.DEVNULL, timeout=900)
    r = subprocess.run([OPT, "-mtriple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-passes=print<uniformity>", "-disable-output", ll],
                       check=True, capture_output=True, text=True, timeout=1800)
    return open(ll).read(), r.stderr


def _function_ir(ll_text, kernel_substr):
    """{block label: (instructions, successor labels)} of the kernel's definition in the .ll"""
    m = re.search(r"^define [^\n]*@(\w*" + re.escape(kernel_substr) + r"\w*)\([^\n]*\{\n(.*?)^\}", ll_text, flags=re.M | re.S)
    assert m, f"{kernel_substr}: no definition in the IR"
    blocks, label = {}, None
    first = True
    for line in m.group(2).split("\n"):
        lm = re.match(r"^(\d+):", line)
        if first and not lm:
            label = "entry"
            blocks[label] = ([], [])
        first = False
        if lm:
            label = lm.group(1)
            blocks[label] = ([], [])
            continue
        t = line.strip()
        if not t:
            continue
        blocks[label][0].append(t)
        if t.startswith(("br ", "switch ")):
            blocks[label][1].extend(re.findall(r"label %(\d+)", t))
    return m.group(1), blocks


def kernel_cycles(ll_text, uni_text, kernel_substr):
    """-> list of dicts, one per cycle with a divergent exit: blocks, divergent exiting branches, cross-lane calls."""
    name, blocks = _function_ir(ll_text, kernel_substr)
    parts = re.split(r"^UniformityInfo for function ", uni_text, flags=re.M)
    body = [b for b in parts if b.startswith("'") and b.split("'")[1] == name]
    assert len(body) == 1, f"{name}: not in the uniformity report"
    body = body[0]
    # divergent terminators per block
    div_term = set()
    cur = None
    for line in body.split("\n"):
        bm = re.match(r"^BLOCK (\S+)", line)
        if bm:
            cur = bm.group(1)
        if re.match(r"\s*DIVERGENT:\s+(br|switch) ", line) and cur is not None:
            div_term.add(cur)
    out = []
    for line in body.split("\n"):
        cm = re.match(r"\s*depth=(\d+): entries\(([^)]*)\)(.*)", line)
        if not cm:
            continue
        members = set(cm.group(2).split()) | set(cm.group(3).split())
        exits = []
        for b in sorted(members, key=lambda s: int(s) if s.isdigit() else -1):
            if b not in blocks:
                continue
            succ = blocks[b][1]
            if b in div_term and any(s not in members for s in succ):
                exits.append(b)
        cross = []
        for b in members:
            for ins in blocks.get(b, ([], []))[0]:
                if "atomicrmw" in ins or any(c in ins for c in CROSS_LANE):
                    cross.append((b, ins[:140]))
        out.append({"depth": int(cm.group(1)), "blocks": members, "divergent_exits": exits, "cross_lane": cross})
    n_div_branches = len(div_term)
    return name, out, n_div_branches


def main():
    src, kernel = sys.argv[1], sys.argv[2]
    with tempfile.TemporaryDirectory() as d:
        ll, uni = compile_ir(src, d, sys.argv[3:])
    name, cycles, nb = kernel_cycles(ll, uni, kernel)
    print(f"{name}: {nb} blocks end in a divergent branch; {len(cycles)} cycle(s) with a divergent exit")
    for c in cycles:
        print(f"  depth {c['depth']}, {len(c['blocks'])} blocks, divergent exiting blocks {c['divergent_exits']}, "
              f"{len(c['cross_lane'])} cross-lane operation(s)")
        for b, ins in c["cross_lane"][:12]:
            print(f"      block {b}: {ins}")


if __name__ == "__main__":
    main()
