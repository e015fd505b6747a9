#!/usr/bin/env python3
"""Basic-block census of one kernel in a `hipcc -S` listing: per block the number of vector, scalar, LDS / memory
instructions, SGPR-spill traffic (v_writelane / v_readlane) and the loop annotations LLVM prints next to the label.
Used on the CPU to see what an inner loop of mm_solve_kernel really issues (scalar instructions per
vector instruction, SGPR spills) without a GPU:

    hipcc -O3 ... -S --cuda-device-only mm_kernels.hip -o mm.s ;  tools/isa_blocks.py mm.s mm_solve_kernelILb0 [min_instrs]
"""
import re
import sys


def census(path, kernel_substr, min_instrs=0):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(kernel_substr) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    # the function body may hold more than one s_endpgm: go on to .Lfunc_end
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], {"label": "entry", "note": "", "ins": []}
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "note": (m.group(2) or "").strip("; ").strip(), "ins": []}
            continue
        t = l.strip()
        if t.startswith("; MARK "):
            cur["note"] = "<<" + t[7:] + ">> " + cur["note"]
            continue
        if not t or t.startswith(";") or t.startswith("."):
            if t.startswith(";") and cur["ins"] == [] and ("Loop" in t or "Header" in t):
                cur["note"] += " | " + t.strip("; ").strip()
            continue
        cur["ins"].append(t.split(";")[0].strip())
    blocks.append(cur)
    tot = {"v": 0, "s": 0, "lane": 0}
    print(f"{'block':>12} {'all':>5} {'valu':>5} {'salu':>5} {'smov':>5} {'lane':>5} {'f64':>4} {'lds':>4} {'mem':>4} {'br':>3}  note")
    for b in blocks:
        ins = b["ins"]
        v = sum(1 for x in ins if x.startswith("v_"))
        s = sum(1 for x in ins if x.startswith("s_") and not x.startswith(("s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_endpgm", "s_barrier")))
        smov = sum(1 for x in ins if x.startswith(("s_mov_b32", "s_mov_b64")))
        lane = sum(1 for x in ins if x.startswith(("v_writelane", "v_readlane")))
        f64 = sum(1 for x in ins if re.match(r"v_\w+_f64", x))
        lds = sum(1 for x in ins if x.startswith("ds_"))
        mem = sum(1 for x in ins if x.startswith(("global_", "flat_", "buffer_", "scratch_", "s_load", "s_buffer")))
        br = sum(1 for x in ins if x.startswith(("s_cbranch", "s_branch")))
        tot["v"] += v
        tot["s"] += s
        tot["lane"] += lane
        if len(ins) >= min_instrs:
            print(f"{b['label']:>12} {len(ins):5d} {v:5d} {s:5d} {smov:5d} {lane:5d} {f64:4d} {lds:4d} {mem:4d} {br:3d}  {b['note'][:90]}")
    print("total", tot)
    return blocks


if __name__ == "__main__":
    census(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 0)
