"""Instruction census of the uniform attempt loop of the run-time compiled user-model kernel, off line (no GPU): dumps what
hiprtc would get (smc_user_model_dump_source), compiles it with hipcc -S and counts the loop that carries the
uniform_tail_attempt mark.   python tools/user_isa.py [MICHAELIS_MENTEN | MICHAELIS_MENTEN_PLAIN | CONSECUTIVE_REACTIONS]"""
import os, re, subprocess, sys
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "MICHAELIS_MENTEN"
ns = 2 if name == "CONSECUTIVE_REACTIONS" else 1
d = os.path.join(g.ROOT, "build", "user_dump")
os.makedirs(d, exist_ok=True)
assert pkg.lib().smc_user_model_dump_source(getattr(pkg.user_models, name).encode(), ns, 3, d.encode()) == 0
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=on", "-fno-fast-math", "-I", d, "-DSMC_ISA_MARKS", "-S",
                    "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", d + "/u.s", d + "/smc_user_model.hip"], capture_output=True, text=True)
assert r.returncode == 0, r.stderr[-3000:]
for l in r.stderr.split("\n"):
    if re.search(r"Function Name|VGPRs:|SGPRs Spill|VGPRs Spill|ScratchSize|Occupancy", l):
        print(l.split("remark:")[1].strip())
lines = open(d + "/u.s").read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("smc_user_solve_kernel:"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = [l.strip() for l in lines[start:end]]
for mark in ("uniform_tail_attempt", "bulk_attempt"):
    for m in [i for i, l in enumerate(body) if "MARK " + mark in l]:
        lab = next(i for i in range(m, -1, -1) if re.match(r"^\.LBB\d+_\d+:", body[i]))
        label = body[lab].split(":")[0]
        back = [i for i in range(m, len(body)) if re.search(r"s_c?branch\w*\s+" + re.escape(label) + r"\b", body[i])]
        if not back:
            print(mark, label, "no back edge"); continue
        loop = [l for l in body[lab:back[-1] + 1] if l and not l.startswith(";") and not l.startswith(".")]
        c = Counter(l.split()[0] for l in loop)
        print(f"{mark} {label}: {len(loop)} instructions, valu {sum(v for k, v in c.items() if k.startswith('v_'))}, salu "
              f"{sum(v for k, v in c.items() if k.startswith('s_'))}, s_mov_b32 {c['s_mov_b32']}, saveexec {sum(v for k, v in c.items() if 'saveexec' in k)}, "
              f"div_scale {c['v_div_scale_f64']}, rcp {c['v_rcp_f64_e32']}")
