#!/usr/bin/env python3
"""Condense rocprofv3 --pmc passes of SQ counters into the per-kernel instruction-issue summary bench.py reads
(profiles/rNN_pmc_sq_summary.json), stamped with the kernel revision.  The passes (four counters each, one run per pass,
--kernel-trace only, as MI355X_MICROARCH.md prescribes):

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/sq1 -- python3 $R/tools/steady_state.py 1000000 1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU ...                          -d .../sq2
  rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 ...         -d .../sq3
  rocprofv3 --pmc SQ_WAVES SQ_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS ...                                                   -d .../sq4
  python3 tools/pmc_sq_summary.py profiles/r02_pmc_sq_summary.json --kernel mm_solve_kernel --last 10 \
          --command "python3 tools/steady_state.py 1000000 1" gpurun_out/sq1 gpurun_out/sq2 gpurun_out/sq3 gpurun_out/sq4

Units (checked against each other on this kernel: SQ_WAVE_CYCLES x 4 = waves x kernel cycles): the SQ *_CYCLES / ACTIVE_* /
WAIT_* counters tick once per 4 clocks; SQ_CYCLES and SQ_BUSY_CYCLES are summed over the 32 shader engines,
GRBM_GUI_ACTIVE over the 8 XCDs.
"""
import argparse
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N_SIMD = 256 * 4
N_XCD = 8


def collect(directory, kernel, last):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {directory}")
    files = [max(files, key=os.path.getmtime)]        # one run per directory: the newest (gpurun merges, it never deletes)
    rows = [r for f in files for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    if last:
        ids = ids[-last:]
    keep = set(ids)
    tot, dur, seen = {}, 0.0, set()
    for r in rows:
        d = int(r["Dispatch_Id"])
        if d not in keep:
            continue
        tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if d not in seen:
            seen.add(d)
            dur += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    n = len(ids)
    return {k: v / n for k, v in tot.items()}, dur / n, n, rows[0]["VGPR_Count"], rows[0]["Grid_Size"], rows[0]["Workgroup_Size"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--kernel", default="mm_solve_kernel")
    ap.add_argument("--last", type=int, default=0, help="only the last N dispatches of the kernel (0 = all)")
    ap.add_argument("--command", default="")
    ap.add_argument("--family", choices=["mm", "k8"], default="mm", help="which sources the revision hash covers (bench.kernel_source_sha)")
    a = ap.parse_args()
    import bench
    c, durs = {}, []
    for d in a.dirs:
        vals, dur, n, vgpr, grid, wg = collect(d, a.kernel, a.last)
        c.update(vals)
        durs.append(dur)
    cyc = c["GRBM_GUI_ACTIVE"] / N_XCD                      # kernel duration in shader clocks
    valu_busy = 4.0 * c["SQ_ACTIVE_INST_VALU"] / (N_SIMD * cyc)
    f64 = c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_TRANS_F64"]
    out = {"meta": {"kernel_source_sha": bench.kernel_source_sha(family=a.family), "kernel": a.kernel, "dispatches_averaged": n,
                    "command": a.command, "vgpr_count": int(vgpr), "grid": int(grid), "workgroup": int(wg),
                    "avg_dispatch_ns": sum(durs) / len(durs)},
           "counters_per_dispatch": c,
           "derived": {
               "shader_clocks_per_dispatch": cyc,
               "shader_clock_GHz": cyc / (sum(durs) / len(durs)),
               "valu_busy_fraction": valu_busy,
               "valu_busy_note": "4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x clocks): share of the SIMD-cycles in which a vector instruction executes",
               "valu_issue_floor_fraction": 4.0 * c["SQ_INSTS_VALU"] / (N_SIMD * cyc),
               "valu_issue_floor_note": "SQ_INSTS_VALU x 4 clocks (the shortest a wave64 instruction occupies its SIMD) / SIMD-cycles",
               "lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]),
               "fp64_share_of_valu_instructions": f64 / c["SQ_INSTS_VALU"],
               "fma_share_of_fp64_instructions": c["SQ_INSTS_VALU_FMA_F64"] / f64,
               "salu_per_valu_instruction": c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"],
               "waves_resident_fraction": 4.0 * c["SQ_WAVE_CYCLES"] / (c["SQ_WAVES"] * cyc),
               "wave_wait_fraction": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
               "lds_per_valu_instruction": c["SQ_INSTS_LDS"] / c["SQ_INSTS_VALU"] if "SQ_INSTS_LDS" in c else None,
               "valu_instructions_per_dispatch": c["SQ_INSTS_VALU"],
           }}
    # optional fifth pass (K8): LDS pipeline and wait reasons
    for k_out, k_in in (("lds_bank_conflict_share_of_lds_active", ("SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS")),
                        ("lds_active_fraction_of_wave_cycles", ("SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES")),
                        ("wait_lgkm_fraction_of_wave_cycles", ("SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES")),
                        ("scalar_active_fraction_of_wave_cycles", ("SQ_ACTIVE_INST_SCA", "SQ_WAVE_CYCLES"))):
        if all(q in c for q in k_in) and c[k_in[1]]:
            out["derived"][k_out] = c[k_in[0]] / c[k_in[1]]
    json.dump(out, open(a.out, "w"), indent=1)
    d = out["derived"]
    print(f"{a.kernel}: VALU busy {100 * d['valu_busy_fraction']:.1f} % (issue floor {100 * d['valu_issue_floor_fraction']:.1f} %), "
          f"lanes active {100 * d['lane_utilisation']:.1f} %, FP64 {100 * d['fp64_share_of_valu_instructions']:.1f} % of VALU "
          f"instructions, clock {d['shader_clock_GHz']:.2f} GHz, revision {out['meta']['kernel_source_sha']}")


if __name__ == "__main__":
    main()
