#!/usr/bin/env python3
"""Condense two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE need separate passes: MI355X_MICROARCH.md, counter table)
into the per-kernel summary bench.py reads (profiles/rNN_pmc_fetch_write_summary.json), stamped with the kernel revision
(sha256 of the library sources) and the population size, so that bench.py only reports `roofline.traffic` for the kernel
build and workload the counters were collected on.

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE -d $REPO/gpurun_out/pmc_fetch -o run -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE -d $REPO/gpurun_out/pmc_write -o run -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline
  python3 tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r02_pmc_fetch_write_summary.json \
          --particles-per-gpu 1000000 --command "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
"""
import argparse
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_kernel(directory, counter):
    """kernel name (template arguments kept, parameter list dropped) -> dispatches, mean and max counter value per dispatch."""
    acc = {}
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {directory}")
    for f in [max(files, key=os.path.getmtime)]:      # one run per directory: the newest (gpurun merges, it never deletes)
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(.*$", "", row["Kernel_Name"]).strip()
            acc.setdefault(name, []).append(float(row["Counter_Value"]))
    return {k: {"dispatches": len(v), "avg_counter_value": sum(v) / len(v), "max": max(v)} for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("out")
    ap.add_argument("--particles-per-gpu", type=int, required=True)
    ap.add_argument("--command", default="")
    ap.add_argument("--family", choices=["mm", "k8"], default="mm", help="which sources the revision hash covers (bench.kernel_source_sha)")
    ap.add_argument("--kernel", default="mm_solve_kernel", help="kernel whose traffic is printed")
    a = ap.parse_args()
    import bench
    out = {"meta": {"kernel_source_sha": bench.kernel_source_sha(family=a.family), "particles_per_gpu": a.particles_per_gpu,
                    "command": a.command, "unit": "KiB per dispatch (rocprofv3 FETCH_SIZE / WRITE_SIZE); on gfx950 FETCH_SIZE "
                                                  "reports half of the bytes fetched (MI355X_MICROARCH.md, HBM section)"},
           "pmc_fetch": per_kernel(a.fetch_dir, "FETCH_SIZE"), "pmc_write": per_kernel(a.write_dir, "WRITE_SIZE")}
    json.dump(out, open(a.out, "w"), indent=1)
    for k in sorted(out["pmc_fetch"]):
        if a.kernel not in k or k not in out["pmc_write"]:
            continue
        b = (2 * out["pmc_fetch"][k]["avg_counter_value"] + out["pmc_write"][k]["avg_counter_value"]) * 1024
        print(f"{k}: {b / 1e6:.1f} MB of HBM traffic per launch ({out['pmc_fetch'][k]['dispatches']} dispatches); "
              f"kernel revision {out['meta']['kernel_source_sha']}")


if __name__ == "__main__":
    main()
