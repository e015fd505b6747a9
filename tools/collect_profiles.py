#!/usr/bin/env python3
"""Condense what tools/final_profiles.sh left under gpurun_out/final/ into the tracked files profiles/rNN_*:
    python tools/collect_profiles.py r03
bench lines as they are, the rocprofv3 --stats kernel summaries (CSV), the FETCH/WRITE and SQ counter passes through
tools/pmc_summary.py / tools/pmc_sq_summary.py (stamped with the kernel-source hash bench.py checks)."""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, "gpurun_out", "final")
P = os.path.join(ROOT, "profiles")


def newest(pattern):
    files = glob.glob(os.path.join(F, pattern), recursive=True)
    return max(files, key=os.path.getmtime) if files else None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    copies = {"bench_mm.json": f"{tag}_bench_mm.json", "bench_mm_under_rocprof.json": f"{tag}_bench_mm_under_rocprof.json",
              "bench_mm_no_early_reject.json": f"{tag}_bench_mm_no_early_reject.json", "bench_mm_n1e7.json": f"{tag}_bench_mm_n1e7.json",
              "bench_mm_n1e8.json": f"{tag}_bench_mm_n1e8.json",
              "bench_methanation_n1024_under_rocprof.json": f"{tag}_bench_methanation_n1024_under_rocprof.json",
              "user_model_bench.log": f"{tag}_user_model_bench.log", "progress.log": f"{tag}_final_profiles_progress.log",
              "bench_mm_exact.json": f"{tag}_bench_mm_exact.json", "bench_mm_r3_host_loop.json": f"{tag}_bench_mm_r3_host_loop.json",
              "fp64_fma_peak.json": f"{tag}_fp64_fma_peak.json", "bench_methanation_n1024.json": f"{tag}_bench_methanation_n1024.json"}
    for src, dst in copies.items():
        if os.path.exists(os.path.join(F, src)) and os.path.getsize(os.path.join(F, src)) > 0:
            shutil.copy(os.path.join(F, src), os.path.join(P, dst))
            print("copied", dst)
        else:
            print("MISSING", src)
    with open(os.path.join(P, f"{tag}_steady_and_tail.log"), "w") as out:
        out.write("# tools/steady_state.py (1e6, 1e7 particles) and tools/tail_latency.py on the final kernel revision of the round\n")
        for f in ("steady.log", "tail.log"):
            if os.path.exists(os.path.join(F, f)):
                out.write(open(os.path.join(F, f)).read())
    for sub, dst in (("stats", f"{tag}_mm_kernel_stats.csv"), ("meth_stats", f"{tag}_methanation_kernel_stats.csv")):
        f = newest(os.path.join(sub, "**", "*kernel_stats.csv"))
        if f:
            shutil.copy(f, os.path.join(P, dst))
            print("copied", dst, "from", os.path.relpath(f, F))
        else:
            print("MISSING kernel stats under", sub)
    py = sys.executable
    subprocess.run([py, os.path.join(ROOT, "tools", "pmc_summary.py"), os.path.join(F, "pmc_fetch"), os.path.join(F, "pmc_write"),
                    os.path.join(P, f"{tag}_pmc_fetch_write_summary.json"), "--particles-per-gpu", "1000000", "--command",
                    "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"], check=False)
    subprocess.run([py, os.path.join(ROOT, "tools", "pmc_sq_summary.py"), os.path.join(P, f"{tag}_pmc_sq_summary.json"), "--kernel",
                    "mm_solve_kernel", "--last", "10", "--command", "python3 tools/steady_state.py 1000000 1"] +
                   [os.path.join(F, f"sq{i}") for i in (1, 2, 3, 4)], check=False)
    subprocess.run([py, os.path.join(ROOT, "tools", "pmc_sq_summary.py"), os.path.join(P, f"{tag}_pmc_sq_summary_whole_run.json"),
                    "--kernel", "mm_solve_kernel", "--command", "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"] +
                   [os.path.join(F, f"sqrun{i}") for i in (1, 2, 3, 4)], check=False)
    if os.path.isdir(os.path.join(F, "k8_pmc_fetch")):      # round 4: K8 (methanation DAE kernel), stamped with the hash of ITS sources
        k8cmd = "python3 bench.py --workload methanation --particles-per-gpu 1024 --steps 1 --warmup 0 --no-cpu-baseline"
        subprocess.run([py, os.path.join(ROOT, "tools", "pmc_summary.py"), os.path.join(F, "k8_pmc_fetch"), os.path.join(F, "k8_pmc_write"),
                        os.path.join(P, f"{tag}_k8_pmc_fetch_write_summary.json"), "--particles-per-gpu", "1024", "--family", "k8",
                        "--kernel", "meth_particles_dae_split_kernel", "--command", k8cmd], check=False)
        subprocess.run([py, os.path.join(ROOT, "tools", "pmc_sq_summary.py"), os.path.join(P, f"{tag}_k8_pmc_sq_summary.json"), "--kernel",
                        "meth_particles_dae_split_kernel", "--family", "k8", "--command", k8cmd] +
                       [os.path.join(F, f"k8_sq{i}") for i in (1, 2, 3, 4, 5, 6)], check=False)


if __name__ == "__main__":
    main()
