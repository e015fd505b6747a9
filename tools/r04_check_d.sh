#!/bin/bash
# rocprofv3 per-kernel averages of the MM bench for several accept-kernel grid caps
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04d
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cap in 1024 512 256; do
  SMC_FINISH_GRID_CAP=$cap timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$cap -o run -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$cap.json 2> $O/stats_$cap.err || { tail -5 $O/stats_$cap.err; exit 1; }
  f=$(find $O/stats_$cap -name "*kernel_stats.csv" | head -1)
  echo "== cap $cap"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("finish", "propose", "cost_", "mh_control", "moments_reduce", "mh_transform", "mm_solve", "resample", "ess_partial", "max_")):
        print(f"{n[:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
done
