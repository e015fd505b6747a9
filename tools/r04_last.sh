#!/bin/bash
# end of round 4: the whole -m gpu suite, smoke, and the two bench lines with the committed counter summaries in place
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04last
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/ -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?
tail -3 $O/pytest_gpu.log; grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head
python -c "import __graft_entry__ as e; e.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $O/bench_mm.json 2> $O/bench_mm.err || { tail -3 $O/bench_mm.err; exit 1; }
timeout -k 10 300 python3 bench.py --workload methanation --particles-per-gpu 1024 --steps 1 --warmup 0 > $O/bench_methanation_n1024.json 2> $O/bench_meth.err || { tail -3 $O/bench_meth.err; exit 1; }
python3 - <<PY
import json
d=json.loads(open("$O/bench_mm.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("MM: ms_per_step %.2f value %.4g frac %.4f traffic %s valu %s cpu %s" % (d["ms_per_step"], d["value"], r["frac"], r["traffic"], (r.get("valu_issue") or {}).get("valu_busy_fraction"), d.get("cpu_baseline",{}).get("value")))
d=json.loads(open("$O/bench_methanation_n1024.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("methanation N=1024: %.2f s, %.0f solves/s, frac %.4f, traffic %s, cpu %s" % (d["ms_per_step"]/1e3, d["dae_solves_per_s"], r["frac"], r["traffic"], d.get("cpu_baseline",{}).get("dae_solves_per_s")))
PY
exit $rc
