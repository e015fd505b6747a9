#!/bin/bash
# A/B on one box: cost-ordered, in-phase hand-out of heterogeneous Metropolis sweeps (smc_set_cost_order) on / off
mkdir -p gpurun_out/cost
python -m pytest tests/test_gpu_parity.py -q -x -k "cost_ordered" > gpurun_out/cost/pytest.log 2>&1; tail -3 gpurun_out/cost/pytest.log
for rep in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/cost/bench_on_$rep.json 2>gpurun_out/cost/err.log
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cost-order > gpurun_out/cost/bench_off_$rep.json 2>>gpurun_out/cost/err.log
done
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/cost/bench_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    ss = d.get("steady_state", {})
    print(f"{f.split('/')[-1]:28s} ms_per_step {d['ms_per_step']:7.2f}  steady solve {ss.get('solve_kernel_ms_per_sweep', 0):.3f} ms  mh avg {d['roofline'].get('mh_sweep_avg_ms', 0):.3f}  value {d['value']:.3e}")
P
python tools/sweep_profile.py 1000000 1000 | tail -36
