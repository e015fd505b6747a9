#!/bin/bash
# A/B on one box: idle lanes that trigger a hand-out (SMC_REFILL_AT; default 24)
mkdir -p gpurun_out/refill
for rep in 1 2; do
for lib in default refill12 refill16 refill32; do
  if [ $lib = default ]; then unset SMC_HIP_LIB; else export SMC_HIP_LIB=$PWD/build/ab/$lib/libsmc_hip.so; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/refill/bench_${lib}_$rep.json 2>gpurun_out/refill/err.log
done
done
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/refill/bench_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    ss = d.get("steady_state", {})
    print(f"{f.split('/')[-1]:28s} ms_per_step {d['ms_per_step']:7.2f}  steady solve {ss.get('solve_kernel_ms_per_sweep', 0):.3f} ms  mh avg {d['roofline'].get('mh_sweep_avg_ms', 0):.3f}  value {d['value']:.3e}")
P
