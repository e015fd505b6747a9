#!/bin/bash
# A/B on one box: threshold of the solo list (Vmax > ratio * Km) now that a solo solve runs the hand-written loop
mkdir -p gpurun_out/solo2
for rep in 1 2; do
for lib in default solo200 solo400 solo700; do
  if [ $lib = default ]; then unset SMC_HIP_LIB; else export SMC_HIP_LIB=$PWD/build/ab/$lib/libsmc_hip.so; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/solo2/bench_${lib}_$rep.json 2>gpurun_out/solo2/err.log
done
done
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/solo2/bench_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    k = d.get("kernels", {})
    print(f"{f.split('/')[-1]:28s} ms_per_step {d['ms_per_step']:7.2f}  value {d['value']:.3e}")
P
