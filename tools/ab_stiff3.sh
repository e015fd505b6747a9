#!/bin/bash
# A/B on one box: threshold of the stiff list with the cost order in place (default 60)
mkdir -p gpurun_out/stiff3
for rep in 1 2 3; do
for lib in default stiff30 stiff40 stiff50; do
  if [ $lib = default ]; then unset SMC_HIP_LIB; else export SMC_HIP_LIB=$PWD/build/ab/$lib/libsmc_hip.so; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/stiff3/bench_${lib}_$rep.json 2>gpurun_out/stiff3/err.log
done
done
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/stiff3/bench_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f"{f.split('/')[-1]:28s} ms_per_step {d['ms_per_step']:7.2f}  value {d['value']:.3e}")
P
