#!/bin/bash
# A/B on one box: the hand-written lone-chain loop (smc_set_fast_tail) in the default build (3 waves per SIMD) and in a build
# held at 4 waves per SIMD (tools/ab_build.sh w4 -DSMC_SOLVE_WAVES=4), against the compiled step function.
mkdir -p gpurun_out/fast
python -m pytest tests/test_gpu_parity.py -q -x -k "hand_written" > gpurun_out/fast/pytest.log 2>&1; tail -3 gpurun_out/fast/pytest.log
for lib in default w4; do
  if [ $lib = default ]; then unset SMC_HIP_LIB; else export SMC_HIP_LIB=$PWD/build/ab/$lib/libsmc_hip.so; fi
  echo "== $lib, block on"; python tools/tail_latency.py built-in
  echo "== $lib, block off"; python tools/tail_latency.py built-in-nofast
  for rep in 1 2; do
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/fast/bench_${lib}_on_$rep.json 2>gpurun_out/fast/err.log
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fast-tail > gpurun_out/fast/bench_${lib}_off_$rep.json 2>>gpurun_out/fast/err.log
  done
done
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/fast/bench_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    ss = d.get("steady_state", {})
    print(f"{f.split('/')[-1]:28s} ms_per_step {d['ms_per_step']:7.2f}  steady solve {ss.get('solve_kernel_ms_per_sweep', 0):.3f} ms  value {d['value']:.3e}")
P
