#!/bin/bash
# A/B on one box: FAST instantiation (hand-written lone-chain loop) for every population size (build fastall) against the
# default build, which launches it up to kFastTailMaxParticles particles per sweep only.
mkdir -p gpurun_out/fast2
python -m pytest tests/test_gpu_parity.py -q -x -k "hand_written or stiff_first_handout" > gpurun_out/fast2/pytest.log 2>&1; tail -2 gpurun_out/fast2/pytest.log
for lib in default fastall; do
  if [ $lib = default ]; then unset SMC_HIP_LIB; else export SMC_HIP_LIB=$PWD/build/ab/$lib/libsmc_hip.so; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/fast2/bench_${lib}_1e6.json 2>gpurun_out/fast2/err.log
  python bench.py --particles-per-gpu 10000000 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/fast2/bench_${lib}_1e7.json 2>>gpurun_out/fast2/err.log
  python bench.py --particles-per-gpu 100000000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/fast2/bench_${lib}_1e8.json 2>>gpurun_out/fast2/err.log
  echo "== $lib"; python tools/steady_state.py 1000000 1; python tools/steady_state.py 10000000 1
done
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/fast2/bench_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    ss = d.get("steady_state", {})
    print(f"{f.split('/')[-1]:28s} ms_per_step {d['ms_per_step']:8.2f}  steady solve {ss.get('solve_kernel_ms_per_sweep', 0):.3f} ms  value {d['value']:.3e}  {d['roofline']['kernel'][:40]}")
P
