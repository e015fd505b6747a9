#!/bin/bash
# round 3, GPU call A: parity tests, then A/B on one box: round-2 library vs working tree (stiff-first on / off, list entries
# per chunk), the lone stiff chain, the steady-state sweep.
out=gpurun_out/r03a; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -5 $out/pytest.log
if [ $rc -ne 0 ]; then tail -60 $out/pytest.log; exit $rc; fi
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
for rep in 1 2; do
  SMC_HIP_LIB=build/ab/r02/libsmc_hip.so $B > $out/bench_r02_$rep.json 2>$out/err.log || exit 1
  $B > $out/bench_new_$rep.json 2>>$out/err.log || exit 1
  $B --no-stiff-first > $out/bench_new_nostiff_$rep.json 2>>$out/err.log || exit 1
  SMC_HIP_LIB=build/ab/spc64/libsmc_hip.so $B > $out/bench_spc64_$rep.json 2>>$out/err.log || exit 1
  SMC_HIP_LIB=build/ab/spc4/libsmc_hip.so $B > $out/bench_spc4_$rep.json 2>>$out/err.log || exit 1
done
python - <<'PY' | tee $out/summary.txt
import json,glob
for f in sorted(glob.glob("gpurun_out/r03a/bench_*.json")):
    d=json.load(open(f)); k=d["kernel_ms"]
    print(f"{f.split('/')[-1]:28s} ms_per_step {d['ms_per_step']:7.2f}  loglik/launch {k['loglik']['ms']/k['loglik']['launches']:6.2f} ms  mh/launch {k['mh']['ms']/k['mh']['launches']:6.3f} ms  steady {d['steady_state']['solve_kernel_ms_per_sweep']:.3f} ms  value {d['value']:.3e}")
PY
for v in r02 new; do
  lib=""; [ $v = r02 ] && lib=build/ab/r02/libsmc_hip.so
  echo "== $v" | tee -a $out/tail.log
  SMC_HIP_LIB=$lib python tools/tail_latency.py 2>&1 | tee -a $out/tail.log
  SMC_HIP_LIB=$lib python tools/steady_state.py 1000000 1 2>&1 | tee -a $out/tail.log
done
