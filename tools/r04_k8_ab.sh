#!/bin/bash
# K8 two-ended elimination (round 4) against the one-way scans of rounds 1-3 on ONE box: correctness first (GPU tests of the
# methanation path, v3 vs v2), then throughput (tools/meth_dae_bench.py), cycle counters, and the N = 4096 run
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04k8ab
rm -rf $O; mkdir -p $O
cd $R
log() { echo "$(date +%T) $*" | tee -a $O/progress.log; }
log "methanation GPU tests (two-ended build)"
timeout -k 10 900 python -m pytest tests/test_gpu_methanation.py -q -m gpu > $O/pytest_meth.log 2>&1 || { grep -E '^(FAILED|ERROR|E  )' $O/pytest_meth.log | head -30; }
tail -2 $O/pytest_meth.log
log "v3 vs v2"
timeout -k 10 600 python3 tools/meth_v3_check.py 8 > $O/v3_vs_v2.log 2>&1; tail -3 $O/v3_vs_v2.log
for lib in twisted oneway twisted oneway; do
  log "dae bench $lib"
  if [ $lib = oneway ]; then export SMC_HIP_LIB=$R/build/ab/oneway/libsmc_hip.so; else unset SMC_HIP_LIB; fi
  timeout -k 10 300 python3 tools/meth_dae_bench.py 64 512 2048 2>&1 | tee -a $O/dae_bench_$lib.log | tail -3
done
for lib in twisted_prof oneway_prof; do
  log "cycle counters $lib"
  SMC_HIP_LIB=$R/build/ab/$lib/libsmc_hip.so timeout -k 10 300 python3 tools/meth_dae_bench.py 512 > $O/prof_$lib.log 2>&1; grep "meth profile" $O/prof_$lib.log | head -20
done
unset SMC_HIP_LIB
for lib in twisted oneway; do
  log "N = 4096 run $lib"
  if [ $lib = oneway ]; then export SMC_HIP_LIB=$R/build/ab/oneway/libsmc_hip.so; else unset SMC_HIP_LIB; fi
  timeout -k 10 400 python3 bench.py --workload methanation --particles-per-gpu 4096 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_n4096_$lib.json 2> $O/bench_n4096_$lib.err
  python3 -c "
import json; d=json.loads(open('$O/bench_n4096_$lib.json').read().strip().splitlines()[-1]); print('$lib N=4096: %.1f s, %.0f solves/s, frac %.4f, solves %d cancelled %d, posterior mean %s logZ %s' % (d['ms_per_step']/1e3, d['dae_solves_per_s'], d['roofline']['frac'], d['dae_solves'], d['dae_solves_cancelled'], [round(v,4) for v in d['posterior_mean']], d['logZ']))"
done
log done
