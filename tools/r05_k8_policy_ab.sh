#!/bin/bash
# Round 5: K8's control policy (IDA's matrix reuse + Newton test, SMC_K8_POLICY=1, the in-tree build) against rounds 1-4's
# (SciPy's, build/ab/k8p0) and two other reuse windows (k8x15, k8x10) on ONE box: correctness first, then throughput
# alternating, then complete runs at N = 1024 and N = 4096.   tools/ab_build.sh k8p0 -DSMC_K8_POLICY=0 ; ... k8x15 -DSMC_K8_XRATE=0.15
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05k8ab
rm -rf $O; mkdir -p $O
cd $R
log() { echo "$(date +%T) $*" | tee -a $O/progress.log; }
log "methanation GPU tests (policy-1 build)"
timeout -k 10 1000 python -m pytest tests/test_gpu_methanation.py -q -m gpu > $O/pytest_meth.log 2>&1 || { grep -E '^(FAILED|ERROR|E  )' $O/pytest_meth.log | head -30; }
tail -2 $O/pytest_meth.log
for lib in p1 k8p0 k8x15 k8x10 p1 k8p0; do
  log "dae bench $lib"
  if [ $lib = p1 ]; then unset SMC_HIP_LIB; else export SMC_HIP_LIB=$R/build/ab/$lib/libsmc_hip.so; fi
  timeout -k 10 300 python3 tools/meth_dae_bench.py 512 2048 2>&1 | tee -a $O/dae_bench_$lib.log | tail -2
done
for lib in p1 k8p0 k8x15; do
  for n in 1024 4096; do
    log "N = $n run $lib"
    if [ $lib = p1 ]; then unset SMC_HIP_LIB; else export SMC_HIP_LIB=$R/build/ab/$lib/libsmc_hip.so; fi
    timeout -k 10 400 python3 bench.py --workload methanation --particles-per-gpu $n --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_n${n}_$lib.json 2> $O/bench_n${n}_$lib.err
    python3 -c "
import json; d=json.loads(open('$O/bench_n${n}_$lib.json').read().strip().splitlines()[-1]); print('$lib N=$n: %.2f s, %.0f solves/s, frac %.4f, solves %d cancelled %d, per solve %s, posterior mean %s std %s logZ %s' % (d['ms_per_step']/1e3, d['dae_solves_per_s'], d['roofline']['frac'], d['dae_solves'], d['dae_solves_cancelled'], {k: round(v,1) for k,v in d.get('per_solve',{}).items()}, [round(v,4) for v in d['posterior_mean']], [round(v,4) for v in d['posterior_std']], d['logZ']))" | tee -a $O/runs.log
  done
done
log done
