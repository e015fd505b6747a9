#!/bin/bash
# K8 A/B on ONE box: the working tree's library against build/ab/<name>/libsmc_hip.so (tools/ab_build.sh, e.g. built from a
# stash of HEAD): methanation GPU tests of the new build first, then tools/meth_dae_bench.py alternating, then the N = 4096 run.
#   tools/k8_ab.sh base [skip_n4096]
R=${GRAFT_REPO_ROOT:-/root/repo}
B=$R/build/ab/${1:-base}/libsmc_hip.so
O=$R/gpurun_out/k8ab_${1:-base}
rm -rf $O; mkdir -p $O
cd $R
log() { echo "$(date +%T) $*" | tee -a $O/progress.log; }
log "methanation GPU tests (new build)"
timeout -k 10 900 python -m pytest tests/test_gpu_methanation.py -q -m gpu > $O/pytest_meth.log 2>&1 || { grep -E '^(FAILED|ERROR|E  )' $O/pytest_meth.log | head -30; tail -2 $O/pytest_meth.log; exit 1; }
tail -1 $O/pytest_meth.log
for lib in new base new base new base; do
  if [ $lib = base ]; then export SMC_HIP_LIB=$B; else unset SMC_HIP_LIB; fi
  timeout -k 10 300 python3 tools/meth_dae_bench.py 512 2048 2>&1 | grep "solves/s" | sed "s/^/$lib: /" | tee -a $O/dae_bench.log || exit 1
done
[ -n "$2" ] && { log done; exit 0; }
for lib in new base; do
  log "N = 4096 run $lib"
  if [ $lib = base ]; then export SMC_HIP_LIB=$B; else unset SMC_HIP_LIB; fi
  timeout -k 10 400 python3 bench.py --workload methanation --particles-per-gpu 4096 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_n4096_$lib.json 2> $O/bench_n4096_$lib.err || { tail -5 $O/bench_n4096_$lib.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$O/bench_n4096_$lib.json').read().strip().splitlines()[-1]); print('$lib N=4096: %.2f s, %.0f solves/s, solves %d cancelled %d, steps %s sweeps %d, posterior mean %s logZ %s' % (d['ms_per_step']/1e3, d['dae_solves_per_s'], d['dae_solves'], d['dae_solves_cancelled'], d['tempering_steps_per_run'], d['mutation_sweeps'], [round(v,5) for v in d['posterior_mean']], d['logZ']))"
done
log done
