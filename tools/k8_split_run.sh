#!/bin/bash
# K8 v4 in the SMC run: methanation GPU tests with SMC_K8_SPLIT=1, then the N = 4096 run with both kernels (same seed: the results
# must be identical - v4 reproduces v3 bit for bit)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/k8splitrun
rm -rf $O; mkdir -p $O
cd $R
echo "$(date +%T) methanation GPU tests, split" | tee -a $O/progress.log
SMC_K8_SPLIT=1 timeout -k 10 900 python -m pytest tests/test_gpu_methanation.py -q -m gpu > $O/pytest_meth.log 2>&1 || { grep -E '^(FAILED|ERROR|E  )' $O/pytest_meth.log | head -30; tail -2 $O/pytest_meth.log; exit 1; }
tail -1 $O/pytest_meth.log
for v in 1 0; do
  echo "$(date +%T) N = 4096 run split=$v" | tee -a $O/progress.log
  SMC_K8_SPLIT=$v timeout -k 10 400 python3 bench.py --workload methanation --particles-per-gpu 4096 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_n4096_$v.json 2> $O/bench_n4096_$v.err || { tail -5 $O/bench_n4096_$v.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$O/bench_n4096_$v.json').read().strip().splitlines()[-1]); print('split=$v N=4096: %.2f s, %.0f solves/s, solves %d cancelled %d, steps %s sweeps %d, posterior mean %s logZ %s' % (d['ms_per_step']/1e3, d['dae_solves_per_s'], d['dae_solves'], d['dae_solves_cancelled'], d['tempering_steps_per_run'], d['mutation_sweeps'], [round(v,6) for v in d['posterior_mean']], d['logZ']))"
done
