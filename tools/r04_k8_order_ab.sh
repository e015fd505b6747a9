#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04order
mkdir -p $O
cd $R
for rep in 1 2; do
for mode in ratio misfit; do
  SMC_METH_ORDER=$mode timeout -k 10 400 python3 bench.py --workload methanation --particles-per-gpu 4096 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_$mode$rep.json 2> $O/err_$mode$rep.log || { tail -3 $O/err_$mode$rep.log; exit 1; }
  python3 -c "
import json; d=json.loads(open('$O/bench_$mode$rep.json').read().strip().splitlines()[-1]); print('$mode $rep N=4096: %.2f s, %.0f solves/s, solves %d cancelled %d (%.1f %%), logZ %.6f' % (d['ms_per_step']/1e3, d['dae_solves_per_s'], d['dae_solves'], d['dae_solves_cancelled'], 100*d['dae_solves_cancelled']/(d['dae_solves']+d['dae_solves_cancelled']), d['logZ'][0]))"
done
done
