#!/bin/bash
# the methanation bench lines of round 4 on one box: N = 1024 (the population the K8 counter passes were taken at: carries
# roofline.traffic) with the CPU leg, then config 4 at its stated size (10^5 particles, one complete run, progress lines)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04lines
mkdir -p $O
cd $R
timeout -k 10 300 python3 bench.py --workload methanation --particles-per-gpu 1024 --steps 1 --warmup 0 > $O/bench_methanation_n1024.json 2> $O/n1024.err || { tail -5 $O/n1024.err; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/bench_methanation_n1024.json').read().strip().splitlines()[-1]); r=d['roofline']; print('N=1024: %.1f s, %.0f solves/s, frac %.4f, traffic %s, cpu %s' % (d['ms_per_step']/1e3, d['dae_solves_per_s'], r['frac'], r['traffic'], d.get('cpu_baseline',{}).get('dae_solves_per_s')))"
timeout -k 10 1000 python3 bench.py --workload methanation --particles-per-gpu 100000 --steps 1 --warmup 0 --no-cpu-baseline --progress > $O/bench_methanation_config4_full_run.json 2> $O/config4.err || { tail -5 $O/config4.err; exit 1; }
cp gpurun_out/bench_methanation_progress.log $O/config4_progress.log 2>/dev/null
python3 -c "
import json; d=json.loads(open('$O/bench_methanation_config4_full_run.json').read().strip().splitlines()[-1]); r=d['roofline']; print('config 4: %.1f s, %.0f solves/s, frac %.4f, solves %d cancelled %d' % (d['ms_per_step']/1e3, d['dae_solves_per_s'], r['frac'], d['dae_solves'], d['dae_solves_cancelled']))"
