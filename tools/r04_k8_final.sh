#!/bin/bash
# K8 at HEAD: counter passes (tools/r04_k8_profiles.sh), then config 4 complete
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/r04_k8_profiles.sh > $R/gpurun_out/r04k8_profiles_stdout.log 2>&1 || { tail -5 $R/gpurun_out/r04k8_profiles_stdout.log; exit 1; }
tail -4 $R/gpurun_out/r04k8_profiles_stdout.log | cut -c1-400
cd $R
O=$R/gpurun_out/r04lines2
mkdir -p $O
rm -f gpurun_out/bench_methanation_progress.log
timeout -k 10 900 python3 bench.py --workload methanation --particles-per-gpu 100000 --steps 1 --warmup 0 --no-cpu-baseline --progress > $O/bench_methanation_config4_full_run.json 2> $O/config4.err || { tail -5 $O/config4.err; exit 1; }
cp gpurun_out/bench_methanation_progress.log $O/config4_progress.log 2>/dev/null
python3 -c "
import json; d=json.loads(open('$O/bench_methanation_config4_full_run.json').read().strip().splitlines()[-1]); r=d['roofline']; print('config 4: %.1f s, %.0f solves/s, frac %.4f, solves %d cancelled %d, steps %s sweeps %d' % (d['ms_per_step']/1e3, d['dae_solves_per_s'], r['frac'], d['dae_solves'], d['dae_solves_cancelled'], d['tempering_steps_per_run'], d['mutation_sweeps']))"
