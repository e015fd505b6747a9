#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04c
mkdir -p $O
cd $R
run() { # name, flags
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $2 > $O/bench_$1.json 2> $O/bench_$1.err || { tail -5 $O/bench_$1.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1]); print('$1: ms_per_step %.2f' % d['ms_per_step'], 'value %.4g' % d['value'], 'mh syncs', d['mh_loop_synchronisations'], {k: round(v['ms']/20,2) for k,v in d['kernel_ms'].items()})"
}
for rep in 1 2; do
run "all_on_$rep" ""
run "r3_style_$rep" "--mh-batch 0 --no-defer-resample --no-pinned-results"
run "no_pinned_$rep" "--no-pinned-results"
run "no_defer_$rep" "--no-defer-resample"
run "mb0_$rep" "--mh-batch 0"
done
