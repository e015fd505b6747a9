#!/bin/bash
# the whole -m gpu suite + smoke + a bench line on one box
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04tests
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/ -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?
tail -5 $O/pytest_gpu.log; grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head -20
python -c "import __graft_entry__ as e; e.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for mb in auto 0 auto 0; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --mh-batch $mb > $O/bench_mb_$mb.json 2> $O/bench_mb_$mb.err || { tail -5 $O/bench_mb_$mb.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$O/bench_mb_$mb.json').read().strip().splitlines()[-1]); print('mh_batch $mb: ms_per_step %.2f' % d['ms_per_step'], 'mh syncs', d['mh_loop_synchronisations'], 'noop', d['mh_speculative_noop_sweeps'], {k: round(v['ms']/20,2) for k,v in d['kernel_ms'].items()})"
done
exit $rc
