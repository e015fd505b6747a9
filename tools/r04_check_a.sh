#!/bin/bash
# round 4, first GPU call: the new tests, then the whole-run bench with the loop control on the device vs on the host
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04a
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "device_side_mh or mh_batch or cost_ordered or fused or ess_search or loopback or one_rank or full_run or early_rejection" > $O/pytest_parity_subset.log 2>&1 || { grep -E '^(FAILED|ERROR|E  )' $O/pytest_parity_subset.log | head -40; }
tail -2 $O/pytest_parity_subset.log
timeout -k 10 600 python -m pytest tests/test_gpu_methanation.py -x -q -m gpu -k "sharded" > $O/pytest_meth_sharded.log 2>&1 || { grep -E '^(FAILED|ERROR|E  )' $O/pytest_meth_sharded.log | head -40; }
tail -2 $O/pytest_meth_sharded.log
for mb in auto 0 auto 0; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --mh-batch $mb > $O/bench_mb_$mb.json 2> $O/bench_mb_$mb.err || { tail -5 $O/bench_mb_$mb.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$O/bench_mb_$mb.json').read().strip().splitlines()[-1]); print('mh_batch $mb: ms_per_step %.2f' % d['ms_per_step'], 'value %.4g' % d['value'], 'mh syncs', d['mh_loop_synchronisations'], 'noop', d['mh_speculative_noop_sweeps'], {k: round(v['ms']/20,2) for k,v in d['kernel_ms'].items()})"
done
for cap in 1024 2048 4096; do
  SMC_FINISH_GRID_CAP=$cap timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cap_$cap.json 2> $O/bench_cap_$cap.err || exit 1
  python3 -c "
import json; d=json.loads(open('$O/bench_cap_$cap.json').read().strip().splitlines()[-1]); print('finish cap $cap: ms_per_step %.2f' % d['ms_per_step'], {k: round(v['ms']/20,2) for k,v in d['kernel_ms'].items()})"
done
