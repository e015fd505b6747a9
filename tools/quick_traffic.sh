#!/bin/bash
# HBM bytes per launch of the solve kernel (two --pmc passes of a short bench run) + a clean bench line, on one box
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/qt
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_fetch.err || exit 1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_write.err || exit 1
cd $R
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/summary.json --particles-per-gpu 1000000 --command quick
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'value %.4g' % d['value'])"
