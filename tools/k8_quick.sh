#!/bin/bash
# K8 quick look on one box: v4 against v3 bit for bit, throughput of both, HBM traffic (FETCH/WRITE passes) of the v4 run at N = 1024
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/k8quick
rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 240 python3 tools/meth_v3_check.py 16 v4 v3 2>&1 | tail -1 || exit 1
for v in 1 0 1 0; do
  SMC_K8_SPLIT=$v timeout -k 10 300 python3 tools/meth_dae_bench.py 512 2048 2>&1 | grep "solves/s" | cut -c1-100 | sed "s/^/split=$v: /" | tee -a $O/dae_bench.log || exit 1
done
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --workload methanation --particles-per-gpu 1024 --steps 1 --warmup 0 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o run -- $CMD > $O/fetch.json 2> $O/pmc_fetch.err || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o run -- $CMD > /dev/null 2> $O/pmc_write.err || exit 1
cd $R
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/k8_pmc_fetch_write_summary.json --particles-per-gpu 1024 --family k8 --kernel meth_particles_dae_split_kernel --command "bench.py --workload methanation --particles-per-gpu 1024 --steps 1 --warmup 0 --no-cpu-baseline" | tail -2
python3 -c "
import json; d=json.loads(open('$O/fetch.json').read().strip().splitlines()[-1]); print('N=1024 run under rocprof: %.2f s, %.0f solves/s' % (d['ms_per_step']/1e3, d['dae_solves_per_s']))"
