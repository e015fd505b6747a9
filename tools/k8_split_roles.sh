#!/bin/bash
# K8 v4: which wave of the workgroup integrates (SMC_K8_SPLIT_ROLES 0 = always wave 0, 1 = by the parity of wave 0's hardware slot)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/k8split
mkdir -p $O; cd $R
timeout -k 10 240 python3 tools/meth_v3_check.py 8 v4 v3 2>&1 | tail -1 || exit 1
for rep in 1 2; do
 for pol in 0 1; do
  SMC_K8_SPLIT=1 SMC_K8_SPLIT_ROLES=$pol timeout -k 10 300 python3 tools/meth_dae_bench.py 512 2048 2>&1 | grep "solves/s" | sed "s/^/roles=$pol: /" | tee -a $O/roles.log || exit 1
 done
done
SMC_K8_SPLIT=0 timeout -k 10 300 python3 tools/meth_dae_bench.py 512 2048 2>&1 | grep "solves/s" | sed "s/^/v3: /" | tee -a $O/roles.log
