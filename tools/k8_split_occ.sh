#!/bin/bash
# K8 throughput against the number of resident solves per CU (SMC_METH_WAVES_PER_CU thins the persistent grid): v4 (two waves per
# solve) and v3 (one wave per solve) on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/k8split; rm -f gpurun_out/k8split/occ.log
for g in 1 2 3 4; do
  SMC_K8_SPLIT=1 SMC_METH_WAVES_PER_CU=$g timeout -k 10 300 python3 tools/meth_dae_bench.py 1024 2>&1 | grep -E "solves/s" | cut -c1-90 | sed "s/^/v4, $g solves per CU: /" | tee -a gpurun_out/k8split/occ.log
  SMC_K8_SPLIT=0 SMC_METH_WAVES_PER_CU=$g timeout -k 10 300 python3 tools/meth_dae_bench.py 1024 2>&1 | grep -E "solves/s" | cut -c1-90 | sed "s/^/v3, $g solves per CU: /" | tee -a gpurun_out/k8split/occ.log
done
