#!/bin/bash
# K8 v4 (two waves per solve) on the box: a small run first (a hang shows within seconds: tight timeouts), then v4 against v3 on
# the same solves, then throughput.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/k8split
rm -rf $O; mkdir -p $O
cd $R
echo "--- 2 particles x 30 solves, v4 vs v3"
timeout -k 10 240 python3 tools/meth_v3_check.py 2 v4 v3 2>&1 | tee $O/check2.log | tail -4 || exit 1
echo "--- 64 particles"
timeout -k 10 300 python3 tools/meth_v3_check.py 64 v4 v3 2>&1 | tee $O/check64.log | tail -4 || exit 1
for v in 1 0 1 0; do
  SMC_K8_SPLIT=$v timeout -k 10 300 python3 tools/meth_dae_bench.py 512 2048 2>&1 | grep "solves/s" | sed "s/^/split=$v: /" | tee -a $O/dae_bench.log || exit 1
done
