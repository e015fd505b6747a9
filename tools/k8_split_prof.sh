#!/bin/bash
# cycle counters of K8 v4's integrator wave (a -DSMC_METH_PROFILE build: tools/ab_build.sh v4prof -DSMC_METH_PROFILE), with four
# solves per CU and with one (no other waves on the CU), and of v3 for comparison
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/k8split; rm -f gpurun_out/k8split/prof.log
for g in 4 1; do
  for v in 1 0; do
    echo "== v$((3+v)), $g solves per CU" | tee -a gpurun_out/k8split/prof.log
    SMC_HIP_LIB=$PWD/build/ab/v4prof/libsmc_hip.so SMC_K8_SPLIT=$v SMC_METH_WAVES_PER_CU=$g timeout -k 10 300 python3 tools/meth_dae_bench.py 512 2>&1 | grep -E "meth profile|solves/s" | cut -c1-110 | tee -a gpurun_out/k8split/prof.log
  done
done
