#!/bin/bash
# Where the dispatcher puts the two waves of the K8 v4 workgroups: a census build (tools/ab_build.sh census -DSMC_K8_CENSUS, made in
# the build container) prints one line per wave from HW_REG_HW_ID; the summary is profiles/r04_k8_split_placement.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/k8split
for pol in 4 5; do
SMC_HIP_LIB=$PWD/build/ab/census/libsmc_hip.so SMC_K8_SPLIT=1 SMC_K8_SPLIT_ROLES=$pol timeout -k 10 300 python3 tools/meth_dae_bench.py 512 > gpurun_out/k8split/census_$pol.log 2>&1
grep -c "k8 placement" gpurun_out/k8split/census_$pol.log
done
