cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/k8split
for g in 4 3 2 1; do
  SMC_K8_SPLIT=1 SMC_K8_SPLIT_DEBUG=1 SMC_METH_WAVES_PER_CU=$g timeout -k 10 300 python3 tools/meth_dae_bench.py 1024 2>&1 | grep -E "solves/s|k8 split" | sed "s/^/v4 groups per CU $g: /" | tee -a gpurun_out/k8split/occ.log
done
for g in 4 3 2; do
  SMC_K8_SPLIT=0 SMC_METH_WAVES_PER_CU=$g timeout -k 10 300 python3 tools/meth_dae_bench.py 1024 2>&1 | grep -E "solves/s" | sed "s/^/v3 waves per CU $g: /" | tee -a gpurun_out/k8split/occ.log
done
