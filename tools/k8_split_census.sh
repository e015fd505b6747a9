cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/k8split
for pol in 4 5; do
SMC_K8_SPLIT=1 SMC_K8_SPLIT_ROLES=$pol timeout -k 10 300 python3 tools/meth_dae_bench.py 512 > gpurun_out/k8split/census_$pol.log 2>&1
grep -c "k8 placement" gpurun_out/k8split/census_$pol.log
done
