#!/bin/bash
# working tree's K8 against build/ab/<name> (default: head), both with the default (two-wave) kernel: bit-identity, then throughput
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
B=$R/build/ab/${1:-head}/libsmc_hip.so
O=$R/gpurun_out/k8ab2; mkdir -p $O
python3 - <<PY || exit 1
import os, sys, subprocess, json
import numpy as np
out = {}
for tag, lib in (("new", ""), ("base", "$B")):
    env = dict(os.environ, SMC_CHILD=f"/tmp/meth_{tag}.npz")
    if lib: env["SMC_HIP_LIB"] = lib
    subprocess.run([sys.executable, "tools/meth_v3_check.py", "16"], env=env, check=True, timeout=300)
    out[tag] = np.load(f"/tmp/meth_{tag}.npz")
a, b = out["new"], out["base"]
print("new vs base: status equal", np.array_equal(a["status"], b["status"]), " flows identical", np.array_equal(a["flows"], b["flows"]), a["info"] == b["info"] or (a["info"], b["info"]))
PY
for rep in 1 2 3; do
  for lib in new base; do
    if [ $lib = base ]; then export SMC_HIP_LIB=$B; else unset SMC_HIP_LIB; fi
    timeout -k 10 300 python3 tools/meth_dae_bench.py 512 2048 2>&1 | grep "solves/s" | cut -c1-100 | sed "s/^/$lib: /" | tee -a $O/dae_bench.log || exit 1
  done
done
