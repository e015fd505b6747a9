#!/bin/bash
# Generic A/B runner (round 5; replaces the one-off ab_*.sh / k8_*.sh / r04_*.sh wrappers of rounds 2-4): runs ONE command REPS
# times for each library, alternating between the libraries, on the box it is started on, and keeps every output.
#   tools/ab_build.sh k8p0 -DSMC_K8_POLICY=0                         # a variant build under build/ab/<name>/
#   tools/ab_run.sh gpurun_out/myab 2 "python3 tools/meth_dae_bench.py 512 2048" tree k8p0
# `tree` = the in-tree libsmc_hip.so; any other name = build/ab/<name>/libsmc_hip.so (handed over through SMC_HIP_LIB).
set -u
out=$1; reps=$2; cmd=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$out"
for rep in $(seq 1 "$reps"); do
  for lib in "$@"; do
    if [ "$lib" = tree ]; then unset SMC_HIP_LIB; else export SMC_HIP_LIB=$R/build/ab/$lib/libsmc_hip.so; fi
    echo "$(date +%T) rep $rep $lib: $cmd" | tee -a "$out/progress.log"
    ( cd "$R" && timeout -k 10 "${AB_TIMEOUT:-600}" bash -c "$cmd" ) > "$out/${lib}_rep$rep.log" 2>&1 || echo "   exit $?" | tee -a "$out/progress.log"
    tail -n "${AB_TAIL:-3}" "$out/${lib}_rep$rep.log"
  done
done
