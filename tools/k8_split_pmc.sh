#!/bin/bash
# SQ counter passes on the two-wave K8 kernel (SMC_K8_SPLIT=1) and, for the same box, on the one-wave kernel: where do the wave
# cycles go when two waves share a SIMD?  Counters in their own passes, program directly after --.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/k8splitpmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
N=${K8_N:-1024}
CMD="python3 $R/bench.py --workload methanation --particles-per-gpu $N --steps 1 --warmup 0 --no-cpu-baseline"
for v in 1 0; do
  export SMC_K8_SPLIT=$v
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_WAVES SQ_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_IFETCH SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
    i=$((i+1))
    echo "$(date +%T) split=$v pass $i" >> $O/progress.log
    timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/v$v/sq$i -- $CMD > $O/v${v}_sq$i.log 2>&1 || echo "pass $i failed (split=$v)" | tee -a $O/progress.log
  done
done
cd $R
python3 - <<PY
import csv, glob, collections
for v, kn in ((1, "meth_particles_dae_split_kernel"), (0, "meth_particles_dae_kernel")):
    tot = collections.defaultdict(float); nd = collections.defaultdict(int)
    for f in glob.glob("$O/v%d/sq*/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Kernel_Name", "").startswith(kn + "("):
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); nd[r["Counter_Name"]] += 1
    print("split=%d %s" % (v, kn))
    for k in sorted(tot): print("   %-26s %16.0f  (%d dispatches)" % (k, tot[k] / max(nd[k], 1), nd[k]))
PY
