#!/bin/bash
# SQ counters of the solve kernel on the proposals of a mid-run sweep (tools/sort_probe.py <sweep> <n> <order>): four --pmc passes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/midrun
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SW=${1:-12}
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64" "SQ_WAVES SQ_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS"; do
  i=$((i+1))
  echo "pass $i" >> $O/progress.log
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/sq$i -- python3 $R/tools/sort_probe.py $SW 1000000 "run order" > $O/sq$i.log 2>&1 || exit 1
done
cd $R
python3 tools/pmc_sq_summary.py gpurun_out/midrun/summary.json --kernel mm_solve_kernel --last 5 --command "tools/sort_probe.py $SW" $O/sq1 $O/sq2 $O/sq3 $O/sq4
