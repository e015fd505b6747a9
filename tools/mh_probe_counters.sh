#!/bin/bash
# SQ counters of the solve kernel for one replayed Metropolis sweep (tools/mh_probe.py <sweep> <n> default|index)
R=${GRAFT_REPO_ROOT:-/root/repo}
SW=${1:-13}; V=${2:-default}
O=$R/gpurun_out/mhc_$V
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64" "SQ_WAVES SQ_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/sq$i -- python3 $R/tools/mh_probe.py $SW 1000000 $V > $O/sq$i.log 2>&1 || exit 1
done
cd $R
python3 tools/pmc_sq_summary.py $O/summary.json --kernel mm_solve_kernel --last 5 --command "tools/mh_probe.py $SW $V" $O/sq1 $O/sq2 $O/sq3 $O/sq4
