#!/bin/bash
# round 4: evidence for the methanation line on ONE box - the FP64 FMA probe, the methanation bench with its CPU
# leg, rocprofv3 kernel stats, FETCH / WRITE and SQ counter passes on K8 (counters in their own passes, program directly after --).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04k8
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
log() { echo "$(date +%T) $*" >> $O/progress.log; }
set -e
N=${K8_N:-1024}
CMD="python3 $R/bench.py --workload methanation --particles-per-gpu $N --steps 1 --warmup 0 --no-cpu-baseline"
log "fp64 probe"
$R/tools/fp64_peak > $O/fp64_fma_peak.json 2> $O/fp64.err
cat $O/fp64_fma_peak.json
log "bench methanation N=$N (clean, with cpu leg)"
timeout -k 10 400 python3 $R/bench.py --workload methanation --particles-per-gpu $N --steps 1 --warmup 0 > $O/bench_methanation.json 2> $O/bench_methanation.err
log "kernel stats"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- $CMD > $O/bench_methanation_under_rocprof.json 2> $O/stats.err
log "pmc fetch"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o run -- $CMD > /dev/null 2> $O/pmc_fetch.err
log "pmc write"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o run -- $CMD > /dev/null 2> $O/pmc_write.err
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64" "SQ_WAVES SQ_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_INT32 SQ_INSTS_VMEM"; do
  i=$((i+1))
  log "sq pass $i"
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/sq$i -- $CMD > $O/sq$i.log 2>&1
done
cd $R
log "summaries"
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/k8_pmc_fetch_write_summary.json --particles-per-gpu $N --family k8 --kernel meth_particles_dae_split_kernel --command "bench.py --workload methanation --particles-per-gpu $N --steps 1 --warmup 0 --no-cpu-baseline" | tee $O/traffic.txt
python3 tools/pmc_sq_summary.py $O/k8_pmc_sq_summary.json $O/sq1 $O/sq2 $O/sq3 $O/sq4 $O/sq5 $O/sq6 --kernel meth_particles_dae_split_kernel --family k8 --command "bench.py --workload methanation --particles-per-gpu $N --steps 1 --warmup 0 --no-cpu-baseline" | tee $O/sq.txt
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
log "done"
tail -c 1500 $O/bench_methanation.json
