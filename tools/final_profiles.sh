#!/bin/bash
# Final profile collection of a round on ONE box (gpurun --timeout 1200 -- 'bash tools/final_profiles.sh'): every step appends to
# gpurun_out/final/progress.log (the box's watchdog sees the run is alive); tools/collect_profiles.py condenses the result
# into profiles/rNN_*.  rocprofv3: program directly after `--`, counters in their own passes (--kernel-trace only).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
log() { echo "$(date +%T) $*" >> $O/progress.log; }
set -e
log "bench 20/5 (clean)"
timeout -k 10 300 python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_mm.json 2> $O/bench_mm.err
log "kernel stats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_mm_under_rocprof.json 2> $O/stats.err
log "pmc fetch"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> $O/pmc_fetch.err
log "pmc write"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> $O/pmc_write.err
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64" "SQ_WAVES SQ_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS"; do
  i=$((i+1))
  log "sq pass $i (steady state)"
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/sq$i -- python3 $R/tools/steady_state.py 1000000 1 > $O/sq$i.log 2>&1
  log "sq pass $i (whole run)"
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/sqrun$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $O/sqrun$i.log 2>&1
done
log "methanation N = 1024 under rocprof"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/meth_stats -o run -- python3 $R/bench.py --workload methanation --particles-per-gpu 1024 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_methanation_n1024_under_rocprof.json 2> $O/meth_stats.err
# round 4: counter passes on K8 (FETCH / WRITE and six SQ sets), the FP64 FMA probe
K8CMD="python3 $R/bench.py --workload methanation --particles-per-gpu 1024 --steps 1 --warmup 0 --no-cpu-baseline"
log "k8 pmc fetch"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/k8_pmc_fetch -o run -- $K8CMD > /dev/null 2> $O/k8_pmc_fetch.err
log "k8 pmc write"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/k8_pmc_write -o run -- $K8CMD > /dev/null 2> $O/k8_pmc_write.err
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64" "SQ_WAVES SQ_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_INT32 SQ_INSTS_VMEM"; do
  i=$((i+1))
  log "k8 sq pass $i"
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/k8_sq$i -- $K8CMD > $O/k8_sq$i.log 2>&1
done
log "fp64 probe"
$R/tools/fp64_peak > $O/fp64_fma_peak.json 2> $O/fp64.err
cd $R
log "parity arithmetic (--exact)"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --exact > $O/bench_mm_exact.json 2> $O/exact.err
log "round-3 host loop (A/B)"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --mh-batch 0 --no-defer-resample --no-pinned-results > $O/bench_mm_r3_host_loop.json 2> $O/r3loop.err
log "no early reject"
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-early-reject > $O/bench_mm_no_early_reject.json 2> $O/ner.err
log "1e7"
timeout -k 10 300 python3 bench.py --particles-per-gpu 10000000 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_mm_n1e7.json 2> $O/n1e7.err
log "1e8"
timeout -k 10 400 python3 bench.py --particles-per-gpu 100000000 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_mm_n1e8.json 2> $O/n1e8.err
log "steady + tail + user model"
timeout -k 10 200 python3 tools/steady_state.py 1000000 1 > $O/steady.log 2>&1
timeout -k 10 200 python3 tools/steady_state.py 10000000 1 >> $O/steady.log 2>&1
timeout -k 10 200 python3 tools/tail_latency.py > $O/tail.log 2>&1
timeout -k 10 200 python3 tools/user_model_bench.py 200000 > $O/user_model_bench.log 2>&1
log "methanation N = 1024 line (with cpu_baseline)"
timeout -k 10 300 python3 bench.py --workload methanation --particles-per-gpu 1024 --steps 1 --warmup 0 > $O/bench_methanation_n1024.json 2> $O/meth_line.err
log "done"
