#!/bin/bash
# Round 5: the index-ordered pass of the solve scheduler as chunks of ONE block x kChunk/64 experiments (solve_sched.h: kExPerChunk)
# against the experiment-major order of rounds 1-4 (-DSMC_EX_PER_CHUNK=1) and larger chunks: time (steady state, whole runs) and
# the solve kernel's HBM traffic (FETCH_SIZE / WRITE_SIZE passes), alternating on one box.   tools/ab_build.sh exmajor -DSMC_EX_PER_CHUNK=1 ; ... chunk192 -DSMC_CHUNK=192 ; ... chunk384 -DSMC_CHUNK=384
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05l; mkdir -p $O; cd $R
LIBS="tree exmajor chunk192 chunk384"
AB_TAIL=1 tools/ab_run.sh gpurun_out/r05l/steady 3 "python3 tools/steady_state.py 1000000 1" $LIBS
AB_TAIL=1 tools/ab_run.sh gpurun_out/r05l/bench 2 "python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra | python3 -c \"import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['steady_state']['solve_kernel_ms_per_sweep'], d['kernel_ms']['solve'], d['kernel_ms']['loglik'])\"" $LIBS
cd /tmp && export TMPDIR=/tmp
for lib in $LIBS; do
  if [ $lib = tree ]; then unset SMC_HIP_LIB; else export SMC_HIP_LIB=$R/build/ab/$lib/libsmc_hip.so; fi
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$lib -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> $O/pmc_fetch_$lib.err
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$lib -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> $O/pmc_write_$lib.err
  echo "$lib: $(cd $R && python3 tools/pmc_summary.py $O/pmc_fetch_$lib $O/pmc_write_$lib $O/pmc_summary_$lib.json --particles-per-gpu 1000000 --command bench | grep mm_solve)" | tee -a $O/traffic.log
done
