#!/bin/bash
# the list phase of round 3 (profiles/r03_list_phase_experiment.patch) at 128 VGPRs (amdgpu_waves_per_eu 4 on the default-mode solve
# kernels) against build/ab/head: MM parity tests, then the headline bench alternating, then the sweep profile of one run
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r04list; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu > $O/pytest.log 2>&1 || { grep -E '^(FAILED|ERROR|E  )' $O/pytest.log | head -20; tail -2 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for rep in 1 2 3; do
  for lib in new base; do
    if [ $lib = base ]; then export SMC_HIP_LIB=$R/build/ab/head/libsmc_hip.so; else unset SMC_HIP_LIB; fi
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${lib}_$rep.json 2> $O/bench_${lib}_$rep.err || { tail -5 $O/bench_${lib}_$rep.err; exit 1; }
    python3 -c "
import json; d=json.loads(open('$O/bench_${lib}_$rep.json').read().strip().splitlines()[-1]); print('$lib: ms_per_step %.2f' % d['ms_per_step'], {k: round(v['ms']/20,2) for k,v in d['kernel_ms'].items() if k in ('loglik','mh','solve')}, 'attempts/run %.4g' % (d.get('rk_attempts_per_run') or 0))"
  done
done
