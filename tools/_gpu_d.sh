out=gpurun_out/r03d; mkdir -p $out
python -m pytest tests/test_user_model.py tests/test_gpu_parity.py -m gpu -x -q -s > $out/pytest.log 2>&1; rc=$?; grep -E "stiff band|passed|failed|Error|error" $out/pytest.log | tail -15
if [ $rc -ne 0 ]; then tail -60 $out/pytest.log; exit $rc; fi
python tools/user_model_bench.py 200000 2>&1 | tee $out/user_model_bench.log
python tools/tail_latency.py 2>&1 | tee $out/tail.log
python tools/steady_state.py 1000000 1 2>&1 | tee -a $out/tail.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_new.json 2>$out/err.log || exit 1
SMC_HIP_LIB=build/ab/r02/libsmc_hip.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_r02.json 2>>$out/err.log || exit 1
python - <<'PY'
import json
for f in ("bench_r02","bench_new"):
    d=json.load(open(f"gpurun_out/r03d/{f}.json")); k=d["kernel_ms"]
    print(f"{f:12s} ms_per_step {d['ms_per_step']:7.2f} loglik/launch {k['loglik']['ms']/k['loglik']['launches']:6.2f} mh/launch {k['mh']['ms']/k['mh']['launches']:6.3f} steady {d['steady_state']['solve_kernel_ms_per_sweep']:.3f} ess_wall {d.get('ess_iters_per_s_wall')} ess_kernel {d.get('ess_iters_per_s')} syncs {d.get('ess_search_synchronisations')}")
PY
