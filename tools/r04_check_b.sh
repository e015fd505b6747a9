#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04b
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "enqueued or pinned or resample or full_run or loopback or device_side" > $O/pytest.log 2>&1 || { grep -E '^(FAILED|ERROR|E  )' $O/pytest.log | head -40; }
tail -2 $O/pytest.log
python3 - <<'PY' 2>&1 | tee $O/download_times.txt
import time, numpy as np, __graft_entry__ as e
pkg = e.load_package()
n = 1_000_000
eng = pkg.HipEngine(n, 3)
eng.sample_prior_device(1, 0) if False else None
for pinned in (False, True, False, True):
    ts = []
    for rep in range(5):
        eng.synchronize(); t0 = time.perf_counter()
        a = eng.download_particles(pkg.SMC_SET_PRED, pinned=pinned); b = eng.download_lk(pkg.SMC_SET_PRED, pinned=pinned)
        ts.append(time.perf_counter() - t0)
        if rep < 4: del a, b
    print("pinned" if pinned else "pageable", "download of 1e6 particles + lk: ms", [round(1e3 * t, 2) for t in ts])
PY
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$i.json 2> $O/bench_$i.err || { tail -5 $O/bench_$i.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$O/bench_$i.json').read().strip().splitlines()[-1]); print('bench $i: ms_per_step %.2f' % d['ms_per_step'], 'value %.4g' % d['value'], 'mh syncs', d['mh_loop_synchronisations'], {k: round(v['ms']/20,2) for k,v in d['kernel_ms'].items()})"
done
