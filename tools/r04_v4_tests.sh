#!/bin/bash
# K8 v4 as the default: the whole -m gpu suite + smoke on one box
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04v4tests
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests/ -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?
tail -3 $O/pytest_gpu.log; grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head -20
python -c "import __graft_entry__ as e; e.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
exit $rc
