#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04soak
mkdir -p $O
cd $R
timeout -k 10 1000 python3 tests/soak_features.py 150 device > $O/soak_device.log 2>&1; rc=$?
tail -3 $O/soak_device.log
timeout -k 10 150 python3 tests/soak_features.py 12 numpy > $O/soak_numpy.log 2>&1; rc2=$?
tail -2 $O/soak_numpy.log
exit $((rc + rc2))
