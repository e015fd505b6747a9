#!/bin/bash
# A/B on one box: cost order through gathers (build/ab/gather = commit b0 with order[] gathers) against the sorted staging copy (working tree)
mkdir -p gpurun_out/sorted
for rep in 1 2; do
for lib in default gather; do
  if [ $lib = default ]; then unset SMC_HIP_LIB; else export SMC_HIP_LIB=$PWD/build/ab/$lib/libsmc_hip.so; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/sorted/bench_${lib}_$rep.json 2>gpurun_out/sorted/err.log
done
done
unset SMC_HIP_LIB
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/sorted/bench_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f"{f.split('/')[-1]:28s} ms_per_step {d['ms_per_step']:7.2f}  mh avg {d['roofline'].get('mh_sweep_avg_ms', 0):.3f}  value {d['value']:.3e}")
P
R=$PWD; O=$R/gpurun_out/sorted
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_fetch.err || exit 1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_write.err || exit 1
cd $R
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/summary.json --particles-per-gpu 1000000 --command quick
for k in 3 8 13 25; do python tools/mh_probe.py $k | sed -n 2,3p; done
