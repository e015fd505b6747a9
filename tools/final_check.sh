#!/bin/bash
# the round's last look on one box: full GPU suite, then the bench line of the driver's configuration
python -m pytest tests -m gpu -x -q > gpurun_out/final_check_pytest.log 2>&1; tail -2 gpurun_out/final_check_pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/final_check_bench.json 2>/dev/null
python - <<'P'
import json
d = json.loads(open("gpurun_out/final_check_bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"])
P
