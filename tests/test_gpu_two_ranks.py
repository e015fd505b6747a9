"""Two real ranks (needs two visible GPUs; skipped on a one-GPU box): `bench.py --gpus 2` through its own launcher, i.e. the
in-place ncclAllReduce / ncclAllGather of every stage and the ncclSend / ncclRecv exchange of the resampling step between
two processes.  With the device RNG every draw is keyed by the GLOBAL particle index, so the sharded run must follow the
one-rank run of the same global population: same tempering schedule length, evidence and posterior mean up to the order of
the floating-point reductions."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _visible_gpus():
    code = ("import ctypes\n"
            "try:\n"
            "    l = ctypes.CDLL('libamdhip64.so'); n = ctypes.c_int(0)\n"
            "    print(n.value if l.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0 else 0)\n"
            "except OSError:\n"
            "    print(0)\n")
    try:   # a child process: this one must not initialise HIP before it forks rank processes
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        return int((out.stdout.strip().splitlines() or ["0"])[-1])
    except (subprocess.TimeoutExpired, ValueError):
        return 0


def _bench(*args):
    env = dict(os.environ, MASTER_PORT="29641")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900,
                       env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_two_rank_run_follows_the_one_rank_run():
    if _visible_gpus() < 2:
        pytest.skip("needs two visible GPUs")
    n = 100_000
    two = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--particles-per-gpu", str(n // 2), "--no-cpu-baseline", "--no-extra")
    one = _bench("--gpus", "1", "--steps", "1", "--warmup", "0", "--particles-per-gpu", str(n), "--no-cpu-baseline", "--no-extra")
    assert two["n_gpus"] == 2 and two["config"]["particles_total"] == n == one["config"]["particles_total"]
    assert two["scaling"] == "weak" and two["value"] > 0
    assert two["rccl_ranks"] == 2 and len(two["per_rank_ms_per_step"]) == 2 and one["rccl_ranks"] == 0      # RCCL itself saw both ranks
    assert two["tempering_steps_per_run"] == one["tempering_steps_per_run"]
    # the Metropolis loops ran with their control on the device: identical decisions on both ranks mean identical loop lengths,
    # and the all-reduces of the speculatively enqueued sweeps stayed matched (the run would hang or diverge otherwise)
    assert two["mutation_sweeps"] == one["mutation_sweeps"] and two["mh_loop_synchronisations"] == one["mh_loop_synchronisations"]
    assert abs(two["logZ"][0] - one["logZ"][0]) < 1e-6 * abs(one["logZ"][0]) + 1e-6
    for a, b in zip(two["posterior_mean"], one["posterior_mean"]):
        assert abs(a - b) < 1e-3 * abs(b)      # the two-rank line reports rank 0's half of the population
