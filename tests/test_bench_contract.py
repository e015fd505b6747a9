"""bench.py's output contract: ONE JSON line on stdout with the fields the driver reads (and the roofline object)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--workload", "methanation", "--particles-per-gpu", "64"]])
def test_bench_prints_one_json_line_with_the_contract_fields(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
           "--particles-per-gpu", "20000"] + extra
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 0 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic" and d["scaling"] == "weak"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] > 0
    assert r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    if not extra:
        # the roofline numerator is device-counted: solves that produced their outputs, launches with work
        w = r["device_counts"]
        assert w["solve_launches"] + w["noop_launches"] == r["launches_enqueued"] and r["launches"] == w["solve_launches"]
        assert 0 < w["solved_items"] <= 6 * 20000 * w["solve_launches"] and w["rk_attempts"] == r["rk_attempts"]
        assert r["frac"] <= r["frac_round4_formula"]
        # ... and the default one-GPU line carries the parity-arithmetic runs and the methanation line (item 2b)
        x, m = d["exact_mode"], d["methanation_n1024"]
        assert x["runs"] == 3 and x["ms_per_step"] > 0 and x["ratio_to_default"] > 0.5
        assert m["config"]["particles_per_gpu"] == 1024 and m["value"] > 0 and m["roofline"]["frac"] > 0
        assert m["dae_solves"] > 0 and set(m["per_solve"]) == {"bdf_steps", "newton_iters", "factorisations"}


# ---- N > 1 launch plumbing, no GPU needed -------------------------------------------------------------------------------
def _run(cmd, env=None, timeout=240):
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` - the way the driver calls the 1-GPU bench - must start the N rank processes itself (the
    parent makes no GPU call), hand RCCL's 128-byte id from rank 0 to the others, and relay rank 0's single JSON line."""
    p = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    assert json.loads(lines[0]) == {"launch_check": True, "world": 2, "ranks_seen": [0, 1], "same_id": True}


def test_bench_launcher_propagates_a_failed_rank():
    env = dict(os.environ, SMC_BENCH_FAIL_RANK="1")
    p = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--launch-check"], env=env)
    assert p.returncode == 3 and "the run is void" in p.stderr


def test_bench_rank_that_never_reaches_the_rendezvous_ends_the_run():
    """a rank that does not get through the rendezvous within the time limit exits non-zero by itself (a
    fresh exit, status 4), and the launcher then ends the ranks that are waiting for it."""
    env = dict(os.environ, SMC_BENCH_STALL_RANK="1", SMC_BENCH_INIT_TIMEOUT="3")
    p = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env, timeout=120)
    assert p.returncode == 4 and "did not finish the rendezvous within 3 s" in p.stderr and "the run is void" in p.stderr
    env = dict(os.environ, SMC_BENCH_STALL_RANK="0", SMC_BENCH_INIT_TIMEOUT="3")      # rank 0 never publishes the id
    p = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--launch-check"], env=env, timeout=120)
    assert p.returncode != 0 and "the run is void" in p.stderr


def test_bench_under_torch_distributed_run():
    """The contract's launcher for N > 1: ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    p = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
              "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["same_id"] is True


def test_bench_does_not_import_torch():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "import torch" not in src


def test_roofline_traffic_is_tied_to_kernel_revision_and_population(tmp_path, monkeypatch):
    """roofline.traffic may only come from a PMC profile of the kernel revision and population size being benchmarked
    (the round-1 bench read a stale file for any --particles-per-gpu)."""
    sys.path.insert(0, ROOT)
    import bench
    k = "void smc::mm_solve_kernel<false>"
    prof = {"meta": {"kernel_source_sha": bench.kernel_source_sha(), "particles_per_gpu": 1000, "command": "x"},
            "pmc_fetch": {k: {"avg_counter_value": 10.0}}, "pmc_write": {k: {"avg_counter_value": 5.0}}}
    (tmp_path / "profiles").mkdir()
    json.dump(prof, open(tmp_path / "profiles" / "r99_pmc_fetch_write_summary.json", "w"))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_sha", lambda root=None, family="mm": prof["meta"]["kernel_source_sha"])
    b, note = bench.measured_traffic(k, 1000)
    assert b == (2 * 10.0 + 5.0) * 1024 and "r99_pmc" in note
    assert bench.measured_traffic(k, 2000)[0] is None                       # another population size
    # the K8 line (methanation) reads its own summaries, stamped with the hash of ITS sources
    assert bench.measured_traffic("smc::meth_particles_dae_kernel", 1000, family="k8")[0] is None
    k8 = dict(prof, pmc_fetch={"smc::meth_particles_dae_kernel": {"avg_counter_value": 3.0}},
              pmc_write={"smc::meth_particles_dae_kernel": {"avg_counter_value": 1.0}})
    json.dump(k8, open(tmp_path / "profiles" / "r99_k8_pmc_fetch_write_summary.json", "w"))
    assert bench.measured_traffic("smc::meth_particles_dae_kernel", 1000, family="k8")[0] == (2 * 3.0 + 1.0) * 1024
    assert bench.measured_traffic(k, 1000)[0] == (2 * 10.0 + 5.0) * 1024   # ... and the MM line never picks a K8 file
    monkeypatch.setattr(bench, "kernel_source_sha", lambda root=None, family="mm": "somethingelse")
    b, note = bench.measured_traffic(k, 1000)
    assert b is None and "another kernel revision" in note                  # another kernel build


def test_k8_revision_hash_covers_the_dae_sources_only(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    assert bench.kernel_source_sha(family="k8") != bench.kernel_source_sha(family="mm")
    assert any("meth_dae_elem.h" in f for f in bench.K8_KERNEL_SOURCES) and any("meth_dae_split.h" in f for f in bench.K8_KERNEL_SOURCES)
    monkeypatch.delenv("SMC_K8_SPLIT", raising=False)
    assert bench.k8_kernel_name() == "meth_particles_dae_split_kernel"      # the two-wave kernel is what the launches use ...
    monkeypatch.setenv("SMC_K8_SPLIT", "0")
    assert bench.k8_kernel_name() == "meth_particles_dae_kernel"            # ... unless the one-wave kernel is asked for and not any("mm_rk45" in f for f in bench.K8_KERNEL_SOURCES)


def test_measured_fp64_peak_comes_from_a_committed_probe_log(tmp_path, monkeypatch):
    """the FP64 FMA figure in the bench line is read from the probe's log under profiles/, or is null."""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.measured_fp64_peak()[0] is None
    (tmp_path / "profiles").mkdir()
    (tmp_path / "profiles" / "r98_fp64_fma_peak.json").write_text('{"probe": "fp64_fma_peak", "device": "X", "fp64_fma_tflops": 61.25}\n')
    v, note = bench.measured_fp64_peak()
    assert v == 61.25 and "r98_fp64_fma_peak.json" in note


def test_kernel_revision_hash_ignores_comments(tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    import shutil
    pkg = os.path.basename(bench.entry.PKG_DIR)
    shutil.copytree(os.path.join(ROOT, pkg, "csrc"), tmp_path / pkg / "csrc",
                    ignore=shutil.ignore_patterns("*.o", "*.so"))
    shutil.copytree(os.path.join(ROOT, "include"), tmp_path / "include")
    a = bench.kernel_source_sha(str(tmp_path))
    assert a == bench.kernel_source_sha()
    f = tmp_path / pkg / "csrc" / "mm_rk45.h"
    f.write_text("// a new comment\n" + f.read_text() + "\n/* and\n another */\n")
    assert bench.kernel_source_sha(str(tmp_path)) == a
    f.write_text(f.read_text().replace("#define A21 (1.0 / 5)", "#define A21 (1.0 / 4)"))
    assert bench.kernel_source_sha(str(tmp_path)) != a


def test_run_watchdog_ends_a_rank_that_stops_making_progress():
    """N > 1: a rank that waits inside a collective its peers never enter would hang for ever; bench.py's run watchdog reports
    where the rank stood and leaves with status 5 (rehearsed without a GPU: the class alone, in a child process)."""
    code = ("import os, sys, time\n"
            "os.environ['SMC_BENCH_RUN_TIMEOUT'] = '2'\n"
            f"sys.path.insert(0, {ROOT!r})\n"
            "import bench\n"
            "with bench.run_watchdog(1, 2) as dog:\n"
            "    dog.beat('timed run 3')\n"
            "    time.sleep(30)\n"
            "print('not reached')\n")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert p.returncode == 5 and "rank 1 of 2 made no progress" in p.stderr and "timed run 3" in p.stderr and "not reached" not in p.stdout
    code1 = code.replace("run_watchdog(1, 2)", "run_watchdog(0, 1)").replace("time.sleep(30)", "time.sleep(4)")
    p = subprocess.run([sys.executable, "-c", code1], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0 and "not reached" in p.stdout          # one rank: no watchdog (a long single-GPU run is not a hang)


def test_k8_split_switch_is_parsed_as_the_library_parses_it(monkeypatch):
    """bench.py labels the K8 roofline by the kernel SMC_K8_SPLIT selects; the library reads the switch with C's atoi
    (csrc/meth_dae_split.h), so must bench.py - "false" or "00" are OFF for both, " 2" and "1x" are ON for both."""
    sys.path.insert(0, ROOT)
    import bench
    for v, on in [(None, True), ("0", False), ("", False), ("1", True), ("00", False), ("false", False), (" 2", True), ("-1", True), ("1x", True)]:
        if v is None:
            monkeypatch.delenv("SMC_K8_SPLIT", raising=False)
        else:
            monkeypatch.setenv("SMC_K8_SPLIT", v)
        assert bench.k8_split_enabled() is on, v
        assert bench.k8_kernel_name() == ("meth_particles_dae_split_kernel" if on else "meth_particles_dae_kernel")
