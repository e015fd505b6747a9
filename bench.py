#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X likelihood-tempered SMC engine.

Metric (BASELINE.json): particle-mutation-steps/s (+ ESS-search iterations/s) on the
Michaelis-Menten model, 1e6 particles per GPU, adaptive tempering + residual-systematic
resampling, reference default hyper-parameters (configs[1]; with --gpus N the N x 1e6 particles of
configs[2] are sharded one block per rank: weak scaling).

One "step" = one COMPLETE adaptive-tempering SMC run (prior draw -> gamma = 1) over the resident
particle population: every stage of the hot path (likelihood sweep, fused ESS search, resampling,
moments, fused Metropolis sweeps) runs inside the timed region, with device RNG so that nothing but
a few scalars per stage crosses PCIe.  value = particle-mutation-steps executed / wall time.

Launch:  python bench.py --gpus N --steps K --warmup W          (N > 1: this process spawns the N rank processes itself,
                                                                  before anything touches a GPU, and relays rank 0's line)
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                --master-port P bench.py --gpus N --steps K --warmup W      (RANK / LOCAL_RANK / WORLD_SIZE from the env)
No torch anywhere: the ranks talk through the engine's RCCL communicator; its 128-byte id travels through a file.
"""
from __future__ import annotations

import argparse
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

# K8: algorithmic FP64 flop = factorisations x 1.1e5 (block-tridiagonal LU, 51 block rows of
# 7x7) + Newton iterations x (1.5e4 block solve + 2e4 residual evaluation); all counts measured on the device
FLOP_PER_FACTORISATION, FLOP_PER_NEWTON_ITERATION = 1.1e5, 1.5e4 + 2.0e4
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X: 256 CU x 4 SIMD x 16 FP64 FMA lanes x 2 flop x 2.4 GHz (datasheet)
HBM_PEAK_GBPS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md
# algorithmic FP64 flop of one particle-mutation-step (DESIGN.md "Kernels"):
FLOP_PER_RK_ATTEMPT = 100        # 6 RHS (3 flop each) + stage sums + y_new + error estimate + controller
FLOP_PER_PARTICLE_FIXED = 6 * 40 * 14 + 240 * 4 + 6 * 12 + 50   # dense-output points, residuals, logL, accept
# ... the same fixed term split by who earns it: a (particle, experiment) solve that RAN TO THE END produced 40 dense outputs
# (14 flop each), 40 residuals (4 each) and one logL term (12); a particle of a sweep that had work is accepted or rejected (50).
# Cancelled (exact early rejection) and masked (out of support) proposals, and the speculative launches of a batch that find
# the loop ended, earn nothing
FLOP_PER_SOLVED_ITEM = 40 * 14 + 40 * 4 + 12
FLOP_PER_PARTICLE_ACCEPT = 50
HBM_BYTES_PER_PARTICLE_SOLVE = 24 + 1 + 6 * (8 + 4)  # read theta + support flag; write 6 x (sum_r2, info)


def load_mm_data():
    z = np.load(os.path.join(ROOT, "tests", "golden", "mm_data.npz"))
    return z["t"], z["P_obs"], z["S0"]


def effective_cpus():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (a container on a 256-core
    host may be limited to a 16-CPU share; sizing the pool by the mask would oversubscribe it and misreport `cores`)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(round(int(txt[0]) / int(txt[1])))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(round(q / per))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_model_name():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def cpu_stage_times(O):
    """ESS iteration and resampling loop exactly as the reference writes them (Micmem_SMC_main.py:124-134, :147-184; NumPy
    passes and a pure-Python loop, one core), at the sizes SURVEY.md 8(d) names."""
    rs = np.random.RandomState(0)
    out = {"ess_iteration_s": {}, "ess_iters_per_s": {}, "resample_loop_s": {}}
    for n in (1_000, 100_000, 1_000_000):
        lk = -np.abs(rs.standard_normal(n)) * 300
        d_lk = lk - lk.max()
        reps = 200 if n == 1_000 else 20 if n == 100_000 else 5
        t0 = time.perf_counter()
        for _ in range(reps):
            w = np.exp(d_lk * 0.01)               # :124
            sw = np.sum(w)                        # :126
            w = w / sw                            # :128
            ess = 1.0 / np.sum(w ** 2) / n        # :130-134
        dt = (time.perf_counter() - t0) / reps
        out["ess_iteration_s"][str(n)] = dt
        out["ess_iters_per_s"][str(n)] = 1.0 / dt
        assert 0.0 < ess <= 1.0
        if n <= 100_000:                          # the pure-Python loop: 2.3 s at 1e6 in the survey container - not repeated here
            p_pred, p_filt, lk1 = rs.uniform(0, 10, (n, 3)), np.zeros((n, 3)), np.zeros(n)
            t0 = time.perf_counter()
            _, n_out, _ = O.resample_python(w, 0.5, p_pred, lk, p_filt, lk1)
            out["resample_loop_s"][str(n)] = time.perf_counter() - t0
            assert n_out == n
    return out


def cpu_baseline(sample_seconds_target=4.0, n=1000):
    """the NumPy/SciPy counterpart of the reference's driver (oracle.run_smc: the statement sequence of
    Micmem_SMC_main.py:98-262 on NumPy's global legacy RNG; scipy.solve_ivp RK45 per particle x experiment on a fork pool with
    one worker per host core = the reference's one Ray task per particle, Micmem_likelihood.py:83; the resampling loop in pure
    Python as written) run END TO END at N = 1000 on the reference's seed.  Before its time counts the run must reproduce the
    reference run's schedule (tests/golden/mm_ref_run_n1000.npz: gamma bit for bit, accept counts, loop lengths, final
    particles).  The reference's own files do not travel to this box: `kind` is "port"."""
    O = entry.load_oracle()
    data = O.MMData.load()
    cores = effective_cpus()
    g = np.load(os.path.join(ROOT, "tests", "golden", "mm_ref_run_n1000.npz"))
    t0 = time.perf_counter()
    out = O.run_smc(data, O.SMCSettings(n_particle=n), seed=int(g["seed"]), loglik="scipy", n_threads=cores, record_mh=False, resample_impl="python")
    dt_run = time.perf_counter() - t0
    rec = out["records"]
    # the golden run is the reference's own at N = 1000 (its default); another size (--cpu-baseline-particles 10000: SURVEY.md 8(d)
    # also names 10^4, ~10 x the time) has no reference run to be held against and says so
    pinned = n == int(g["n_particle"])
    if pinned and not (out["step"] == int(g["final_step"]) and np.array_equal([r.gamma_new for r in rec], g["sched_gamma"])
                       and np.array_equal([r.n_accept for r in rec], g["sched_accept"]) and np.array_equal([r.last_j for r in rec], g["sched_last_j"])
                       and np.array_equal(out["p_pred"], g["final_p_pred"])):
        raise RuntimeError("cpu_baseline: the NumPy/SciPy port did not reproduce the reference run's schedule; its time is void")
    pms = out["n_mutation_sweeps"] * n
    # second figure: one likelihood pass over posterior-like particles (the steady-state regime of `steady_state`), SciPy and the
    # C restatement of the same arithmetic on all cores
    rs = np.random.RandomState(0)
    per_core_rate = 230.0  # likelihoods/s/core on the GPU box's host CPU (78 in the survey container); sizes the sample only
    m = int(max(cores * 8, min(60000, per_core_rate * cores * sample_seconds_target)))
    theta = np.array([1.2254, 0.5218, 0.02048]) + rs.standard_normal((m, 3)) * np.array([0.025, 0.0295, 0.00094])
    t0 = time.perf_counter()
    lk = O.mm_loglik_batch_scipy(theta, data, n_workers=cores)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    lk_c, _, info = O.mm_loglik_batch(np.tile(theta, (8, 1)), data, n_threads=cores)
    dt_c = time.perf_counter() - t1
    assert np.max(np.abs(lk - lk_c[:m]) / np.maximum(1, np.abs(lk))) < 1e-9
    st = cpu_stage_times(O)
    return {"value": pms / dt_run, "unit": "particle-mutation-steps/s", "cores": cores, "kind": "port", "cpu_model": cpu_model_name(),
            "sample": f"one COMPLETE adaptive-tempering run at N = {n} on the reference's seed (prior -> gamma = 1: {out['step']} tempering "
                      f"steps, {out['n_mutation_sweeps']} Metropolis sweeps + the initial sweep, {out['n_ess_iters']} ESS iterations), "
                      f"scipy.solve_ivp RK45 x 6 experiments per particle on a fork pool of {cores} workers, resampling loop in pure "
                      f"Python, {dt_run:.1f} s" + ("; schedule, accept counts and final particles equal the reference run's (asserted first)" if pinned
                                                     else "; no reference run of this size exists to hold it against"),
            "run_s": dt_run, "run_stage_s": out["stage_s"], "schedule_pinned": pinned,
            "ess_iters_per_s_in_run": out["n_ess_iters"] / out["stage_s"]["ess_search"],
            "likelihood_pass": {"value": m / dt, "unit": "particle-mutation-steps/s", "particles": m, "seconds": dt,
                                "what": "one likelihood pass (the > 98 % term of a mutation step) over posterior-like particles, same pool"},
            "c_restatement_value": 8 * m / dt_c, "c_restatement_cores": cores,
            "ess_iteration_s": st["ess_iteration_s"], "ess_iters_per_s": st["ess_iters_per_s"], "resample_loop_s": st["resample_loop_s"],
            "ess_iters_per_s_n1e6_1core": st["ess_iters_per_s"]["1000000"]}


def cpu_baseline_methanation(pkg, cond, guess, sample_seconds_target=15.0):
    """Config 4's CPU leg (missing 3).  The reference's per-particle path is cal_parallel_new -> my_model: 30 IDA
    solves of the 357-state DAE per particle, one Ray task per particle (methanation_functions.py:44-92).  Assimulo / SUNDIALS
    are absent here and on this box, so what is timed is the checker's integrator of the same class on the same equations
    (`kind: "port"`: oracle/meth_dae_oracle.c - variable-order BDF, finite-difference Jacobian, pivoted banded LU; K8's checker,
    parity-unpinned like K8 itself) on all host cores: one thread per core, each solving whole (particle, experiment) items -
    a bounded sample of posterior-like particles x the 30 experiments.  One particle-mutation-step = 30 solves."""
    import ctypes
    from concurrent.futures import ThreadPoolExecutor
    from oracle import methanation as OM                  # test infrastructure: this leg only (never the product path)
    M = pkg.methanation
    cores = effective_cpus()
    per_core_rate = 9.0                                   # solves/s/core (sizes the sample only)
    n_part = int(max(2, min(64, round(per_core_rate * cores * sample_seconds_target / 30.0))))
    rs = np.random.RandomState(0)
    L = OM._dae_lib()
    items = []
    for k in range(n_part):
        pr = M.BASEPARAMS * (1.0 + 0.02 * rs.standard_normal(len(M.BASEPARAMS)))       # posterior-like: within 2 % of the generating values
        for i in range(30):
            items.append((np.ascontiguousarray(guess[i], dtype=np.float64), OM.p0_tuple(cond, i, pr)))

    def solve(it):
        y0, p = it
        f, yf, st = np.empty(5), np.empty(357), OM.DaeStats()
        L.meth_model_one(OM._p(y0), OM._p(p), OM.S_AREA, OM.P_STP, OM._p(f), OM._p(yf), ctypes.byref(st))     # releases the GIL
        return st.status, st.steps
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        res = list(ex.map(solve, items))
    dt = time.perf_counter() - t0
    return {"value": n_part / dt, "unit": "particle-mutation-steps/s", "cores": cores, "kind": "port",
            "dae_solves_per_s": len(items) / dt, "failed_solves": int(sum(1 for r in res if r[0] != 0)),
            "bdf_steps_per_solve": float(np.mean([r[1] for r in res])),
            "sample": f"{n_part} posterior-like particles x 30 experiments = {len(items)} DAE solves (357 states, t = 0..75) by the "
                      f"checker's BDF integrator (oracle/meth_dae_oracle.c: same class, equations and tolerances as K8; the "
                      f"reference's IDA is not in the image), {cores} threads, {dt:.1f} s"}


# the translation unit of the Michaelis-Menten kernels: mm_kernels.hip and everything it includes
MM_KERNEL_SOURCES = ("csrc/mm_kernels.hip", "csrc/mm_rk45.h", "csrc/rk45_math.h", "csrc/pow_fifth_exact.h", "csrc/solve_sched.h",
                     "csrc/sweep_args.h", "csrc/philox.h", "csrc/prior.h", "csrc/smc_internal.h", "include/smc_hip.h")


# ... and of K8, the methanation DAE kernel (meth_smc.hip and everything it includes).  The launches use the two-wave kernel
# (meth_dae_split.h) unless SMC_K8_SPLIT=0 asks for the one-wave kernel of rounds 1-4 (bit-identical results, A/B runs)
def k8_split_enabled():
    """SMC_K8_SPLIT as the library reads it (csrc/meth_dae_split.h: meth_split_enabled - C's atoi: leading white space, an optional
    sign, then digits; anything that does not start like a number is 0 = off; unset = on)."""
    import re
    e = os.environ.get("SMC_K8_SPLIT")
    if e is None:
        return True
    m = re.match(r"\s*([+-]?\d+)", e)
    return bool(m) and int(m.group(1)) != 0


def k8_kernel_name():
    return "meth_particles_dae_split_kernel" if k8_split_enabled() else "meth_particles_dae_kernel"


K8_KERNEL_SOURCES = ("csrc/meth_smc.hip", "csrc/meth_dae_split.h", "csrc/meth_dae_elem.h", "csrc/meth_dae_wave.h", "csrc/meth_dae.h", "csrc/meth_model.h",
                     "csrc/sweep_args.h", "csrc/philox.h", "csrc/prior.h", "csrc/smc_internal.h", "include/smc_hip.h")


def kernel_source_sha(root=None, family="mm"):
    """sha256 over the CODE a kernel family is compiled from (MM_KERNEL_SOURCES / K8_KERNEL_SOURCES; comments and white space
    stripped, so that editing a comment does not pretend to be a new kernel): ties a profile
    (profiles/*pmc*summary.json) to a kernel revision."""
    import re
    root = root or ROOT
    h = hashlib.sha256()
    for rel in (K8_KERNEL_SOURCES if family == "k8" else MM_KERNEL_SOURCES):
        f = os.path.join(root, rel) if rel.startswith("include/") else os.path.join(root, os.path.basename(entry.PKG_DIR), rel)
        txt = open(f, encoding="utf-8", errors="replace").read()
        txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)          # block comments
        txt = re.sub(r"//[^\n]*", " ", txt)                        # line comments (no string literal of the sources holds //)
        h.update(os.path.basename(f).encode())
        h.update(" ".join(txt.split()).encode())
    return h.hexdigest()[:16]


def measured_valu_issue(n_local):
    """Vector-ALU occupancy of the solve kernel from the newest committed SQ counter summary (tools/pmc_sq_summary.py), only
    if it was taken on THIS kernel revision; otherwise (None, why).  The algorithmic-flop fraction in `roofline.frac` says how
    much of the FP64 peak the method's arithmetic uses; this says how busy the vector units are with everything the kernel
    issues (divisions as six operations, step controller, dense output, scheduling) - the number that tells whether there is
    idle issue capacity left."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq_summary.json")))
    if not files:
        return None, "no SQ counter summary under profiles/"
    f = files[-1]
    try:
        d = json.load(open(f))
        if d["meta"].get("kernel_source_sha") != kernel_source_sha():
            return None, f"{os.path.basename(f)} was taken on another kernel revision ({d['meta'].get('kernel_source_sha')})"
        keep = ("valu_busy_fraction", "valu_issue_floor_fraction", "lane_utilisation", "fp64_share_of_valu_instructions",
                "shader_clock_GHz")
        out = {k: d["derived"][k] for k in keep}
        out["workload"] = d["meta"].get("command")
        return out, f"rocprofv3 --pmc SQ_* passes ({os.path.basename(f)})"
    except (KeyError, ValueError, OSError) as e:
        return None, f"{os.path.basename(f)}: {e!r}"


def measured_fp64_peak():
    """FP64 FMA throughput a kernel of nothing but independent v_fma_f64 reaches on an MI355X of this pool, from the newest
    committed log of tools/fp64_peak.hip (profiles/rNN_fp64_fma_peak.json); (None, why) without one."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_fp64_fma_peak.json")))
    if not files:
        return None, "no profiles/r*_fp64_fma_peak.json (tools/fp64_peak.hip has not been run)"
    try:
        d = json.loads(open(files[-1]).read().strip().splitlines()[-1])
        return float(d["fp64_fma_tflops"]), f"tools/fp64_peak.hip on {d.get('device')} ({os.path.basename(files[-1])})"
    except (KeyError, ValueError, OSError, IndexError) as e:
        return None, f"{os.path.basename(files[-1])}: {e!r}"


def measured_traffic(kernel, n_local, family="mm"):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 --pmc summary (tools/pmc_summary.py), but only
    if that profile was taken on THIS kernel revision at THIS population size; otherwise (None, why)."""
    pattern = "r*_k8_pmc_fetch_write_summary.json" if family == "k8" else "r*_pmc_fetch_write_summary.json"
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", pattern)) if family == "k8" or "_k8_" not in os.path.basename(f))
    if not files:
        return None, "no PMC summary under profiles/"
    f = files[-1]
    try:
        pmc = json.load(open(f))
        meta = pmc.get("meta", {})
        if meta.get("kernel_source_sha") != kernel_source_sha(family=family):
            return None, f"{os.path.basename(f)} was taken on another kernel revision ({meta.get('kernel_source_sha')})"
        if int(meta.get("particles_per_gpu", -1)) != int(n_local):
            return None, f"{os.path.basename(f)} was taken at {meta.get('particles_per_gpu')} particles per GPU"
        # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes fetched (MI355X_MICROARCH.md,
        # HBM section; confirmed on mm_propose_kernel in round 1: 11.8 MiB reported for 24 MB read)
        kf = next(k for k in pmc["pmc_fetch"] if kernel in k)       # names as rocprofv3 prints them ("void smc::...<...>", or without "void")
        kw = next(k for k in pmc["pmc_write"] if kernel in k)
        b = (2 * pmc["pmc_fetch"][kf]["avg_counter_value"] + pmc["pmc_write"][kw]["avg_counter_value"]) * 1024
        return b, f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `{meta.get('command')}` ({os.path.basename(f)}), FETCH_SIZE doubled"
    except (KeyError, ValueError, OSError, StopIteration) as e:
        return None, f"{os.path.basename(f)}: {e!r}"


# ---- ranks ------------------------------------------------------------------------------------------------------------
def rendezvous_dir():
    """Where rank 0 leaves RCCL's unique id for the other ranks of this node.  Self-launched: a fresh directory made by the
    parent (SMC_BENCH_RDZV).  Under torch.distributed.run: derived from what all ranks of one launch share."""
    d = os.environ.get("SMC_BENCH_RDZV")
    if not d:
        tag = f"{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{os.getppid()}"
        d = os.path.join(tempfile.gettempdir(), f"smc_bench_rdzv_{tag}")
    os.makedirs(d, exist_ok=True)
    return d


def bootstrap_via_file(rank):
    """bootstrap(uid_or_None) -> uid for comm.RcclComm: rank 0 writes the 128 bytes (atomically), the others wait for them."""
    path = os.path.join(rendezvous_dir(), "rccl_unique_id.bin")

    def bootstrap(uid):
        if rank == 0:
            tmp = path + ".tmp"
            with open(tmp, "wb") as fh:
                fh.write(uid)
            os.replace(tmp, path)
            return uid
        t_end = time.time() + COMM_INIT_TIMEOUT_S
        while time.time() < t_end:
            try:
                data = open(path, "rb").read()
                if len(data) == 128:
                    return data
            except OSError:
                pass
            time.sleep(0.05)
        raise RuntimeError(f"rank {rank}: no unique id at {path} after {COMM_INIT_TIMEOUT_S:.0f} s")
    return bootstrap


COMM_INIT_TIMEOUT_S = float(os.environ.get("SMC_BENCH_INIT_TIMEOUT", "120"))


class init_watchdog:
    """A rank that has not got through the rendezvous + ncclCommInitRank within COMM_INIT_TIMEOUT_S leaves with status 4 (a
    fresh exit of this process - nothing is re-executed): a peer that never arrives would otherwise keep every other rank
    waiting inside RCCL for ever, and the launcher (launch_ranks / torch.distributed.run) then ends the others."""

    def __init__(self, rank, what):
        import threading
        self.t = threading.Timer(COMM_INIT_TIMEOUT_S, self._fire)
        self.t.daemon = True
        self.rank, self.what = rank, what

    def _fire(self):
        print(f"bench.py: rank {self.rank} did not finish {self.what} within {COMM_INIT_TIMEOUT_S:.0f} s; leaving", file=sys.stderr, flush=True)
        os._exit(4)

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *a):
        self.t.cancel()


RUN_STALL_TIMEOUT_S = float(os.environ.get("SMC_BENCH_RUN_TIMEOUT", "300"))


class run_watchdog:
    """N > 1 only: every rank must finish a complete run (73 ms of work at the default size) within RUN_STALL_TIMEOUT_S of the
    previous one.  Ranks that disagree on a collective - which the matched-by-construction all-reduces of a speculative batch
    and the planned exchange of a resampling step must never do, and which no test on real GPUs has exercised yet - would wait
    inside RCCL for ever; this rank then reports where it stood and leaves with status 5, and the launcher ends the others."""

    def __init__(self, rank, world):
        import threading
        self.rank, self.world, self.last, self.what = rank, world, time.monotonic(), "start"
        self.stop = threading.Event()
        self.t = threading.Thread(target=self._watch, daemon=True)

    def beat(self, what):
        self.last, self.what = time.monotonic(), what

    def _watch(self):
        while not self.stop.wait(1.0):
            if time.monotonic() - self.last > RUN_STALL_TIMEOUT_S:
                print(f"bench.py: rank {self.rank} of {self.world} made no progress for {RUN_STALL_TIMEOUT_S:.0f} s after '{self.what}'; leaving",
                      file=sys.stderr, flush=True)
                os._exit(5)

    def __enter__(self):
        if self.world > 1:
            self.t.start()
        return self

    def __exit__(self, *a):
        self.stop.set()


def make_comm(pkg, eng, rank, world):
    """SingleComm, or the engine's RCCL communicator."""
    if world == 1:
        return pkg.SingleComm()
    with init_watchdog(rank, "the RCCL rendezvous (smc_comm_init)"):
        return pkg.RcclComm(eng, rank, world, bootstrap_via_file(rank))


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N rank processes (same command line, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in the env) BEFORE this process makes any GPU call - it never does - wait for them, hand rank
    0's stdout through, return non-zero if any rank failed (the others are then terminated by pid)."""
    rdzv = tempfile.mkdtemp(prefix="smc_bench_rdzv_")
    procs = []
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=os.environ.get("MASTER_PORT", "29500"), SMC_BENCH_RDZV=rdzv)
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # this pool's driver only supports dmabuf IPC (RCCL needs it)
            out = None if r == 0 else sys.stderr            # one JSON line on stdout: rank 0's
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out))
        rc = 0
        live = list(procs)
        while live and rc == 0:
            time.sleep(0.1)
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0:
                    rc = code if code > 0 else 1
        if rc != 0:                                          # a failed rank leaves its peers waiting in a collective
            for p in live:
                p.terminate()
            for p in live:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            print(f"bench.py: a rank exited with status {rc}; the run is void", file=sys.stderr)
        return rc
    finally:
        shutil.rmtree(rdzv, ignore_errors=True)


def launch_check(rank, world):
    """--launch-check: the rank plumbing without a GPU (tests/test_bench_contract.py): every rank obtains the 128 bytes rank 0
    published; rank 0 prints one JSON line."""
    fail_rank = os.environ.get("SMC_BENCH_FAIL_RANK")
    if fail_rank is not None and int(fail_rank) == rank:
        sys.exit(3)
    stall_rank = os.environ.get("SMC_BENCH_STALL_RANK")     # rehearsal of a rank that hangs before the rendezvous
    with init_watchdog(rank, "the rendezvous"):
        if stall_rank is not None and int(stall_rank) == rank:
            time.sleep(3600)                                 # the watchdog ends this rank with status 4
        uid = os.urandom(128) if rank == 0 else None
        got = bootstrap_via_file(rank)(uid)
    assert len(got) == 128
    with open(os.path.join(rendezvous_dir(), f"seen_{rank}"), "w") as fh:
        fh.write(hashlib.sha256(got).hexdigest())
    if rank == 0:
        t_end = time.time() + 60
        seen = {}
        while len(seen) < world and time.time() < t_end:
            for r in range(world):
                try:
                    seen[r] = open(os.path.join(rendezvous_dir(), f"seen_{r}")).read()
                except OSError:
                    pass
            time.sleep(0.05)
        print(json.dumps({"launch_check": True, "world": world, "ranks_seen": sorted(seen),
                          "same_id": len(set(seen.values())) == 1 and len(seen) == world}), flush=True)
        if "SMC_BENCH_RDZV" not in os.environ:           # under torch.distributed.run nobody else removes it
            shutil.rmtree(rendezvous_dir(), ignore_errors=True)


def bench_methanation(args):
    """Config 4 (BASELINE.json configs[3]): methanation kinetics, 30 experiments per particle, 357-state DAE per
    experiment.  One step = one complete adaptive-tempering SMC run on one GPU.  The reference's inlet table is
    missing upstream: synthetic conditions (tests/golden/methanation_information.csv), observations = model at
    baseparams + sigma = 5 noise (SMC_methanation_main.py:89-101).  K8 is parity-unpinned."""
    n = args.particles_per_gpu if args.particles_per_gpu != 1_000_000 else 1024
    line = methanation_line(args, n, args.steps, args.warmup, cpu_seconds=15.0)
    if line is not None:
        print(json.dumps(line), flush=True)


def methanation_line(args, n, steps, warmup, cpu_seconds):
    """The methanation workload's whole bench line as a dict (rank 0; None on the other ranks): `--workload methanation` prints
    it, the default Michaelis-Menten line carries it at N = 1024 under the key `methanation_n1024`."""
    pkg = entry.load_package()
    M = pkg.methanation                          # settings-layer conversions (methanation_set_conditon.py as functions)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))   # config 5: particle-sharded
    cond = M.load_conditions(os.path.join(ROOT, "tests", "golden", "methanation_information.csv"))
    guess = M.initial_guess(cond)
    lo, hi, pos = M.prior_box()
    base = np.append(M.BASEPARAMS, M.SIGMA_TRUE)
    priors = {nm: {"dist": "uniform", "low": float(lo[i]), "high": float(hi[i])}
              for nm, i in zip(["Af", "Eaf", "Ar", "Ear", "sigma"], pos)}
    mh_batch = args.mh_batch if args.mh_batch == "auto" else int(args.mh_batch)
    s = pkg.SMCSettings(n_particle=n * world, priors=priors, mh_batch=mh_batch)
    p0 = M.p0_rows(cond, M.BASEPARAMS)
    dev = int(os.environ.get("SMC_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    flows0, _, _, _ = pkg.methanation.dae_solve_batch(p0, guess, device=dev)   # synthetic data
    np.random.seed(20250205)
    obs = flows0.T + 5.0 * np.random.standard_normal((5, 30))
    eng = pkg.HipEngine(n, 5, device=dev, n_global=n * world)
    eng.set_model_methanation(cond, guess, obs, base, pos)
    eng.set_prior(priors)
    if args.meth_sweeps > 0:
        bench_methanation_sweeps(args, pkg, eng, s, n)
        return None
    comm = make_comm(pkg, eng, rank, world)
    s.early_reject = not args.no_early_reject
    s.stiff_first = not args.no_stiff_first      # methanation: adaptive experiment order of the early-rejection sweeps
    if args.progress and rank == 0:      # a config-4 run lasts many minutes: one line per sweep on stderr (and in a file the
        t_begin = time.perf_counter()    # GPU box's watchdog can see), so that a long run does not look hung
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        prog = open(os.path.join(ROOT, "gpurun_out", "bench_methanation_progress.log"), "a")

        def wrap(name):
            f = getattr(eng, name)

            def w(*a, **k):
                out = f(*a, **k)
                chk = eng.meth_sweep_check()      # of the last sweep that ran
                line = (f"[{time.perf_counter() - t_begin:8.1f} s] {name}: gamma {a[0] if name != 'loglik' else 0.0:.6g}, "
                        f"{chk['completed_solves']} solved + {chk.get('cancelled_solves', 0)} cancelled of {chk['expected_solves']}"
                        + (f", accepted {out.get('accepted_now')}" if isinstance(out, dict) and "accepted_now" in out else "")
                        + (f", {out['n_done']} iterations" if isinstance(out, dict) and "n_done" in out else ""))
                print(line, file=sys.stderr, flush=True)
                prog.write(line + "\n")
                prog.flush()
                return out
            setattr(eng, name, w)
        for nm in ("loglik", "mh_iteration_device_rng", "mh_sweeps_device_rng"):
            wrap(nm)
    for i in range(warmup):
        pkg.run_smc(eng, s, comm=comm, rng="device", verbose=False, seed_device=900 + i)
    eng.timing_enable(True)
    eng.timing_reset()
    comm.barrier()
    eng.synchronize()
    t0 = time.perf_counter()
    outs = [pkg.run_smc(eng, s, comm=comm, rng="device", verbose=False, seed_device=1000 + i) for i in range(steps)]
    comm.barrier()
    eng.synchronize()
    elapsed = float(comm.allreduce_max([time.perf_counter() - t0])[0])
    tm = eng.timing_get()
    if rank != 0:
        comm.barrier()
        eng.close()
        return None
    pms = sum(o["stats"]["particle_mutation_steps"] for o in outs)
    sweeps = sum(o["stats"]["mutation_sweeps"] for o in outs) + steps
    solves = sum(o["stats"].get("dae_solves", 0) for o in outs)                       # device-counted: solves actually done
    cancelled = sum(o["stats"].get("dae_solves_cancelled", 0) for o in outs)          # ... and skipped by exact early rejection
    k8 = {k: sum(o["stats"].get(k, 0) for o in outs) for k in ("bdf_steps", "newton_iters", "factorisations", "failed_solves")}
    k8_flop = k8["factorisations"] * FLOP_PER_FACTORISATION + k8["newton_iters"] * FLOP_PER_NEWTON_ITERATION
    # counter traffic of K8 per launch: only from a PMC summary of THIS kernel revision at THIS population (same rule as the MM line)
    traffic, traffic_note = measured_traffic("smc::" + k8_kernel_name(), n, family="k8")
    k8_launches = max(1, tm["solve"]["launches"])
    cpu = None if (args.no_cpu_baseline or world != 1 or cpu_seconds <= 0) else cpu_baseline_methanation(pkg, cond, guess, cpu_seconds)
    line = {
        "metric": "particle-mutation-steps/sec", "value": pms / elapsed, "unit": "particle-mutation-steps/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "methanation kinetics (30 experiments x 357-state DAE per particle), adaptive tempering, "
                               "reference defaults; one step = one full SMC run; synthetic inlet table and observations",
                   "particles_per_gpu": n, "rng": "device Philox4x32-10", "parity": "K8 unpinned (no IDA in the image)",
                   "early_reject": bool(s.early_reject), "mh_batch": mh_batch},
        "dae_solves_per_s": solves / (tm["solve"]["ms"] * 1e-3), "dae_solves": solves, "dae_solves_cancelled": cancelled,
        "dae_solves_without_early_rejection": solves + cancelled,
        "per_solve": {k: k8[k] / max(1, solves) for k in ("bdf_steps", "newton_iters", "factorisations")},
        "mh_loop_synchronisations": sum(o["stats"].get("mh_syncs", 0) for o in outs),
        "tempering_steps_per_run": [o["step"] for o in outs], "mutation_sweeps": sweeps - steps,
        "posterior_mean": outs[-1]["p_pred"].mean(axis=0).tolist(), "posterior_std": outs[-1]["p_pred"].std(axis=0).tolist(),
        "logZ": [o["logZ"] for o in outs], "kernel_ms": tm,
        "roofline": {"kernel": k8_kernel_name() + " (BDF solve per workgroup: integrator wave + chain-server wave)", "bound": "mfma", "bound_note": "FP64 vector FMAs, latency-bound scans; MFMA unused (7x7 blocks)",
                     "achieved": k8_flop / (tm["solve"]["ms"] * 1e-3) / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": k8_flop / (tm["solve"]["ms"] * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                     "device_counts": k8, "flop_per_factorisation": FLOP_PER_FACTORISATION,
                     "flop_per_newton_iteration": FLOP_PER_NEWTON_ITERATION, "traffic": traffic, "traffic_note": traffic_note,
                     "kernel_source_sha": kernel_source_sha(family="k8"), "launches": k8_launches,
                     "avg_launch_ms": tm["solve"]["ms"] / k8_launches,
                     "algorithmic_flop_per_launch": k8_flop / k8_launches,
                     "peak_measured_fp64_fma_tflops": measured_fp64_peak()[0],
                     # algorithmic HBM bytes of a launch: per solve the 357-value start profile + 10 inlet numbers in, 5 flows +
                     # status out (~100 B per particle plus the shared tables) - idle, as the counters confirm
                     "hbm": {"algorithmic_bytes_per_launch": (solves / k8_launches) * (357 * 8 + 10 * 8 + 5 * 8 + 4),
                             "peak_GBps": HBM_PEAK_GBPS}},
        **({"cpu_baseline": cpu} if cpu else {}),
    }
    comm.barrier()
    eng.close()
    return line


def bench_methanation_sweeps(args, pkg, eng, s, n):
    """Config 4 at its stated size (1e5 particles = 3e6 DAE solves per sweep) does not finish a whole run inside one
    GPU call, so this mode times its two building blocks on a prior-drawn population - the hardest one of a run (stiff
    corners, failed solves): the initial likelihood sweep and `--meth-sweeps` Metropolis sweeps at a small gamma."""
    eng.sample_prior_device(4242, 0)
    eng.timing_enable(True)
    eng.timing_reset()
    t0 = time.perf_counter()
    info = eng.loglik(pkg.SMC_SET_PRED)
    eng.synchronize()
    t_lk = time.perf_counter() - t0
    k8 = eng.meth_sweep_counters()
    k8_flop = k8["factorisations"] * FLOP_PER_FACTORISATION + k8["newton_iters"] * FLOP_PER_NEWTON_ITERATION
    eng.upload_particles(pkg.SMC_SET_FILT, eng.download_particles(pkg.SMC_SET_PRED))
    eng.upload_lk(pkg.SMC_SET_FILT, eng.download_lk(pkg.SMC_SET_PRED))
    w_cov = s.w_cov()
    comm = pkg.SingleComm()
    t0 = time.perf_counter()
    acc = []
    for j in range(args.meth_sweeps):
        cov_m = pkg.proposal_cov(eng, comm, s, w_cov)
        out = eng.mh_step_device_rng(0.01, 1.0, pkg.mvn_transform(cov_m), 777, j, 0)
        acc.append(out["accepted_now"])
        print(f"sweep {j}: {time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)
    eng.synchronize()
    t_mh = time.perf_counter() - t0
    tm = eng.timing_get()
    print(json.dumps({
        "metric": "particle-mutation-steps/sec", "value": args.meth_sweeps * n / t_mh, "unit": "particle-mutation-steps/s",
        "n_gpus": 1, "steps": args.meth_sweeps, "warmup": 0, "ms_per_step": 1e3 * t_mh / args.meth_sweeps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "methanation kinetics (30 experiments x 357-state DAE per particle): Metropolis sweeps over a "
                               "prior-drawn population at gamma = 0.01; one step = one sweep", "particles_per_gpu": n,
                   "rng": "device Philox4x32-10", "parity": "K8 unpinned (no IDA in the image)"},
        "initial_likelihood_sweep_s": t_lk, "dae_solves_initial_sweep": n * 30,
        "dae_solves_per_s": n * 30 / (tm["loglik"]["ms"] * 1e-3),      # from the unmasked initial sweep
        "accepted_per_sweep": acc, "note": "proposals outside the prior box are not solved (their share is 1 - accept-eligible)",
        "kernel_ms": tm,
        "roofline": {"kernel": k8_kernel_name() + " (BDF solve per workgroup, element-layout scans)",
                     "bound": "mfma", "bound_note": "FP64 vector FMAs, latency-bound scans; MFMA unused (7x7 blocks)",
                     "achieved": k8_flop / (tm["loglik"]["ms"] * 1e-3) / 1e12,
                     "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": k8_flop / (tm["loglik"]["ms"] * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                     "device_counts_initial_sweep": k8, "flop_per_factorisation": FLOP_PER_FACTORISATION,
                     "flop_per_newton_iteration": FLOP_PER_NEWTON_ITERATION, "traffic": None},
    }), flush=True)
    eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--particles-per-gpu", type=int, default=1_000_000)
    ap.add_argument("--particles-total", type=int, default=0,
                    help="STRONG scaling: this many particles in total, split evenly over the --gpus ranks (BASELINE.json: '1->8-GPU "
                         "scaling curve for 10^6 particles'); default 0 = weak scaling with --particles-per-gpu per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-particles", type=int, default=1000,
                    help="population of the CPU baseline's end-to-end run (1000 = the reference's own run, whose schedule is asserted "
                         "first; SURVEY.md 8(d) also names 10000: ~100 s on 16 cores, nothing to assert against)")
    ap.add_argument("--workload", choices=["mm", "methanation"], default="mm",
                    help="mm = BASELINE.json's headline configuration (default); methanation = config 4 (one GPU)")
    ap.add_argument("--meth-sweeps", type=int, default=0,
                    help="methanation only: time this many Metropolis sweeps (and the initial likelihood sweep) instead "
                         "of whole runs - the mode for config 4's full size, --particles-per-gpu 100000")
    ap.add_argument("--no-early-reject", action="store_true",
                    help="A/B switch: complete every solve even when its proposal is already certain to be rejected (SMCSettings.early_reject)")
    ap.add_argument("--no-stiff-first", action="store_true",
                    help="A/B switch: hand the (particle, experiment) solves out in plain index order (SMCSettings.stiff_first)")
    ap.add_argument("--no-in-phase", action="store_true",
                    help="A/B switch: never let a wave wait for all of its lanes before a hand-out (SMCSettings.in_phase)")
    ap.add_argument("--no-cost-order", action="store_true",
                    help="A/B switch: heterogeneous Metropolis sweeps hand their solves out in index order (SMCSettings.cost_order)")
    ap.add_argument("--no-fast-tail", action="store_true",
                    help="A/B switch: lone chains run the compiled step function, not the hand-written loop (smc_set_fast_tail)")
    ap.add_argument("--exact", action="store_true",
                    help="run the PARITY arithmetic (smc_set_exact_pow(1): correctly rounded step-controller power, the mode whose "
                         "results are pinned to the reference to 1e-9 with equal RK45 step sequences) on the same device-RNG "
                         "workload: the price of bit-parity next to the default line")
    ap.add_argument("--no-defer-resample", action="store_true", help="A/B switch: resampling with its two host synchronisations (SMCSettings.defer_resample)")
    ap.add_argument("--no-pinned-results", action="store_true", help="A/B switch: final particles into pageable NumPy arrays (SMCSettings.pinned_results)")
    ap.add_argument("--mh-batch", default="auto",
                    help="Metropolis iterations enqueued per host synchronisation, loop control on the device (SMCSettings.mh_batch): "
                         "'auto' (default), an integer, or 0 = one call and one host decision per iteration (round 3's loop)")
    ap.add_argument("--progress", action="store_true",
                    help="methanation only: one line per sweep on stderr and in gpurun_out/bench_methanation_progress.log")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the two extra keys of the default one-GPU line (`exact_mode`: 3 runs in the parity arithmetic; "
                         "`methanation_n1024`: one complete methanation run at N = 1024 with its own roofline and cpu_baseline)")
    ap.add_argument("--launch-check", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:      # no launcher: become one (no GPU call in this process)
        sys.exit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.launch_check:
        return launch_check(rank, world)
    if args.workload == "methanation":
        return bench_methanation(args)
    args.gpus = world

    pkg = entry.load_package()
    strong = args.particles_total > 0
    if strong and args.particles_total % world:
        raise SystemExit(f"--particles-total {args.particles_total} is not a multiple of the {world} ranks")
    n_local = args.particles_total // world if strong else args.particles_per_gpu
    n_global = n_local * world
    t, P_obs, S0 = load_mm_data()
    mh_batch = args.mh_batch if args.mh_batch == "auto" else int(args.mh_batch)
    s = pkg.SMCSettings(n_particle=n_global, early_reject=not args.no_early_reject, stiff_first=not args.no_stiff_first, in_phase=not args.no_in_phase,
                        cost_order=not args.no_cost_order, mh_batch=mh_batch, exact_pow=True if args.exact else None,
                        defer_resample=not args.no_defer_resample, pinned_results=not args.no_pinned_results)

    # SMC_BENCH_DEVICE pins every rank to one device (rehearsing the multi-rank path on a one-GPU box)
    dev = int(os.environ.get("SMC_BENCH_DEVICE", local_rank))
    eng = pkg.HipEngine(n_local, 3, device=dev, n_global=n_global)
    eng.set_model_mm(t, P_obs, S0)
    eng.set_prior(s.priors)
    eng.set_fast_tail(not args.no_fast_tail)
    comm = make_comm(pkg, eng, rank, world)

    def one_run(i):
        return pkg.run_smc(eng, s, comm=comm, rng="device", verbose=False, seed_device=1000 + i)

    with run_watchdog(rank, world) as dog:
        for i in range(args.warmup):
            one_run(-1 - i)
            dog.beat(f"warm-up run {i}")
        eng.timing_enable(True)
        eng.timing_reset()
        comm.barrier()
        eng.synchronize()
        t0 = time.perf_counter()
        outs = []
        for i in range(args.steps):
            out = one_run(i)
            dog.beat(f"timed run {i}")
            if outs:                      # only the last run's particles are looked at below: let the earlier result arrays go (their
                outs[-1].pop("p_pred")    # page-locked buffers return to the pool and serve the next run's download)
                outs[-1].pop("lk")
            outs.append(out)
        comm.barrier()
        eng.synchronize()
        elapsed = time.perf_counter() - t0
        per_rank_s = np.asarray(comm.allgather([elapsed]), dtype=np.float64).reshape(-1)      # every rank's own clock
    elapsed = float(per_rank_s.max())
    rccl = eng.comm_info() if world > 1 else {"count": 0, "user_rank": -1, "device": -1}
    timing = eng.timing_get()
    work = eng.work_totals()                     # device-counted: solves that produced their outputs, launches that had work
    eng.timing_enable(False)

    # steady state (outside the timed region): Metropolis sweeps at gamma = 1 on the final, posterior
    # population - the regime SURVEY.md 8(d) calls "posterior-like"; no stiff stragglers left
    steady = None
    if world == 1:
        eng.timing_enable(True)
        eng.timing_reset()
        eng.upload_particles(pkg.SMC_SET_FILT, outs[-1]["p_pred"])
        eng.upload_lk(pkg.SMC_SET_FILT, outs[-1]["lk"])
        w_cov = s.w_cov()
        ts = time.perf_counter()
        n_ss = 10
        att_ss = 0
        for j in range(n_ss):                  # the fused iteration run_smc uses: moments -> factor -> propose -> solve -> accept
            att_ss += eng.mh_iteration_device_rng(1.0, 1.0, w_cov, 424242, j, 0)["rk_attempts"]
        eng.synchronize()
        dt_ss = time.perf_counter() - ts
        tm_ss = eng.timing_get()
        ss_solve_s = tm_ss["solve"]["ms"] / n_ss * 1e-3
        work_ss = eng.work_totals()
        if work_ss:    # same numerator as `roofline`: device-counted solves that produced their outputs
            ss_flop = (FLOP_PER_RK_ATTEMPT * att_ss + FLOP_PER_SOLVED_ITEM * work_ss["solved_items"]) / n_ss + FLOP_PER_PARTICLE_ACCEPT * n_local
        else:
            ss_flop = FLOP_PER_RK_ATTEMPT * att_ss / n_ss + FLOP_PER_PARTICLE_FIXED * n_local
        ss_tflops = ss_flop / ss_solve_s / 1e12
        steady = {"particle_mutation_steps_per_s_wall": n_ss * n_local / dt_ss,
                  "solve_kernel_ms_per_sweep": tm_ss["solve"]["ms"] / n_ss,
                  "sweep_ms_wall": 1e3 * dt_ss / n_ss, "sweeps": n_ss, "rk_attempts_per_sweep": att_ss / n_ss,
                  "solved_items_per_sweep": work_ss["solved_items"] / n_ss if work_ss else None,
                  "solve_kernel_tflops": ss_tflops, "solve_kernel_roofline_frac": ss_tflops / FP64_VECTOR_PEAK_TFLOPS}
        eng.timing_enable(False)

    pms = sum(o["stats"]["particle_mutation_steps"] for o in outs)          # global count
    sweeps = sum(o["stats"]["mutation_sweeps"] for o in outs)
    ess_iters = sum(o["stats"]["ess_iters"] for o in outs)
    ess_wall_s = sum(o["stats"].get("ess_search_s", 0.0) for o in outs)
    ess_syncs = sum(o["stats"].get("ess_syncs", 0) for o in outs)

    result = None
    if rank == 0:
        mh = timing["mh"]
        sv = timing["solve"]                      # the persistent RK45 solve kernel: >99 % of every sweep
        n_init = args.steps                       # one initial likelihood sweep per run also launches it
        mh_ms = mh["ms"] / max(1, mh["launches"])
        solve_ms = sv["ms"] / max(1, sv["launches"])
        # rank 0's fused-MH launches: algorithmic flop from the device-counted RK45 attempts of those launches
        att = sum(o["stats"]["rk_attempts"] for o in outs)       # all solve launches of the timed region (rank 0)
        # round 4's numerator, kept one more round next to the honest one: the fixed term for EVERY particle of EVERY enqueued launch
        flop_all_r4 = FLOP_PER_RK_ATTEMPT * att + FLOP_PER_PARTICLE_FIXED * n_local * sv["launches"]
        frac_r4 = flop_all_r4 / max(1, sv["launches"]) / (solve_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS
        # the honest one: device-counted solves that produced their outputs, launches that had work (the speculative no-op
        # launches of a batch leave numerator AND launch count; their few microseconds each stay in the time: conservative)
        launches_work = work["solve_launches"] if work else sv["launches"]
        assert work is None or work["solve_launches"] + work["noop_launches"] == sv["launches"], (work, sv)
        assert work is None or work["rk_attempts"] == att, (work, att)
        if work:
            flop_all = (FLOP_PER_RK_ATTEMPT * att + FLOP_PER_SOLVED_ITEM * work["solved_items"]
                        + FLOP_PER_PARTICLE_ACCEPT * n_local * launches_work)
        else:
            flop_all = flop_all_r4
        solve_ms = sv["ms"] / max(1, launches_work)
        flop_per_launch = flop_all / max(1, launches_work)
        ach_tflops = flop_per_launch / (solve_ms * 1e-3) / 1e12
        hbm_gbps = HBM_BYTES_PER_PARTICLE_SOLVE * n_local / (solve_ms * 1e-3) / 1e9
        ess_ms = timing["ess"]["ms"]
        # the instantiation a sweep of this size launches (csrc/mm_kernels.hip: kFastTailMaxParticles): <WRITE_PRED, EXACT, FAST>
        fast = (not args.no_fast_tail) and n_local <= 4_000_000 and not args.exact
        kernel_name = "mm_solve_kernel<false, true, false>" if args.exact else "mm_solve_kernel<false, false, %s>" % ("true" if fast else "false")
        traffic, traffic_note = measured_traffic("void smc::" + kernel_name, n_local)
        valu, valu_note = measured_valu_issue(n_local)
        ess_l = timing["ess"]["launches"]
        ess_avg_ms = ess_ms / max(1, ess_l)
        result = {
            "metric": "particle-mutation-steps/sec", "value": pms / elapsed, "unit": "particle-mutation-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            # what RCCL itself reports (ncclCommCount; 0 = one rank, no communicator) and every rank's own wall time per step
            "rccl_ranks": rccl["count"], "per_rank_ms_per_step": [1e3 * float(v) / args.steps for v in per_rank_s],
            "config": {"workload": "Michaelis-Menten (6 experiments x 40 points), adaptive tempering + "
                                   "residual-systematic resampling, reference defaults; one step = one full SMC run "
                                   "prior->gamma=1", "particles_per_gpu": n_local, "particles_total": n_global,
                       "rng": "device Philox4x32-10", "parallelism": f"particle-sharded x{world}",
                       "arithmetic": "parity mode (smc_set_exact_pow(1): correctly rounded pow(x, -0.2) in the step controller; the "
                                     "instantiation pinned to the reference)" if args.exact else
                                     "default (fast inverse fifth root in the step controller; <= 7.6e-8 from parity mode on stiff-band particles)"},
            # ESS-search half of the metric, twice: iterations / time of the ESS kernels alone (HIP events), and iterations /
            # WALL time spent in the search (max(lk) + passes + all-reduce + read-back + the host's decision) - what a
            # caller of the search actually waits for
            "ess_iters_per_s": ess_iters / (ess_ms * 1e-3) if ess_ms > 0 else None,
            "ess_iters_per_s_wall": ess_iters / ess_wall_s if ess_wall_s > 0 else None,
            "ess_iters": ess_iters, "ess_kernel_ms_total": ess_ms, "ess_search_wall_ms_total": 1e3 * ess_wall_s,
            "ess_search_synchronisations": ess_syncs,
            # host synchronisations of the Metropolis loops (round 3: one per sweep; now the loop control runs on the device and
            # the host waits once per batch of enqueued sweeps), and the enqueued sweeps that found their loop already ended
            "mh_loop_synchronisations": sum(o["stats"].get("mh_syncs", 0) for o in outs),
            "mh_speculative_noop_sweeps": sum(o["stats"].get("mh_noop_sweeps", 0) for o in outs), "mh_batch": mh_batch,
            "tempering_steps_total": sum(o["step"] for o in outs),
            "tempering_steps_per_run": [o["step"] for o in outs],
            "mutation_sweeps": sweeps, "logZ": [o["logZ"] for o in outs],
            "posterior_mean": outs[-1]["p_pred"].mean(axis=0).tolist(),
            "kernel_ms": {k: v for k, v in timing.items()},
            "steady_state": steady,
            "roofline": {"kernel": kernel_name + " (persistent RK45 solve, lane-level dynamic scheduling: csrc/solve_sched.h; lone chains: hand-written loop of csrc/mm_rk45.h)",
                         "bound": "mfma",
                         "bound_note": "compute roof: the kernel issues FP64 vector FMAs (no contraction larger than 3x3, so "
                                       "MFMA is unused); on MI355X the FP64 matrix and vector peaks coincide (78.6 TFLOP/s), "
                                       "which is the `peak` below; the HBM side is in `hbm`",
                         "achieved": ach_tflops, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach_tflops / FP64_VECTOR_PEAK_TFLOPS, "traffic": traffic, "traffic_note": traffic_note,
                         "kernel_source_sha": kernel_source_sha(), "valu_issue": valu, "valu_issue_note": valu_note,
                         "peak_measured_fp64_fma_tflops": measured_fp64_peak()[0], "peak_measured_note": measured_fp64_peak()[1],
                         "avg_launch_ms": solve_ms, "launches": launches_work, "launches_enqueued": sv["launches"],
                         "launches_noop": sv["launches"] - launches_work, "mh_sweep_avg_ms": mh_ms,
                         "algorithmic_flop_per_launch": flop_per_launch,
                         "numerator": "100 flop x device-counted RK45 attempts + 732 flop x device-counted solves that produced their "
                                      "40 outputs + 50 flop x particles of the launches that had work; no-op launches excluded from "
                                      "numerator and launch count",
                         "device_counts": work, "rk_attempts": att,
                         "frac_round4_formula": frac_r4,
                         "hbm": {"algorithmic_bytes_per_launch": HBM_BYTES_PER_PARTICLE_SOLVE * n_local,
                                 "achieved_GBps": hbm_gbps, "peak_GBps": HBM_PEAK_GBPS,
                                 "frac": hbm_gbps / HBM_PEAK_GBPS},
                         "mfma": "unused (largest contraction is 3x3)"},
            # the ESS-search half of BASELINE.json's metric has its own kernel and its own bound: one pass over lk (8 B per
            # particle) evaluates up to 16 tempering candidates = 16 exp + 32 FMA per 8 B, so the pass is exp-bound, not
            # HBM-bound (an 8 MB read would take ~1.3 us at the HBM roof)
            "ess_roofline": {"kernel": "ess_partial_kernel<K> + sum_rows_final_kernel (one fused pass for K <= 16 candidates)",
                             "bound": "FP64 exp throughput (K exp per 8 B read); HBM side reported",
                             "avg_launch_ms": ess_avg_ms, "launches": ess_l, "candidates_per_launch": ess_iters / max(1, ess_l),
                             "algorithmic_bytes_per_launch": 8 * n_local,
                             "achieved_GBps": 8 * n_local / (ess_avg_ms * 1e-3) / 1e9 if ess_avg_ms > 0 else None,
                             "peak_GBps": HBM_PEAK_GBPS,
                             "exp_per_s": (ess_iters * n_local) / (ess_ms * 1e-3) if ess_ms > 0 else None},
        }
        # What follows the headline's timed region must never cost the headline its line: each extra measurement reports its own
        # failure under its key (and on stderr) instead of raising.
        def guarded(key, fn):
            try:
                result[key] = fn()
            except Exception as ex:  # noqa: BLE001
                import traceback
                traceback.print_exc(file=sys.stderr)
                result[key] = {"error": f"{type(ex).__name__}: {ex}"}

        if world == 1 and not args.no_extra and not args.exact:
            # Two more measurements in the SAME process, after the headline's timed region: the price of the
            # parity arithmetic, and the methanation workload's line at N = 1024 - so that both are driver-run numbers
            def exact_mode():
                s_exact = pkg.SMCSettings(**{**s.__dict__, "exact_pow": True})
                one = lambda i: pkg.run_smc(eng, s_exact, comm=comm, rng="device", verbose=False, seed_device=1000 + i)   # noqa: E731
                one(-1)
                eng.synchronize()
                tx = time.perf_counter()
                ox = [one(i) for i in range(3)]
                eng.synchronize()
                dx = (time.perf_counter() - tx) / 3
                eng.set_exact_pow(False)
                return {"ms_per_step": 1e3 * dx, "runs": 3, "ratio_to_default": 1e3 * dx / result["ms_per_step"],
                        "value": sum(o["stats"]["particle_mutation_steps"] for o in ox) / (3 * dx),
                        "arithmetic": "smc_set_exact_pow(1): correctly rounded pow(x, -0.2) in the step controller, separately rounded stage "
                                      "sums - the instantiation pinned to the reference (equal RK45 step sequences, <= 1e-9 on logL)",
                        "same_seeds_as_headline_runs": [1000, 1001, 1002],
                        "tempering_steps_per_run": [o["step"] for o in ox], "logZ": [o["logZ"] for o in ox]}
            guarded("exact_mode", exact_mode)
            m_args = argparse.Namespace(**{**vars(args), "meth_sweeps": 0, "progress": False})
            guarded("methanation_n1024", lambda: methanation_line(m_args, 1024, 1, 0, cpu_seconds=0.0 if args.no_cpu_baseline else 4.0))
        if not args.no_cpu_baseline and world == 1:     # rank 0 at N = 1 only
            guarded("cpu_baseline", lambda: cpu_baseline(n=args.cpu_baseline_particles))
        print(json.dumps(result), flush=True)
    comm.barrier()
    eng.close()
    if world > 1 and rank == 0:                       # everybody is past the last collective: the id file has served
        shutil.rmtree(rendezvous_dir(), ignore_errors=True)


if __name__ == "__main__":
    main()
