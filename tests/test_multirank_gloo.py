"""N > 1 path on CPU: world_size-2 gloo runs of the PRODUCT driver (run_smc) over a CPU test-double engine with a gloo
communicator (tests/_torch_comm.py) - once with the reductions composed by the driver (host-side communicator) and once
through the engine's *_global entry points (the shape of the product's RCCL path) - compared with the single-process oracle run on the same seed.  Validates the
sharded control logic: global max / sums, residual prefix across ranks, output-slot bases, particle
exchange plan, MH loop control, identical random streams on every rank."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, seed, q, engine_side):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import __graft_entry__ as g
    from _cpu_engine import EngineSideComm, OracleEngine
    from _torch_comm import TorchDistComm
    pkg = g.load_package()
    O = g.load_oracle()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = O.MMData.load()
        s = pkg.SMCSettings(n_particle=n, seed=seed)
        eng = OracleEngine(O, data, s.priors, n // world, n, rank, world, dist)
        comm = EngineSideComm(TorchDistComm()) if engine_side else TorchDistComm()
        out = pkg.run_smc(eng, s, comm=comm, rng="numpy", verbose=False)
        q.put((rank, out["p_pred"], out["lk"], [r["gamma_new"] for r in out["records"]],
               [r["n_accept"] for r in out["records"]], [r["last_j"] for r in out["records"]], out["logZ"],
               [r["n_offspring"] for r in out["records"]], getattr(eng, "n_deferred", 0), out["step"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,seed,engine_side", [(128, 20250205, False), (200, 3, False), (128, 20250205, True)])
def test_two_rank_gloo_run_equals_single_process_oracle(O, data, n, seed, engine_side):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, seed, q, engine_side)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = sorted([q.get(timeout=180) for _ in procs], key=lambda x: x[0])
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
    finally:                       # a rank that raised leaves its peer waiting in a collective: end both
        for p in procs:
            if p.is_alive():
                p.terminate()
    ref = O.run_smc(data, O.SMCSettings(n_particle=n), seed=seed, record_mh=False)
    p_all = np.concatenate([r[1] for r in res])
    lk_all = np.concatenate([r[2] for r in res])
    for r in res:                                   # every rank saw the same schedule
        assert r[3] == [x.gamma_new for x in ref["records"]]
        assert r[4] == [x.n_accept for x in ref["records"]]
        assert r[5] == [x.last_j for x in ref["records"]]
        assert all(o == n for o in r[7])
        assert abs(r[6] - ref["logZ"]) < 1e-9 * abs(ref["logZ"])
        # engine-side reductions: every resampling went through the deferred form (enqueue, numbers read after the Metropolis
        # loop); a host-side communicator composes the three phases itself and defers nothing
        assert r[8] == (r[9] if engine_side else 0)
    assert np.abs(p_all - ref["p_pred"]).max() < 1e-9
    assert np.max(np.abs(lk_all - ref["lk"]) / np.maximum(1, np.abs(ref["lk"]))) < 1e-9


def test_single_rank_oracle_engine_equals_oracle(O, data):
    """Same driver, world 1, no communicator: the product loop over the test double reproduces the oracle run."""
    import __graft_entry__ as g
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _cpu_engine import OracleEngine
    pkg = g.load_package()
    s = pkg.SMCSettings(n_particle=300, seed=11)
    eng = OracleEngine(O, data, s.priors, 300, 300, 0, 1)
    out = pkg.run_smc(eng, s, rng="numpy", verbose=False)
    ref = O.run_smc(data, O.SMCSettings(n_particle=300), seed=11, record_mh=False)
    assert [r["gamma_new"] for r in out["records"]] == [x.gamma_new for x in ref["records"]]
    assert np.abs(out["p_pred"] - ref["p_pred"]).max() < 1e-9
