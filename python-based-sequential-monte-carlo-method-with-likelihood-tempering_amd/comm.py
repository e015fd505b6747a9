"""Communicators for the particle-sharded path.

The driver only needs four small host-side collectives (they carry a handful of float64/int64
scalars per call) plus the particle exchange after resampling, which the engine performs itself
(RCCL send/recv between device buffers inside smc_resample_phase3).

  SingleComm      one rank, no communication
  RcclComm        the engine's own RCCL communicator (ncclAllReduce / ncclAllGather over xGMI) - the
                  product path for N > 1 GPUs
Host-side communicators used only by the tests (threads on one GPU, gloo on CPUs) live in tests/; they lack
`on_device`, so the driver composes the engine's *_local calls with their reductions.
"""
from __future__ import annotations

import numpy as np


class SingleComm:
    rank, size = 0, 1
    on_device = True      # nothing to reduce: the engine's *_global entry points are the whole story

    def allreduce_sum(self, x):
        return np.array(x, dtype=np.float64, copy=True)

    def allreduce_max(self, x):
        return np.array(x, dtype=np.float64, copy=True)

    def allreduce_sum_i64(self, x):
        return np.array(x, dtype=np.int64, copy=True)

    def allgather(self, x):
        return np.array(x, dtype=np.float64, copy=True)[None]

    def allgather_i64(self, x):
        return np.array(x, dtype=np.int64, copy=True)[None]

    def barrier(self):
        pass


class RcclComm:
    """Collectives of the HipEngine's RCCL communicator.

    bootstrap(rank0_bytes_or_None) -> bytes must broadcast rank 0's 128-byte unique id to every rank
    (bench.py does this through a file in a directory the ranks share; any out-of-band channel works).
    """

    on_device = True      # reductions run inside the engine: RCCL in place on its stream (driver._on_device)

    def __init__(self, engine, rank: int, size: int, bootstrap):
        self.engine, self.rank, self.size = engine, int(rank), int(size)
        uid = engine.comm_get_unique_id() if rank == 0 else None
        uid = bootstrap(uid)
        engine.comm_init(uid, rank, size)

    def allreduce_sum(self, x):
        return self.engine.comm_allreduce_sum_f64(np.atleast_1d(x))

    def allreduce_max(self, x):
        return self.engine.comm_allreduce_max_f64(np.atleast_1d(x))

    def allreduce_sum_i64(self, x):
        return self.engine.comm_allreduce_sum_i64(np.atleast_1d(x))

    def allgather(self, x):
        return self.engine.comm_allgather_f64(np.atleast_1d(x))

    def allgather_i64(self, x):
        return self.engine.comm_allgather_i64(np.atleast_1d(x))

    def barrier(self):
        self.engine.comm_barrier()
