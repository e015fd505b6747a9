"""Communicators for the particle-sharded path.

The driver only needs four small host-side collectives (they carry a handful of float64/int64
scalars per call) plus the particle exchange after resampling, which the engine performs itself
(RCCL send/recv between device buffers inside smc_resample_phase3).

  SingleComm      one rank, no communication
  RcclComm        the engine's own RCCL communicator (ncclAllReduce / ncclAllGather over xGMI) - the
                  product path for N > 1 GPUs
  TorchDistComm   torch.distributed on CPU tensors (gloo) - lets the sharded driver logic be tested
                  with world_size 2 on a machine without GPUs; never selected automatically
"""
from __future__ import annotations

import numpy as np


class SingleComm:
    rank, size = 0, 1

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
    (bench.py does this with torch.distributed's store; any out-of-band channel works).
    """

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


class TorchDistComm:
    """torch.distributed (already initialised by the caller, e.g. gloo) on CPU tensors."""

    def __init__(self):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.size = dist.get_rank(), dist.get_world_size()

    def _allreduce(self, x, dtype, op):
        t = self.torch.from_numpy(np.array(np.atleast_1d(x), dtype=dtype, copy=True))
        self.dist.all_reduce(t, op=op)
        return t.numpy()

    def allreduce_sum(self, x):
        return self._allreduce(x, np.float64, self.dist.ReduceOp.SUM)

    def allreduce_max(self, x):
        return self._allreduce(x, np.float64, self.dist.ReduceOp.MAX)

    def allreduce_sum_i64(self, x):
        return self._allreduce(x, np.int64, self.dist.ReduceOp.SUM)

    def _allgather(self, x, dtype):
        t = self.torch.from_numpy(np.array(np.atleast_1d(x), dtype=dtype, copy=True))
        outs = [self.torch.empty_like(t) for _ in range(self.size)]
        self.dist.all_gather(outs, t)
        return np.stack([o.numpy() for o in outs])

    def allgather(self, x):
        return self._allgather(x, np.float64)

    def allgather_i64(self, x):
        return self._allgather(x, np.int64)

    def barrier(self):
        self.dist.barrier()
