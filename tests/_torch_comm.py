"""torch.distributed (gloo) communicator on CPU tensors - test infrastructure: lets the particle-sharded driver logic run
with world_size 2 on a machine without GPUs (tests/test_multirank_gloo.py).  Not part of the product: the product's
multi-GPU communicator is the engine's own RCCL one (comm.RcclComm)."""
import numpy as np


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
