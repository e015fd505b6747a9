"""TEST INFRASTRUCTURE: an in-process communicator for W ranks running as threads (one HipEngine context per
thread on ONE device).  Same interface as smc_lt_amd.comm.*; used to rehearse the multi-rank path with the
real kernels on a one-GPU box (RCCL refuses two ranks on one device)."""
import threading

import numpy as np


class ThreadWorld:
    def __init__(self, size):
        self.size = size
        self.barrier = threading.Barrier(size)
        self.slots = [None] * size

    def comm(self, rank):
        return ThreadComm(self, rank)


class ThreadComm:
    def __init__(self, world, rank):
        self.w, self.rank, self.size = world, rank, world.size

    def _exchange(self, x):
        self.w.slots[self.rank] = x
        self.w.barrier.wait()
        out = [np.array(s, copy=True) for s in self.w.slots]
        self.w.barrier.wait()
        return out

    def allreduce_sum(self, x):
        parts = self._exchange(np.atleast_1d(np.asarray(x, dtype=np.float64)))
        s = parts[0].copy()
        for p in parts[1:]:
            s = s + p
        return s

    def allreduce_max(self, x):
        return np.max(np.stack(self._exchange(np.atleast_1d(np.asarray(x, dtype=np.float64)))), axis=0)

    def allreduce_sum_i64(self, x):
        return np.sum(np.stack(self._exchange(np.atleast_1d(np.asarray(x, dtype=np.int64)))), axis=0)

    def allgather(self, x):
        return np.stack(self._exchange(np.atleast_1d(np.asarray(x, dtype=np.float64))))

    def allgather_i64(self, x):
        return np.stack(self._exchange(np.atleast_1d(np.asarray(x, dtype=np.int64))))

    def barrier(self):
        self.w.barrier.wait()


class PeerComm(ThreadComm):
    """The same threads, but with the collectives of the hot path INSIDE the engines (HipEngine.debug_peer_collectives): the
    driver then takes the *_global / fused / batched entry points - the world > 1 branches a run over RCCL takes - and only
    the few host-side reductions of run_smc (failure counts, barriers) go through the Python slots."""
    on_device = True
