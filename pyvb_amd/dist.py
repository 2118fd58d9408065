"""Host-side plumbing for multi-GPU runs: one process per GPU, replicates sharded contiguously.

The data path has exactly one exchange, the all-reduce of the six lower-bound parts; on GPUs it
runs inside libpyvb_hip.so over RCCL (pyvb_lds_comm_init / pyvb_lds_elbo_total).  This module
only provides the rendezvous around it -- barrier, max over ranks for timing, and the broadcast
of the RCCL unique id -- over torch.distributed's gloo backend, and the same all-reduce on gloo
for CPU tests of the sharding logic.  torch is imported only when world > 1.
"""
import os

import numpy as np

__all__ = ["init", "shard_range", "LocalComm", "GlooComm"]


def shard_range(n_total, rank, world):
    """Contiguous partition of n_total replicates: ranks 0..r-1 get one extra when it does not divide."""
    base, extra = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class LocalComm(object):
    rank, world = 0, 1

    def barrier(self):
        pass

    def max_float(self, x):
        return float(x)

    def broadcast_bytes(self, b):
        return b

    def allreduce_sum(self, a):
        return np.asarray(a, dtype=np.float64).copy()

    def close(self):
        pass


class GlooComm(object):
    def __init__(self, world, rank):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if not dist.is_initialized():
            dist.init_process_group("gloo", rank=rank, world_size=world)
        self.rank, self.world = rank, world

    def barrier(self):
        self._dist.barrier()

    def max_float(self, x):
        t = self._torch.tensor([float(x)], dtype=self._torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t[0])

    def broadcast_bytes(self, b):
        box = [b]
        self._dist.broadcast_object_list(box, src=0)
        return box[0]

    def allreduce_sum(self, a):
        t = self._torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return t.numpy()

    def close(self):
        if self._dist.is_initialized():
            self._dist.destroy_process_group()


def init(world=None, rank=None):
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
    rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
    return LocalComm() if world <= 1 else GlooComm(world, rank)
