"""Multi-GPU sharding of the update.  The reference is single-device
(README.md:40-44); this is an MI355X-native addition: rollout slots are partitioned
contiguously over the ranks (one process per GPU), every rank runs rollout + scan + forward +
backward on its own shard with the loss means taken over the GLOBAL batch, and one RCCL
all-reduce(sum) over the flat gradient arena per update makes the clip + optimiser step
identical on every rank.  Whole-batch statistics (``norm_advs``, updater.py:97-98) need one
extra all-reduce of three doubles before the loss.  xGMI is point-to-point, so the gradient goes
out as ONE message (2.7 MB for A3CModel ... 230 MB for ConvModel) instead of per-tensor buckets.

``torch.distributed`` backend "nccl" is RCCL on ROCm; the same code runs on "gloo" with CPU
tensors, which is how the tests cover it without GPUs.
"""
import os

import torch
import torch.distributed as dist


class Shard:
    """This rank's view of a sharded update."""

    def __init__(self, rank=0, world=1, group=None):
        self.rank, self.world, self.group = rank, world, group

    @classmethod
    def from_env(cls):
        """Join the process group described by RANK / WORLD_SIZE / MASTER_* if there is one."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        force = os.environ.get("A2C_FORCE_COLLECTIVES") == "1" and "RANK" in os.environ   # test hook
        if world <= 1 and not force:
            return cls()
        if not dist.is_initialized():
            backend = os.environ.get("A2C_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            if backend == "nccl":
                torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
            dist.init_process_group(backend=backend)
        sh = cls(dist.get_rank(), dist.get_world_size())
        sh._force = force
        return sh

    @property
    def active(self):
        return self.world > 1 or getattr(self, "_force", False)

    def slot_range(self, n_rollouts_global):
        """Contiguous block of global rollout slots owned by this rank."""
        per = n_rollouts_global // self.world
        extra = n_rollouts_global % self.world
        lo = self.rank * per + min(self.rank, extra)
        return lo, lo + per + (1 if self.rank < extra else 0)

    def global_count(self, n_local):
        """Total number of samples over all ranks (python int)."""
        if not self.active:
            return int(n_local)
        t = torch.tensor([float(n_local)], dtype=torch.float64,
                         device="cuda" if dist.get_backend(self.group) == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return int(t.item())

    def allreduce_(self, t):
        """In-place sum over ranks (no-op on one rank).  Returns ``t``."""
        if self.active:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def barrier(self):
        if self.active:
            dist.barrier(group=self.group)


def moments_to_mean_std(sums, n_global):
    """(sum x, sum x^2) -> (mean, unbiased std), the host-side twin of a2c_normalize's maths
    (used by tests of the sharded statistics)."""
    s0, s1 = float(sums[0]), float(sums[1])
    mean = s0 / n_global
    var = max((s1 - n_global * mean * mean) / (n_global - 1), 0.0)
    return mean, var ** 0.5
