"""Multi-GPU layer: replicas shard embarrassingly (one process per GPU, each with its own
FlowSim handle); the only exchange is an RCCL all-gather of the batched observation /
reward / done to the learner.  The reference has no collective at all:
its workers return sample batches through the Ray object store (examples/train.py:149).

Messages are small (4096 x 44 floats = 720 KB per rank), i.e. latency-bound on xGMI, so
the three tensors travel in ONE all-gather of a packed [R, obs_dim + 2] float32 buffer.
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world_size):
    """Contiguous replica block [lo, hi) of ``rank`` (replica ids are global, so a replica's
    trajectory and noise stream do not depend on the number of GPUs)."""
    base, rem = divmod(int(total), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ObservationGather(object):
    """all_gather_into_tensor of packed (obs, reward, done) rows; works with nccl (=RCCL on
    ROCm) on device tensors and with gloo on CPU tensors (tests)."""

    def __init__(self, replicas_per_rank, obs_dim, world_size, device, group=None):
        self.R, self.D, self.W = int(replicas_per_rank), int(obs_dim), int(world_size)
        self.group = group
        self.send = torch.empty((self.R, self.D + 2), dtype=torch.float32, device=device)
        self.recv = torch.empty((self.W * self.R, self.D + 2), dtype=torch.float32, device=device)

    def __call__(self, obs, rew, done):
        """Blocking gather: pack, all-gather, return the unpacked global batch."""
        self.launch(obs, rew, done)
        return self.result()

    def launch(self, obs, rew, done):
        """Pack and start the all-gather without waiting for it (the simulator's next fragment overlaps the
        collective); the previous gather, if any, is completed first because it owns ``send`` / ``recv``."""
        self.wait()
        self.send[:, :self.D].copy_(obs)
        self.send[:, self.D].copy_(rew)
        self.send[:, self.D + 1].copy_(done)
        self._work = dist.all_gather_into_tensor(self.recv, self.send, group=self.group, async_op=True)

    def wait(self):
        work = getattr(self, "_work", None)
        if work is not None:
            work.wait()
            self._work = None

    def result(self):
        self.wait()
        return self.unpack()

    def unpack(self):
        """(obs [W*R, D], reward [W*R], done [W*R] bool) in global replica order."""
        return self.recv[:, :self.D], self.recv[:, self.D], self.recv[:, self.D + 1] > 0.5
