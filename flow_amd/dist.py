"""Multi-GPU layer: replicas shard embarrassingly (one process per GPU, each with its own
FlowSim handle); the only exchange is an RCCL all-gather of the batched observation /
reward / done to the learner.  The reference has no collective at all:
its workers return sample batches through the Ray object store (examples/train.py:149).

Messages are small (4096 x 44 floats = 720 KB per rank), i.e. latency-bound on xGMI, so
the three tensors travel in ONE all-gather of a packed [R, obs_dim + 2] float32 buffer.

Two learner layouts:
  * a learner on ONE rank: ObservationGather brings the batch to it, ActionScatter hands every rank the rows of
    `actions [R_total, n_rl]` that belong to its replica block;
  * a data-parallel learner (examples/train_vec.py --gpus N): the trajectory stays on the rank that produced it, the
    policy is replicated, and what crosses xGMI per update is the flat gradient (allreduce_gradients: a few KB for the
    reference's fcnet [32, 32, 32]) plus three scalars for the global advantage statistics (allreduce_sum).
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


class ActionScatter(object):
    """The way back: rank ``src`` holds the actions of ALL replicas, ``[W * R, n_rl]`` in global replica order, every
    rank receives the ``[R, n_rl]`` block of the replicas it steps ("one broadcast/scatter of actions
    [R, n_rl] back").  nccl (= RCCL) on device tensors, gloo on CPU tensors."""

    def __init__(self, replicas_per_rank, act_dim, world_size, device, src=0, group=None):
        self.R, self.A, self.W, self.src, self.group = int(replicas_per_rank), int(act_dim), int(world_size), int(src), group
        self.recv = torch.empty((self.R, self.A), dtype=torch.float32, device=device)
        self._work = None

    def launch(self, actions_global=None):
        """Start the scatter; ``actions_global`` is read on rank ``src`` only (the others pass None)."""
        self.wait()
        chunks = None
        if dist.get_rank(self.group) == self.src:
            a = actions_global.reshape(self.W * self.R, self.A).to(torch.float32)
            chunks = [a[r * self.R:(r + 1) * self.R].contiguous() for r in range(self.W)]
        self._work = dist.scatter(self.recv, chunks, src=self.src, group=self.group, async_op=True)

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None

    def result(self):
        self.wait()
        return self.recv

    def __call__(self, actions_global=None):
        self.launch(actions_global)
        return self.result()


def allreduce_sum(t, group=None):
    """Sum of ``t`` over the ranks, in place (identity without an initialised process group)."""
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def allreduce_gradients(params, group=None):
    """Sum the gradients of ``params`` over the ranks as ONE flat all-reduce (a few KB: latency-bound, so one message)."""
    params = [p for p in params if p.grad is not None]
    if not params or not (dist.is_available() and dist.is_initialized()):
        return
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for p in params:
        n = p.grad.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n
