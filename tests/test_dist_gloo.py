"""N > 1 path on CPU: two gloo ranks each own a contiguous replica shard, step it (here with
the oracle standing in for the GPU kernel -- tests may use it), gather packed
observation/reward/done with flow_amd.dist.ObservationGather, and every rank must hold
exactly what a single process over all replicas produces."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def shard_spec(spec, lo, hi):
    sub = dict(spec)
    sub["num_replicas"] = hi - lo
    sub["init_pos"] = np.asarray(spec["init_pos"])[lo:hi]
    sub["ring_length"] = np.asarray(spec["ring_length"])[lo:hi]
    return sub


def worker(rank, world, port, total, steps, out_path):
    from helpers import ring_spec
    from oracle import refsim as S
    from flow_amd.dist import ObservationGather, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = ring_spec(R=total, N=9, length=120.0, bunching=0, horizon=steps)
    rng = np.random.default_rng(4)
    spec["init_pos"] = np.asarray(spec["init_pos"]) + np.abs(rng.normal(0, 0.3, (total, 9)))
    lo, hi = shard_range(total, rank, world)
    sim = S.RingOracle(shard_spec(spec, lo, hi), np.float32)
    sim.reset()
    for _ in range(steps):
        obs, rew, done = sim.step(None)
    g = ObservationGather(hi - lo, obs.shape[1], world, torch.device("cpu"))
    o, r, d = g(torch.from_numpy(obs.astype(np.float32)), torch.from_numpy(rew.astype(np.float32)),
                torch.from_numpy(done.astype(np.uint8)))
    np.savez(out_path % rank, obs=o.numpy(), rew=r.numpy(), done=d.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_everything():
    from flow_amd.dist import shard_range
    for total, world in ((4096, 8), (10, 3), (7, 2), (5, 8)):
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_two_rank_gather_equals_single_process(tmp_path):
    from helpers import ring_spec
    from oracle import refsim as S
    total, steps, world = 12, 25, 2
    out = str(tmp_path / "rank%d.npz")
    mp.start_processes(worker, args=(world, free_port(), total, steps, out), nprocs=world, join=True,
                       start_method="spawn")
    spec = ring_spec(R=total, N=9, length=120.0, bunching=0, horizon=steps)
    rng = np.random.default_rng(4)
    spec["init_pos"] = np.asarray(spec["init_pos"]) + np.abs(rng.normal(0, 0.3, (total, 9)))
    ref = S.RingOracle(spec, np.float32)
    ref.reset()
    for _ in range(steps):
        obs, rew, done = ref.step(None)
    for rank in range(world):
        got = np.load(out % rank)
        np.testing.assert_array_equal(got["obs"], obs.astype(np.float32))
        np.testing.assert_array_equal(got["rew"], rew.astype(np.float32))
        np.testing.assert_array_equal(got["done"], done)
