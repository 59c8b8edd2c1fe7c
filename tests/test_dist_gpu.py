"""N > 1 path with the HIP kernels doing the stepping: two ranks (two processes, both on cuda:0 of the one-GPU test
box, gloo for the exchange -- RCCL refuses two ranks on one device) each own a contiguous replica shard of a NOISY
ring (noise streams are keyed by the GLOBAL replica id), roll it out on the GPU, gather the packed observation /
reward / done with flow_amd.dist.ObservationGather; every rank must hold bit for bit what one handle over all
replicas produces.  Also: `bench.py --gpus N` on a box with fewer devices fails loudly instead of measuring one GPU."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def build_spec(total, steps):
    from helpers import idm_vehicle, ring_spec
    spec = ring_spec(R=total, N=10, length=130.0, bunching=0, horizon=steps, junction_length=0.1,
                     vehicles=[idm_vehicle(noise=0.3) for _ in range(10)], seed=21)
    rng = np.random.default_rng(4)
    spec["init_pos"] = np.asarray(spec["init_pos"]) + np.abs(rng.normal(0, 0.3, (total, 10)))
    return spec


def rollout(spec, steps):
    import torch
    from flow_amd.sim import FlowSim
    sim = FlowSim(spec, "f32")
    dev = torch.device("cuda", 0)
    R = sim.R
    obs = torch.empty((steps, R, sim.obs_dim), device=dev)
    rew = torch.empty((steps, R), device=dev)
    done = torch.empty((steps, R), dtype=torch.uint8, device=dev)
    sim.reset()
    torch.cuda.synchronize()
    sim.rollout_dev(steps, obs, rew, done)
    sim.sync()
    out = obs[-1].cpu(), rew[-1].cpu(), done[-1].cpu()
    sim.close()
    return out


def worker(rank, world, port, total, steps, out_path):
    import torch
    import torch.distributed as dist
    from flow_amd.dist import ObservationGather, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = build_spec(total, steps)
    lo, hi = shard_range(total, rank, world)
    sub = dict(spec, num_replicas=hi - lo, init_pos=np.asarray(spec["init_pos"])[lo:hi],
               ring_length=np.asarray(spec["ring_length"])[lo:hi], replica_offset=lo)
    o, r, d = rollout(sub, steps)
    g = ObservationGather(hi - lo, o.shape[1], world, torch.device("cpu"))
    go, gr, gd = g(o, r, d)
    np.savez(out_path % rank, obs=go.numpy(), rew=gr.numpy(), done=gd.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_gpu_ranks_reproduce_the_unsharded_run(tmp_path):
    import torch.multiprocessing as mp
    total, steps, world = 14, 45, 2                       # 7 + 7 replicas
    out = str(tmp_path / "rank%d.npz")
    mp.start_processes(worker, args=(world, free_port(), total, steps, out), nprocs=world, join=True,
                       start_method="spawn")
    o, r, d = rollout(build_spec(total, steps), steps)
    for rank in range(world):
        got = np.load(out % rank)
        np.testing.assert_array_equal(got["obs"], o.numpy())
        np.testing.assert_array_equal(got["rew"], r.numpy())
        np.testing.assert_array_equal(got["done"], d.numpy() != 0)


def test_bench_refuses_more_gpus_than_the_node_has():
    import torch
    n = torch.cuda.device_count()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "10",
                          "--warmup", "1", "--no-extras"], capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and res.stdout.strip() == ""
    assert "exposes %d GPU" % n in res.stderr
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "1",
                          "--no-extras"], capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode != 0 and "WORLD_SIZE=2" in res.stderr        # launcher and --gpus disagree
