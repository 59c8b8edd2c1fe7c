"""The data-parallel learner path on CPU (gloo, world size 2): two ranks, each stepping its replica block (the oracle
stands in for the GPU kernel) and making examples/train_vec.py's PPO update with the gradient / advantage statistics
all-reduced, end with exactly -- bit for bit -- the policy a single process makes from both blocks; and
flow_amd.dist.ActionScatter hands every rank its rows of the learner's action batch."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "examples"))

TOTAL, K, ITERS = 6, 12, 2


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


def full_spec():
    from helpers import idm_vehicle, ring_spec
    from oracle import refsim as S
    spec = ring_spec(R=TOTAL, N=8, length=120.0, bunching=10, junction_length=0.1, horizon=1000, env=S.ENV_WAVE_ATTENUATION_PO,
                     num_rl=1, action_low=-1.0, action_high=1.0, po_max_length=120.0)
    rng = np.random.default_rng(7)
    spec["init_pos"] = np.asarray(spec["init_pos"]) + np.abs(rng.normal(0, 0.3, (TOTAL, 8)))
    spec["vehicles"] = [idm_vehicle() for _ in range(7)] + [idm_vehicle(controller=S.CTRL_RL, rl_index=0)]
    return spec


class Collector:
    """K closed-loop steps of one replica block on the oracle; the action noise of replica r, step t is a fixed table."""

    def __init__(self, spec, lo, hi):
        from oracle import refsim as S
        self.sim = S.RingOracle(shard_spec(spec, lo, hi), np.float32)
        self.obs = torch.from_numpy(self.sim.reset().astype(np.float32))
        self.noise = torch.from_numpy(np.random.default_rng(11).normal(0, 1, (ITERS, K, TOTAL, 1)).astype(np.float32))[:, :, lo:hi]

    def __call__(self, pi, it):
        obs, act, rew, done = [self.obs], [], [], []
        for t in range(K):
            with torch.no_grad():
                a = pi.mu(obs[-1]) + self.noise[it, t] * pi.log_std.exp()
            o, r, d = self.sim.step(a.numpy())
            obs.append(torch.from_numpy(o.astype(np.float32)))
            act.append(a)
            rew.append(torch.from_numpy(r.astype(np.float32)))
            done.append(torch.from_numpy(d.astype(np.uint8)))
        self.obs = obs[-1]
        return torch.stack(obs), torch.stack(act), torch.stack(rew), torch.stack(done)


def make_policy():
    import train_vec
    torch.manual_seed(5)
    pi = train_vec.GaussianPolicy(3, 1, hidden=8)
    return pi, torch.optim.Adam(pi.parameters(), lr=1e-2)


def worker(rank, world, port, out_path):
    import train_vec
    from flow_amd.dist import ActionScatter, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(TOTAL, rank, world)
    pi, opt = make_policy()
    col = Collector(full_spec(), lo, hi)
    for it in range(ITERS):
        train_vec.ppo_update(pi, opt, [col(pi, it)], epochs=2)
    # the learner-on-one-rank layout: rank 0's action batch reaches the rank that steps the rows
    glob = torch.arange(TOTAL * 2, dtype=torch.float32).reshape(TOTAL, 2)
    mine = ActionScatter(hi - lo, 2, world, torch.device("cpu"))(glob if rank == 0 else None)
    assert torch.equal(mine, glob[lo:hi])
    torch.save([p.detach().clone() for p in pi.parameters()], out_path % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_make_the_single_process_ppo_update_bit_for_bit(tmp_path):
    import train_vec
    from flow_amd.dist import shard_range
    world = 2
    out = str(tmp_path / "rank%d.pt")
    mp.start_processes(worker, args=(world, free_port(), out), nprocs=world, join=True, start_method="spawn")
    torch.set_num_threads(1)
    pi, opt = make_policy()
    cols = [Collector(full_spec(), *shard_range(TOTAL, r, world)) for r in range(world)]
    for it in range(ITERS):
        train_vec.ppo_update(pi, opt, [c(pi, it) for c in cols], epochs=2)
    ref = [p.detach() for p in pi.parameters()]
    moved = False
    fresh, _ = make_policy()
    for rank in range(world):
        got = torch.load(out % rank)
        for a, b, c in zip(got, ref, fresh.parameters()):
            assert torch.equal(a, b)
            moved = moved or not torch.equal(a, c.detach())
    assert moved                                          # (the update did change the policy)
