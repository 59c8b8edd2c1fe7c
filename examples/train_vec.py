"""train_vec.py -- a minimal on-device PPO loop over VecFlowEnv (the GPU counterpart of the reference's
examples/train.py:110-212, where RLlib rollout workers each drive one SUMO process).

    python examples/train_vec.py --replicas 1024 --iterations 20
    python examples/train_vec.py --replicas 4096 --gpus 8        # data-parallel: one process per GPU (RCCL)

Environment: the reference's single-agent ring experiment (examples/exp_configs/rl/singleagent/singleagent_ring.py:
WaveAttenuationPOEnv, 21 IDM humans + 1 RL vehicle on a 230..270 m ring).  A rollout fragment of K steps -- policy
forward, action sampling, Env.step of every replica, reset of finished episodes -- is ONE replay of a captured HIP
graph (VecFlowEnv.capture); observations, actions, rewards and the PPO update all stay in HBM.

--gpus N (the reference's num_workers / ray.init(num_cpus), examples/train.py:149,195): N child processes, one per GPU,
started BEFORE anything touches a GPU; rank r steps the replica block shard_range(R, r, N) (VecFlowEnv(replica_offset=):
the Philox streams are keyed by the global replica id, so the trajectories do not depend on N); the policy is replicated
and every PPO step all-reduces the flat gradient and the advantage statistics over RCCL (flow_amd/dist.py).  The update
is written over SHARDS (ppo_update): a rank holds one, and a single process holding all of them -- gradients accumulated
shard by shard -- makes the same update, which is what tests/test_train_dist_gloo.py checks bit for bit.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
import torch.nn as nn


def ring_flow_params(horizon):
    from flow_amd.controllers import ContinuousRouter, IDMController, RLController
    from flow_amd.core.params import EnvParams, InitialConfig, NetParams, SumoParams, VehicleParams
    from flow_amd.envs import WaveAttenuationPOEnv
    from flow_amd.networks import RingNetwork
    veh = VehicleParams()
    veh.add(veh_id="human", acceleration_controller=(IDMController, {"noise": 0.2}),
            routing_controller=(ContinuousRouter, {}), num_vehicles=21)
    veh.add(veh_id="rl", acceleration_controller=(RLController, {}), routing_controller=(ContinuousRouter, {}),
            num_vehicles=1)
    return dict(exp_tag="stabilizing_the_ring", env_name=WaveAttenuationPOEnv, network=RingNetwork, simulator="traci",
                sim=SumoParams(sim_step=0.1, render=False, seed=3),
                env=EnvParams(horizon=horizon, warmup_steps=0,
                              additional_params={"max_accel": 1, "max_decel": 1, "ring_length": None}),
                net=NetParams(additional_params={"length": 260, "lanes": 1, "speed_limit": 30, "resolution": 40}),
                veh=veh, initial=InitialConfig(bunching=20))


class GaussianPolicy(nn.Module):
    def __init__(self, obs_dim, act_dim, hidden=32):
        super().__init__()
        self.mu = nn.Sequential(nn.Linear(obs_dim, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh(),
                                nn.Linear(hidden, act_dim))
        self.value = nn.Sequential(nn.Linear(obs_dim, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh(),
                                   nn.Linear(hidden, 1))
        self.log_std = nn.Parameter(torch.full((act_dim,), -0.5))

    def act(self, obs):                       # what the captured graph runs every step
        with torch.no_grad():
            mu = self.mu(obs)
            return mu + torch.randn_like(mu) * self.log_std.exp()

    def logp_value(self, obs, act):
        mu = self.mu(obs)
        std = self.log_std.exp()
        logp = (-0.5 * ((act - mu) / std) ** 2 - self.log_std - 0.9189385).sum(-1)
        return logp, self.value(obs).squeeze(-1)


def gae(rew, val, done, last_val, gamma=0.999, lam=0.97):
    """Generalised advantage estimation over a [K, R] fragment.  ``done`` is the simulator's FLAG BYTE (bit 0: horizon
    reached, bit 1: collision -- include/flowsim.h), so "the episode went on" is ``done == 0``, not ``1 - done``."""
    K = rew.shape[0]
    adv = torch.zeros_like(rew)
    run = torch.zeros_like(last_val)
    nxt = last_val
    for t in range(K - 1, -1, -1):
        live = (done[t] == 0).float()
        delta = rew[t] + gamma * nxt * live - val[t]
        run = delta + gamma * lam * live * run
        adv[t] = run
        nxt = val[t]
    return adv, adv + val


def ppo_update(pi, opt, shards, epochs=4, clip=0.2, group=None):
    """One PPO update over `shards` = [(obs [K+1, R, D], act [K, R, A], rew [K, R], done [K, R])]: the shards this
    process holds (one per rank in data-parallel runs, all of them in a single process).  Every quantity that couples the
    shards is a SUM: the advantage statistics and the gradient -- summed over the local shards here, over the ranks by
    allreduce_sum / allreduce_gradients -- so N ranks with one shard each and one process with N shards agree."""
    from flow_amd.dist import allreduce_gradients, allreduce_sum
    prepared, stats = [], None
    for obs, act, rew, done in shards:
        K, R = rew.shape
        o, a = obs[:K].reshape(K * R, -1), act.reshape(K * R, -1)
        with torch.no_grad():
            logp_old, val = pi.logp_value(o, a)
            last_val = pi.value(obs[K]).squeeze(-1)
            adv, ret = gae(rew, val.view(K, R), done, last_val)
            adv = adv.reshape(-1).double()
            st = torch.stack([adv.sum(), (adv * adv).sum(), torch.tensor(float(adv.numel()), dtype=torch.float64, device=adv.device)])
            stats = st if stats is None else stats + st
        prepared.append((o, a, logp_old, adv, ret.reshape(-1)))
    stats = allreduce_sum(stats, group)
    n = stats[2]
    mean = stats[0] / n
    std = ((stats[1] - n * mean * mean) / (n - 1)).clamp_min(0).sqrt()
    params = list(pi.parameters())
    for _ in range(epochs):
        opt.zero_grad()
        for o, a, logp_old, adv, ret in prepared:
            adv_n = ((adv - mean) / (std + 1e-8)).float()
            logp, v = pi.logp_value(o, a)
            ratio = (logp - logp_old).exp()
            # the GLOBAL mean as a sum of per-shard sums
            loss = (-torch.min(ratio * adv_n, ratio.clamp(1 - clip, 1 + clip) * adv_n).sum()
                    + 0.5 * (v - ret).pow(2).sum()) / float(n)
            loss.backward()                                  # (accumulates over the local shards)
        allreduce_gradients(params, group)
        opt.step()


def train_on_device(flow_params, replicas=1024, fragment=100, iterations=20, epochs=4, lr=3e-4, seed=0, log=print,
                    rank=0, world=1, shared_agents=False):
    """PPO on R replicas of ``flow_params`` with everything in HBM: returns the mean step reward per iteration.
    ``world`` > 1: this process is rank ``rank`` of a data-parallel run (torch.distributed is initialised): it steps its
    block of the R replicas on its own GPU.
    ``shared_agents``: a multi-agent experiment whose agents share ONE policy (the reference's multi-agent ring /
    figure-eight / merge experiments map every agent to the policy 'av': examples/exp_configs/rl/multiagent/*.py
    policy_mapping_fn): the observation row holds one block per agent, the action row one column per agent; every agent
    is a sample of the shared policy and receives the shared reward."""
    from flow_amd.dist import allreduce_sum, shard_range
    from flow_amd.envs import VecFlowEnv
    local = int(os.environ.get("LOCAL_RANK", "0")) if world > 1 else 0
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    torch.manual_seed(seed)                                   # the same initial policy on every rank
    lo, hi = shard_range(replicas, rank, world)
    vec = VecFlowEnv(flow_params, num_replicas=hi - lo, device=local, replica_offset=lo)
    n_ag = vec.act_dim if shared_agents else 1
    if shared_agents and (vec.act_dim < 1 or vec.obs_dim % vec.act_dim):
        raise ValueError("shared_agents: the observation (%d) is not one block per action column (%d)" % (vec.obs_dim, vec.act_dim))
    k_ag = vec.obs_dim // n_ag
    pi = GaussianPolicy(k_ag if shared_agents else vec.obs_dim, 1 if shared_agents else vec.act_dim).to(dev)
    opt = torch.optim.Adam(pi.parameters(), lr=lr)
    # the rollout: ONE kernel per fragment where the library has the fused policy + step form for this experiment and
    # model (fs_policy_rollout_dev: rings with one RL vehicle, WaveAttenuationPOEnv, 1..3 hidden layers of 32 tanh units);
    # otherwise K single steps around the torch policy captured as one HIP graph
    fused, graph = None, None
    try:
        if shared_agents:
            raise NotImplementedError("one policy shared by %d agents per replica" % n_ag)
        from flow_amd.utils.device_policy import DevicePolicy
        fused = DevicePolicy([pi.mu[0], pi.mu[2]], pi.mu[4], log_std=pi.log_std, seed=seed)
        vec.reset()
        vec.policy_rollout(fused, 1, reset_done=True)           # (probe: raises NotImplementedError when not built)
        kernel = vec.sim.last_kernel
        vec.reset()
        log("rollout: fused policy + step kernel (%s)" % kernel)
    except NotImplementedError as e:
        fused = None
        log("rollout: HIP graph of %d single steps around the torch policy (%s)" % (fragment, e))
        if world > 1:
            torch.manual_seed(seed + 1000 * (rank + 1))       # the graph's torch.randn: another stream per rank
        R_ = hi - lo

        def act_shared(obs):                                   # [R, agents * k] -> [R, agents]: every agent its own sample
            return pi.act(obs.view(R_ * n_ag, k_ag)).view(R_, n_ag)
        graph = vec.capture(fragment, policy=act_shared if shared_agents else pi.act, reset_done=True)
        graph.begin(vec.reset())
    K, R = fragment, hi - lo
    history = []
    for it in range(iterations):
        vec.redraw_ring_lengths()                              # pending ring length per replica for its next in-graph reset
        t0 = time.perf_counter()
        if fused is not None:
            fused.sync()                                       # the optimiser moved the weights: repack them for the kernel
            obs, act, _, rew, done = vec.policy_rollout(fused, K, reset_done=True)
            torch.cuda.synchronize(dev)
        else:
            obs, act, rew, done = graph.replay()               # K closed-loop steps of R replicas: one graph launch
            graph.synchronize()
        t_roll = time.perf_counter() - t0
        if shared_agents:                                      # agents become samples: [K(+1), R * agents, .]
            shard = (obs.reshape(K + 1, R * n_ag, k_ag), act.reshape(K, R * n_ag, 1),
                     rew.repeat_interleave(n_ag, dim=1), done.repeat_interleave(n_ag, dim=1))
        else:
            shard = (obs, act, rew, done)
        ppo_update(pi, opt, [shard], epochs=epochs)
        tot = allreduce_sum(torch.stack([rew.double().sum(), torch.tensor(float(rew.numel()), dtype=torch.float64, device=dev),
                                         (done != 0).sum().double()]))
        mean_rew = float(tot[0] / tot[1])
        history.append(mean_rew)
        if rank == 0:
            log("iteration %3d  mean step reward %8.4f  rollout %.1f ms (%.2f M env-steps/s per GPU, %d GPU%s)  episodes ended %d"
                % (it, mean_rew, t_roll * 1e3, K * R / t_roll / 1e6, world, "s" if world > 1 else "", int(tot[2])))
    vec.close()
    return history


def spawn_ranks(gpus, argv):
    """--gpus N without a launcher: start the N ranks as child processes.  Nothing in THIS process has touched a GPU
    (torch.cuda.device_count() does not initialise it); never an exec of a process that has."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < gpus:
        raise SystemExit("train_vec.py --gpus %d: this node exposes %d GPU(s)" % (gpus, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    for pr in procs:
        rc = max(rc, abs(pr.wait()))
    raise SystemExit(rc)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--replicas", type=int, default=1024)
    ap.add_argument("--fragment", type=int, default=100, help="env steps per captured graph")
    ap.add_argument("--horizon", type=int, default=500)
    ap.add_argument("--iterations", type=int, default=20)
    ap.add_argument("--epochs", type=int, default=4)
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--gpus", type=int, default=1, help="data-parallel over this many GPUs of the node (one process each)")
    args = ap.parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    force = os.environ.get("TRAIN_FORCE_DIST") == "1"            # the N > 1 code path (RCCL init, all-reduces) with one rank
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus, sys.argv[1:] if argv is None else argv)
    if args.gpus != world and not (args.gpus == 1 and world == 1):
        raise SystemExit("train_vec.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 or force:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29544")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    try:
        return train_on_device(ring_flow_params(args.horizon), replicas=args.replicas, fragment=args.fragment,
                               iterations=args.iterations, epochs=args.epochs, lr=args.lr, rank=rank, world=world)
    finally:
        if world > 1 or force:
            import torch.distributed as dist
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
