"""train_vec.py -- a minimal on-device PPO loop over VecFlowEnv (the GPU counterpart of the reference's
examples/train.py:110-212, where RLlib rollout workers each drive one SUMO process).

    python examples/train_vec.py --replicas 1024 --iterations 20

Environment: the reference's single-agent ring experiment (examples/exp_configs/rl/singleagent/singleagent_ring.py:
WaveAttenuationPOEnv, 21 IDM humans + 1 RL vehicle on a 230..270 m ring).  A rollout fragment of K steps -- policy
forward, action sampling, Env.step of every replica, reset of finished episodes -- is ONE replay of a captured HIP
graph (VecFlowEnv.capture); observations, actions, rewards and the PPO update all stay in HBM.
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


def train_on_device(flow_params, replicas=1024, fragment=100, iterations=20, epochs=4, lr=3e-4, seed=0, log=print):
    """PPO on R replicas of ``flow_params`` with everything in HBM: returns the mean step reward per iteration."""
    from flow_amd.envs import VecFlowEnv
    dev = torch.device("cuda", 0)
    torch.manual_seed(seed)
    vec = VecFlowEnv(flow_params, num_replicas=replicas, device=0)
    pi = GaussianPolicy(vec.obs_dim, vec.act_dim).to(dev)
    opt = torch.optim.Adam(pi.parameters(), lr=lr)
    # the rollout: ONE kernel per fragment where the library has the fused policy + step form for this experiment and
    # model (fs_policy_rollout_dev: rings with one RL vehicle, WaveAttenuationPOEnv, 1..3 hidden layers of 32 tanh units);
    # otherwise K single steps around the torch policy captured as one HIP graph
    fused, graph = None, None
    try:
        from flow_amd.utils.device_policy import DevicePolicy
        fused = DevicePolicy([pi.mu[0], pi.mu[2]], pi.mu[4], log_std=pi.log_std, seed=seed)
        vec.reset()
        vec.policy_rollout(fused, 1, reset_done=True)           # (probe: raises NotImplementedError when not built)
        vec.reset()
        log("rollout: fused policy + step kernel (%s)" % vec.sim.last_kernel)
    except NotImplementedError as e:
        fused = None
        log("rollout: HIP graph of %d single steps around the torch policy (%s)" % (fragment, e))
        graph = vec.capture(fragment, policy=pi.act, reset_done=True)
        graph.begin(vec.reset())
    K, R = fragment, replicas
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
        o, a = obs[:K].reshape(K * R, -1), act.reshape(K * R, -1)
        with torch.no_grad():
            logp_old, val = pi.logp_value(o, a)
            last_val = pi.value(obs[K]).squeeze(-1)
            adv, ret = gae(rew, val.view(K, R), done, last_val)
            adv = ((adv - adv.mean()) / (adv.std() + 1e-8)).reshape(-1)
            ret = ret.reshape(-1)
        for _ in range(epochs):
            logp, v = pi.logp_value(o, a)
            ratio = (logp - logp_old).exp()
            loss = -torch.min(ratio * adv, ratio.clamp(0.8, 1.2) * adv).mean() + 0.5 * (v - ret).pow(2).mean()
            opt.zero_grad()
            loss.backward()
            opt.step()
        mean_rew = float(rew.mean())
        history.append(mean_rew)
        log("iteration %3d  mean step reward %8.4f  rollout %.1f ms (%.2f M env-steps/s)  episodes ended %d"
            % (it, mean_rew, t_roll * 1e3, K * R / t_roll / 1e6, int((done != 0).sum())))
    vec.close()
    return history


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--replicas", type=int, default=1024)
    ap.add_argument("--fragment", type=int, default=100, help="env steps per captured graph")
    ap.add_argument("--horizon", type=int, default=500)
    ap.add_argument("--iterations", type=int, default=20)
    ap.add_argument("--epochs", type=int, default=4)
    ap.add_argument("--lr", type=float, default=3e-4)
    args = ap.parse_args(argv)
    return train_on_device(ring_flow_params(args.horizon), replicas=args.replicas, fragment=args.fragment,
                           iterations=args.iterations, epochs=args.epochs, lr=args.lr)


if __name__ == "__main__":
    main()
