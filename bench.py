#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched ring step loop (BASELINE.json configs[1], "C2").

    python bench.py --gpus N --steps K --warmup W

One "step" = one Env.step of ALL replicas on every rank (controllers -> fail-safes ->
integration -> headways -> crash check -> observation + reward + done written to HBM).
Workload per GPU (weak scaling): 4096 RingNetwork replicas x 22 IDM vehicles, fp32,
sim_step 0.1, AccelEnv observation [4096, 44] written EVERY step, episodes of
`horizon` = 1500 steps; each launch of the rollout kernel advances one fragment
(<= 1500 steps) with the state in registers, and every replica is reset
(Env.reset) when its episode ends.  Inputs are resident in HBM before the timed
region.  For N > 1 each rank owns its own replicas (no data-path collective inside
a step); the only exchange is one RCCL all-gather of the fragment's final
observation/reward/done to the learner per fragment.

Prints ONE JSON line (rank 0).  `value` / `ms_per_step` describe exactly the --steps given.  Whatever --steps is,
`roofline` (dominant kernel, per-launch HIP events) and `parity` (deviation of the reported dtype from the float64
oracle after a 1500-step episode) come from >= 5 full 1500-step launches made in the same process.  Extra keys at
N = 1: `rollout_1500` (the same for mixed / f32 / f64), `cpu_baseline` (oracle C port on the host cores, bounded
sample), `step_api` (one launch per env step), `ring_default_speed_mode` (the reference's ring experiment with its
default speed mode: generic kernel), `c3_*` / `c4_bottleneck` / `c5_merge` (the other BASELINE configs, informational).
`--gpus N` without a launcher starts its N ranks itself.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
IDM_DEFAULT = [30, 1, 1, 1.5, 4, 2, 0, 0]      # flow/controllers/car_following_models.py:437-447


def c2_spec(R, N=22, seed=0, horizon=1500):
    """ring 230 m, 22 IDM vehicles, bunching 20, per-replica N(0, 0.5 m)
    start perturbation so replicas do not share a trajectory."""
    from flow_amd.networks.ring import ring_start_positions
    x0 = ring_start_positions(N, length=230.0, bunching=20.0)
    rng = np.random.default_rng(seed)
    pos = x0[None, :] + np.abs(rng.normal(0.0, 0.5, (R, N)))
    veh = dict(controller=2, p=IDM_DEFAULT, fail_safe=0, noise=0.0, delay=0.0, max_accel=2.6, max_decel=4.5,
               length=5.0, speed_mode=0, sumo_tau=1.0, sumo_min_gap=2.5, sumo_max_speed=30.0, rl_index=-1)
    return dict(num_replicas=R, num_vehicles=N, num_rl=0, sim_step=0.1, junction_length=0.1,
                ring_length=np.full(R, 230.0), max_speed=30.0, env=0, target_velocity=10.0,
                action_low=-3.0, action_high=3.0, horizon=horizon, warmup_steps=0, sims_per_step=1,
                vehicles=[dict(veh) for _ in range(N)], init_pos=pos)


class Runner:
    """Drives one FlowSim handle in fragments of <= horizon steps with episode resets."""

    def __init__(self, spec, precision, device, fragment):
        import torch
        from flow_amd.sim import FlowSim
        self.torch = torch
        self.sim = FlowSim(spec, precision=precision, device=device.index)
        self.sim.set_stream(torch.cuda.current_stream(device).cuda_stream)
        self.R, self.obs_dim = self.sim.R, self.sim.obs_dim
        self.horizon = spec["horizon"]
        self.fragment = min(fragment, self.horizon)
        self.obs = torch.empty((self.fragment, self.R, self.obs_dim), dtype=torch.float32, device=device)
        self.rew = torch.empty((self.fragment, self.R), dtype=torch.float32, device=device)
        self.done = torch.empty((self.fragment, self.R), dtype=torch.uint8, device=device)
        self.obs0 = torch.empty((self.R, self.obs_dim), dtype=torch.float32, device=device)
        self.t_in_episode = 0
        self.sim.reset_dev(self.obs0)
        self.events = []
        self.launches = 0
        # (HIP events for the per-launch times, made here: creating them inside the timed region cost a short run --
        # the driver's --steps 20 is ONE launch -- tens of microseconds of host time per launch)
        self._event_pool = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]

    def run(self, steps, record=False, after_fragment=None):
        """`record`: bracket the whole call with ONE pair of HIP events when it holds a full fragment (their elapsed time per
        fragment feeds `fragment_latency`).  Not one pair per launch: on this stack an event record is a packet of its own
        with ~40 us of latency -- two per launch were 80 us of a 410 us fragment and three quarters of a 20-step run."""
        torch = self.torch
        left = steps
        pair = None
        if record and steps >= self.fragment:
            pair = self._event_pool.pop()
            pair[0].record()
        n_full = 0
        while left > 0:
            k = min(self.fragment, left, self.horizon - self.t_in_episode)
            self.sim.rollout_dev(k, self.obs, self.rew, self.done, obs_every_step=True)
            n_full += 1 if k == self.fragment else 0
            if record:
                self.launches += 1
            if after_fragment is not None:
                after_fragment(self.obs[k - 1], self.rew[k - 1], self.done[k - 1])
            self.t_in_episode += k
            left -= k
            if self.t_in_episode >= self.horizon:          # every replica is done: Env.reset
                self.sim.reset_dev(self.obs0)
                self.t_in_episode = 0
        if pair is not None:
            pair[1].record()
            self.events.append((pair[0], pair[1], steps, n_full))

    def run_step_api(self, steps):
        for _ in range(steps):
            self.sim.step_dev(self.obs[0], self.rew[0], self.done[0])
            self.t_in_episode += 1
            if self.t_in_episode >= self.horizon:
                self.sim.reset_dev(self.obs0)
                self.t_in_episode = 0


def host_cores():
    """Cores this job may really use: affinity, capped by the cgroup CPU quota when there is one
    and by the GPU box's per-GPU CPU share (16)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def c3_closed_loop_leg(device, R=4096, K=500):
    """C3 closed loop: policy (fcnet obs-32-32-32-2 tanh, diagonal Gaussian: examples/train.py:152) -> action -> Env.step ->
    reset of finished episodes, K steps per launch of the fused kernel k_loop_policy (fs_policy_rollout_dev), both heads."""
    import torch
    from flow_amd.envs import VecFlowEnv
    from flow_amd.utils.device_policy import DevicePolicy
    out = {"unit": "env-steps/s", "replicas": R, "steps_per_launch": K,
           "workload": "C3 closed loop: FigureEightNetwork, 13 IDM (noise 0.2) + 1 RL, policy in the loop (one launch per "
                       "%d-step fragment, in-fragment resets)" % K}
    for label, po in (("po_head", True), ("accel_head", False)):
        fp = c3_flow_params(po)
        vec = VecFlowEnv(fp, num_replicas=R, device=device.index)
        hidden = [torch.nn.Linear(vec.obs_dim, 32), torch.nn.Linear(32, 32), torch.nn.Linear(32, 32)]
        head = torch.nn.Linear(32, 2)
        for l in hidden + [head]:
            l.to(device)
        pol = DevicePolicy(hidden, head, seed=1)
        vec.reset()
        res = vec.policy_rollout(pol, K, reset_done=True)
        torch.cuda.synchronize(device)
        ms = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            vec.policy_rollout(pol, K, reset_done=True, out=res)
            e1.record()
            torch.cuda.synchronize(device)
            ms.append(e0.elapsed_time(e1))
        t = float(np.median(ms)) * 1e-3
        out[label] = {"value": R * K / t, "median_launch_ms": t * 1e3, "kernel": vec.sim.last_kernel, "obs_dim": vec.obs_dim,
                      "model": "fcnet %d-32-32-32-2 tanh, diagonal Gaussian" % vec.obs_dim}
        vec.close()
    out["value"] = out["po_head"]["value"]
    return out


def c3_flow_params(po=False, precision="f32"):
    """The flow_params of BASELINE configs[2] (see c3_leg)."""
    from flow_amd.controllers import ContinuousRouter, IDMController, RLController
    from flow_amd.core.params import EnvParams, NetParams, SumoCarFollowingParams, SumoParams, VehicleParams
    from flow_amd.envs import AccelEnv, WaveAttenuationPOEnv
    from flow_amd.networks import FigureEightNetwork
    from flow_amd.networks.figure_eight import ADDITIONAL_NET_PARAMS
    veh = VehicleParams()
    veh.add(veh_id="human", acceleration_controller=(IDMController, {"noise": 0.2}),
            routing_controller=(ContinuousRouter, {}),
            car_following_params=SumoCarFollowingParams(speed_mode="obey_safe_speed", decel=1.5), num_vehicles=13)
    veh.add(veh_id="rl", acceleration_controller=(RLController, {}), routing_controller=(ContinuousRouter, {}),
            car_following_params=SumoCarFollowingParams(speed_mode="obey_safe_speed", decel=1.5), num_vehicles=1)
    add = {"max_accel": 3, "max_decel": 3, "ring_length": None} if po else \
        {"target_velocity": 20, "max_accel": 3, "max_decel": 3, "sort_vehicles": False}
    return dict(exp_tag="figure_eight", env_name=WaveAttenuationPOEnv if po else AccelEnv, network=FigureEightNetwork,
                simulator="traci", sim=SumoParams(sim_step=0.1, render=False, seed=7, precision=precision),
                env=EnvParams(horizon=1500, additional_params=add),
                net=NetParams(additional_params=dict(ADDITIONAL_NET_PARAMS)), veh=veh)


def c3_leg(device, R=4096, steps=3000, po=False, precision="f32"):
    """BASELINE configs[2] (informational, not the headline): FigureEightNetwork, 13 noisy IDM + 1 RL vehicle,
    random RL actions from a pre-generated tape, rollout kernel of segment-table loops (flowsim_fig8.h).  ``po=False``: AccelEnv observation (28),
    the pairing the reference itself uses (singleagent_figure_eight.py:44); ``po=True``: the 3-value
    WaveAttenuationPOEnv observation BASELINE.json names."""
    import torch
    from flow_amd.controllers import ContinuousRouter, IDMController, RLController
    from flow_amd.core.params import EnvParams, NetParams, SumoCarFollowingParams, SumoParams, VehicleParams
    from flow_amd.envs import AccelEnv, VecFlowEnv, WaveAttenuationPOEnv
    from flow_amd.networks import FigureEightNetwork
    from flow_amd.networks.figure_eight import ADDITIONAL_NET_PARAMS
    veh = VehicleParams()
    veh.add(veh_id="human", acceleration_controller=(IDMController, {"noise": 0.2}),
            routing_controller=(ContinuousRouter, {}),
            car_following_params=SumoCarFollowingParams(speed_mode="obey_safe_speed", decel=1.5), num_vehicles=13)
    veh.add(veh_id="rl", acceleration_controller=(RLController, {}), routing_controller=(ContinuousRouter, {}),
            car_following_params=SumoCarFollowingParams(speed_mode="obey_safe_speed", decel=1.5), num_vehicles=1)
    add = {"max_accel": 3, "max_decel": 3, "ring_length": None} if po else \
        {"target_velocity": 20, "max_accel": 3, "max_decel": 3, "sort_vehicles": False}
    fp = dict(exp_tag="figure_eight", env_name=WaveAttenuationPOEnv if po else AccelEnv, network=FigureEightNetwork,
              simulator="traci", sim=SumoParams(sim_step=0.1, render=False, seed=7, precision=precision),
              env=EnvParams(horizon=1500, additional_params=add),
              net=NetParams(additional_params=dict(ADDITIONAL_NET_PARAMS)), veh=veh)
    vec = VecFlowEnv(fp, num_replicas=R, device=device.index)
    K = 1500
    gen = torch.Generator(device=device).manual_seed(1)
    tape = (torch.rand((K, R, 1), device=device, generator=gen) * 2 - 1) * 3.0
    out = (torch.empty((K, R, vec.obs_dim), dtype=torch.float32, device=device),
           torch.empty((K, R), dtype=torch.float32, device=device),
           torch.empty((K, R), dtype=torch.uint8, device=device))
    if os.environ.get("BENCH_C3_NO_ACTIONS") == "1":     # diagnostic: the RL vehicle stays uncommanded (no action tape reads)
        tape = None
    vec.reset()
    vec.rollout(K, tape, out=out)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    done_steps = 0
    while done_steps < steps:
        vec.reset()
        vec.rollout(K, tape, out=out)
        done_steps += K
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    crashed = float((out[2][:-1].max(dim=0).values > 0).float().mean().item())
    kernel = vec.sim.last_kernel
    vec.close()
    return {"value": R * done_steps / dt, "unit": "env-steps/s", "steps": done_steps, "steps_per_launch": K, "replicas": R,
            "obs_dim": 3 if po else 28, "replicas_crashed_before_horizon": crashed, "kernel": kernel, "precision": precision,
            "workload": "C3: FigureEightNetwork r=30, 13 IDM (noise 0.2, obey_safe_speed) + 1 RL, %s, "
                        "random actions" % ("WaveAttenuationPOEnv" if po else "AccelEnv")}


def c5_flow_params(precision="f32", noise=0.2):
    """The flow_params of BASELINE configs[4] (see c5_leg)."""
    from flow_amd.controllers import IDMController, RLController
    from flow_amd.core.params import (EnvParams, InFlows, NetParams, SumoCarFollowingParams, SumoParams,
                                      VehicleParams)
    from flow_amd.envs.multiagent import MultiAgentMergePOEnv
    from flow_amd.networks import MergeNetwork
    from flow_amd.networks.merge import ADDITIONAL_NET_PARAMS
    add = dict(ADDITIONAL_NET_PARAMS)
    add["pre_merge_length"] = 500
    veh = VehicleParams()
    veh.add(veh_id="human", acceleration_controller=(IDMController, {"noise": noise}),
            car_following_params=SumoCarFollowingParams(speed_mode="obey_safe_speed"), num_vehicles=5)
    veh.add(veh_id="rl", acceleration_controller=(RLController, {}),
            car_following_params=SumoCarFollowingParams(speed_mode="obey_safe_speed"), num_vehicles=0)
    inflow = InFlows()
    inflow.add(veh_type="human", edge="inflow_highway", vehs_per_hour=1800, depart_lane="free", depart_speed=10)
    inflow.add(veh_type="rl", edge="inflow_highway", vehs_per_hour=200, depart_lane="free", depart_speed=10)
    inflow.add(veh_type="human", edge="inflow_merge", vehs_per_hour=100, depart_lane="free", depart_speed=7.5)
    fp = dict(exp_tag="multiagent_merge", env_name=MultiAgentMergePOEnv, network=MergeNetwork, simulator="traci",
              sim=SumoParams(sim_step=0.2, render=False, restart_instance=True, seed=11, precision=precision),
              env=EnvParams(horizon=600, sims_per_step=5, warmup_steps=0,
                            additional_params={"max_accel": 1.5, "max_decel": 1.5, "target_velocity": 20}),
              net=NetParams(inflows=inflow, additional_params=add), veh=veh)
    return fp


def c5_leg(device, R=1024, env_steps=600, precision="f32"):
    """BASELINE configs[4] (informational, not the headline): MergeNetwork pre_merge 500 m, 5 initial humans +
    inflows 1800 / 200 (RL) / 100 veh/h, sim_step 0.2, sims_per_step 5, horizon 600, MultiAgentMergePOEnv head
    (examples/exp_configs/rl/multiagent/multiagent_merge.py); 1024 replicas per GPU = 8192 over 8 GPUs; the
    open-network kernel k_steps_open, 64 vehicle slots per replica.  ``precision="f16s"`` is the configuration as BASELINE
    names it -- fp16 state, fp32 integrator: positions (two halves) and speeds (one) are kept as IEEE halves in HBM
    between launches, a launch steps in float32 (include/flowsim.h FS_F16S); "f32" keeps float32 state."""
    import torch
    from flow_amd import _lib as L
    from flow_amd.envs import VecFlowEnv
    fp = c5_flow_params(precision)
    vec = VecFlowEnv(fp, num_replicas=R, device=device.index)
    K = env_steps
    out = (torch.empty((K, R, vec.obs_dim), dtype=torch.float32, device=device),
           torch.empty((K, R), dtype=torch.float32, device=device),
           torch.empty((K, R), dtype=torch.uint8, device=device))
    vec.reset()
    vec.sim.rollout_dev(60, out[0][:60], out[1][:60], out[2][:60])       # warm-up launch
    vec.reset()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    vec.sim.rollout_dev(K, *out)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    cnt = vec.sim.get_state(L.FS_FIELD_COUNTERS)
    route = vec.sim.get_state(L.FS_FIELD_ROUTE)
    sp, al = vec.sim.vel, route >= 0
    res = {"kernel": vec.sim.last_kernel, "value": R * K * 5 / dt, "unit": "env-steps/s (simulation sub-steps)", "env_steps": K, "sims_per_step": 5,
           "replicas": R, "gym_steps_per_s": R * K / dt, "obs_dim": vec.obs_dim, "state": precision,
           "vehicles_in_network_mean": float((route >= 0).sum(axis=1).mean()),
           "departed_mean": float(cnt[:, 6].mean()), "arrived_mean": float(cnt[:, 5].mean()),
           # the replicas differ by their acceleration-noise streams only (the inflow schedule is deterministic)
           "spread_over_replicas": {
               "vehicles_in_network": [int((route >= 0).sum(axis=1).min()), int((route >= 0).sum(axis=1).max())],
               "departed": [int(cnt[:, 6].min()), int(cnt[:, 6].max())], "arrived": [int(cnt[:, 5].min()), int(cnt[:, 5].max())],
               "mean_speed_std_mps": float(np.std([sp[r][al[r]].mean() for r in range(min(R, 256))])),
               "episode_return_std": float(out[1].sum(dim=0).std().item())},
           "workload": "C5: MergeNetwork pre_merge 500 m, inflows 1800 + 200 RL + 100 veh/h, 64 slots, "
                       "MultiAgentMergePOEnv head, sim_step 0.2 x 5 sub-steps, one 600-step episode"}
    vec.close()
    return res


def c4_flow_params(slots=256, precision="f32", lane_change_mode=0):
    """The flow_params of BASELINE configs[3] (see c4_leg).  ``lane_change_mode``: 0 as the reference's experiment ships
    (examples/exp_configs/rl/singleagent/singleagent_bottleneck.py:36-50: nobody changes lane), 1621 as flow/benchmarks/
    bottleneck1.py (SUMO changes lanes: the simplified model M11 here)."""
    from flow_amd.controllers import ContinuousRouter, RLController, SimLaneChangeController
    from flow_amd.core.params import (EnvParams, InFlows, InitialConfig, NetParams, SumoCarFollowingParams,
                                      SumoLaneChangeParams, SumoParams, VehicleParams)
    from flow_amd.envs import BottleneckDesiredVelocityEnv
    from flow_amd.networks import BottleneckNetwork
    veh = VehicleParams()
    veh.add(veh_id="human", lane_change_controller=(SimLaneChangeController, {}),
            routing_controller=(ContinuousRouter, {}),
            car_following_params=SumoCarFollowingParams(speed_mode="all_checks"),
            lane_change_params=SumoLaneChangeParams(lane_change_mode=lane_change_mode), num_vehicles=1)
    veh.add(veh_id="followerstopper", acceleration_controller=(RLController, {}),
            lane_change_controller=(SimLaneChangeController, {}), routing_controller=(ContinuousRouter, {}),
            car_following_params=SumoCarFollowingParams(speed_mode=9),
            lane_change_params=SumoLaneChangeParams(lane_change_mode=0), num_vehicles=1)
    add = {"target_velocity": 40, "disable_tb": True, "disable_ramp_metering": True,
           "controlled_segments": [("1", 1, False), ("2", 2, True), ("3", 2, True), ("4", 2, True), ("5", 1, False)],
           "symmetric": False, "observed_segments": [("1", 1), ("2", 3), ("3", 3), ("4", 3), ("5", 1)],
           "reset_inflow": False, "lane_change_duration": 5, "max_accel": 3, "max_decel": 3,
           "inflow_range": [1000, 2000]}
    inflow = InFlows()
    inflow.add(veh_type="human", edge="1", vehs_per_hour=2300 * 0.9, depart_lane="random", depart_speed=10)
    inflow.add(veh_type="followerstopper", edge="1", vehs_per_hour=2300 * 0.1, depart_lane="random", depart_speed=10)
    fp = dict(exp_tag="DesiredVelocity", env_name=BottleneckDesiredVelocityEnv, network=BottleneckNetwork,
              simulator="traci", sim=SumoParams(sim_step=0.5, render=False, restart_instance=True, seed=5,
                                                max_vehicles=slots, precision=precision),
              env=EnvParams(warmup_steps=40, sims_per_step=1, horizon=1000, additional_params=add),
              net=NetParams(inflows=inflow, additional_params={"scaling": 1, "speed_limit": 23}), veh=veh,
              initial=InitialConfig(spacing="uniform", min_gap=5, lanes_distribution=float("inf"),
                                    edges_distribution=["2", "3", "4", "5"]))
    return fp


def c4_leg(device, R=128, env_steps=1000, slots=256, precision="f32", lane_change_mode=0):
    """BASELINE configs[3] (informational, not the headline): BottleneckNetwork scaling 1 (4 -> 2 -> 1 lanes at two
    zipper junctions), inflow 2300 veh/h (10 % RL) with random entry lanes, all vehicles on the SUMO car-following
    model, BottleneckDesiredVelocityEnv head (141 observations, 20 actions), sim_step 0.5, warm-up 40 + horizon 1000
    (examples/exp_configs/rl/singleagent/singleagent_bottleneck.py); 128 replicas per GPU = 1024 over 8 GPUs."""
    import torch
    from flow_amd import _lib as L
    from flow_amd.envs import VecFlowEnv
    fp = c4_flow_params(slots, precision, lane_change_mode)
    vec = VecFlowEnv(fp, num_replicas=R, device=device.index)
    K = env_steps
    gen = torch.Generator(device=device).manual_seed(3)
    tape = (torch.rand((K, R, vec.act_dim), device=device, generator=gen) * 2 - 1) * 1.5
    out = (torch.empty((K, R, vec.obs_dim), dtype=torch.float32, device=device),
           torch.empty((K, R), dtype=torch.float32, device=device),
           torch.empty((K, R), dtype=torch.uint8, device=device))
    vec.reset()
    vec.sim.rollout_dev(50, out[0][:50], out[1][:50], out[2][:50], actions=tape[:50])       # warm-up launch
    vec.reset()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    vec.sim.rollout_dev(K, *out, actions=tape)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    cnt = vec.sim.get_state(L.FS_FIELD_COUNTERS)
    route = vec.sim.get_state(L.FS_FIELD_ROUTE)
    res = {"kernel": vec.sim.last_kernel, "value": R * K / dt, "unit": "env-steps/s", "env_steps": K, "replicas": R, "obs_dim": vec.obs_dim, "precision": precision,
           "act_dim": vec.act_dim, "vehicles_in_network_mean": float((route >= 0).sum(axis=1).mean()),
           "departed_mean": float(cnt[:, 6].mean()), "arrived_mean": float(cnt[:, 5].mean()),
           "dropped_at_insertion_mean": float(cnt[:, 7].mean()),
           "outflow_veh_per_hour_mean": float(out[1][-200:].mean().item() * 2000.0),
           "slots": slots, "lane_change_mode": lane_change_mode,
           "workload": "C4: BottleneckNetwork 4->2->1 lanes, inflow 2300 veh/h (10 %% RL, random lanes), %d slots, "
                       "BottleneckDesiredVelocityEnv head (141 obs / 20 actions), sim_step 0.5, one 1000-step episode "
                       "after 40 warm-up steps, random actions, lane_change_mode %d" % (slots, lane_change_mode)}
    vec.close()
    return res


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def sumo_probe():
    """is the reference's own CPU path (SUMO + TraCI) runnable on this box?"""
    import importlib.util
    import shutil
    have = {"sumo": shutil.which("sumo"), "netconvert": shutil.which("netconvert"),
            "traci": importlib.util.find_spec("traci") is not None}
    if have["sumo"] and have["netconvert"] and have["traci"]:
        return "SUMO present (%s) -- the reference path is not driven by this bench (no harness ships with the build)" % have["sumo"]
    missing = [k for k, v in have.items() if not v]
    return "SUMO unavailable -- reference CPU path not timed (missing: %s)" % ", ".join(missing)


def _time_c_port(spec_fn, threads, seconds):
    from oracle import cbuild
    R = 4096
    sim = cbuild.CRingIDM(spec_fn(R), np.float32, threads=threads)
    sim.rollout(20, obs_every_step=True)               # warm-up
    chunk, done_steps, t0 = 100, 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:          # episodes of 1500 steps
        if done_steps % 1500 == 0:
            sim.reset()
        sim.rollout(chunk, obs_every_step=True)
        done_steps += chunk
    dt = time.perf_counter() - t0
    return R * done_steps / dt, done_steps, dt


def cpu_baseline(spec_fn, seconds=7.0):
    """BASELINE.md section 3 on the host cores, bounded samples of the same workload: (1) the SUMO probe; (2) the oracle's
    C port (oracle/csim: the build's CPU restatement, NOT SUMO) on all cores -- `value` -- and on one thread; (3) the
    float64 numpy oracle on C1 (1 env x 1500 steps: the reference's arithmetic type and its per-env rate definition,
    core/experiment.py:136-149, 178-179)."""
    from oracle import refsim as S
    cores = host_cores()
    v_all, steps_all, dt_all = _time_c_port(spec_fn, cores, seconds)
    v_one, steps_one, dt_one = _time_c_port(spec_fn, 1, min(seconds, 5.0))
    spec1 = spec_fn(1)
    ora = S.RingOracle(spec1, np.float64)
    ora.reset()
    t0 = time.perf_counter()
    for _ in range(1500):
        ora.step(None)
    dt_np = time.perf_counter() - t0
    return {"value": v_all, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "oracle/csim C port (float32, OpenMP over replicas, obs written every step), "
                      "4096 replicas x 22 vehicles x %d steps in %.1f s" % (steps_all, dt_all),
            "single_thread": {"value": v_one, "cores": 1,
                              "sample": "the same port on one thread, 4096 x 22 x %d steps in %.1f s" % (steps_one, dt_one)},
            "numpy_float64_c1": {"value": 1500.0 / dt_np, "cores": 1,
                                 "sample": "oracle/refsim.py RingOracle float64, C1 (1 env x 22 vehicles x 1500 steps) in %.2f s" % dt_np},
            "sumo": sumo_probe(), "nproc": os.cpu_count(), "cpu_model": cpu_model()}


def ring_defaults_leg(device, R=4096, K=1500):
    """The reference's own 22-IDM ring experiment AS SHIPPED (examples/exp_configs/non_rl/ring.py:13-61: the
    SumoCarFollowingParams default speed_mode 'right_of_way' = 25, whose bit 0 makes SUMO's safe-speed rule cap
    every command) through VecFlowEnv, in float32 and in FS_MIXED: k_rollout_pair with the speed-mode clamps
    compiled in (round 1 and the first half of round 2 ran it on the generic k_steps kernel, 2.5 G env-steps/s); and
    the same experiment with IDMController(noise=0.2), the human model of the reference's RL experiments."""
    import torch
    from flow_amd.controllers import ContinuousRouter, IDMController
    from flow_amd.core.params import EnvParams, InitialConfig, NetParams, SumoParams, VehicleParams
    from flow_amd.envs import AccelEnv, VecFlowEnv
    from flow_amd.networks import RingNetwork
    from flow_amd.networks.ring import ADDITIONAL_NET_PARAMS
    out = {"unit": "env-steps/s", "replicas": R,
           "workload": "examples/exp_configs/non_rl/ring.py through VecFlowEnv: 22 IDM, speed_mode 'right_of_way' "
                       "(the SumoCarFollowingParams default), %d-step rollout launches" % K}
    for precision, noise in (("mixed", 0.0), ("f32", 0.0), ("f32", 0.2)):
        veh = VehicleParams()
        veh.add(veh_id="idm", acceleration_controller=(IDMController, {"noise": noise} if noise else {}),
                routing_controller=(ContinuousRouter, {}), num_vehicles=22)
        fp = dict(exp_tag="ring", env_name=AccelEnv, network=RingNetwork, simulator="traci",
                  sim=SumoParams(render=False, sim_step=0.1, precision=precision),
                  env=EnvParams(horizon=1500, additional_params={"max_accel": 3, "max_decel": 3, "target_velocity": 10,
                                                                 "sort_vehicles": False}),
                  net=NetParams(additional_params=dict(ADDITIONAL_NET_PARAMS)), veh=veh,
                  initial=InitialConfig(bunching=20))
        vec = VecFlowEnv(fp, num_replicas=R, device=device.index)
        buf = (torch.empty((K, R, vec.obs_dim), dtype=torch.float32, device=device),
               torch.empty((K, R), dtype=torch.float32, device=device),
               torch.empty((K, R), dtype=torch.uint8, device=device))
        vec.reset()
        vec.rollout(K, None, out=buf)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        n = 0
        for _ in range(3):
            vec.reset()
            vec.rollout(K, None, out=buf)
            n += K
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        out[precision + ("_noise_%g" % noise if noise else "")] = {"value": R * n / dt, "steps": n,
                                                                   "kernel": vec.sim.last_kernel}
        vec.close()
    out["value"] = out["mixed"]["value"]
    return out


def rl_ring_legs(device, R=4096, K=1500):
    """The reference's RL ring experiment (examples/exp_configs/rl/singleagent/singleagent_ring.py:17-65: 21 x
    IDMController(noise=0.2) + 1 x RLController, WaveAttenuationPOEnv, ring length 220..270 per replica) through
    VecFlowEnv: (a) open loop with an action tape on k_ring_pair (kernel time of 1500-step launches), float32 and FS_MIXED
    with the noise (the experiment as shipped) and FS_MIXED without; (b) closed loop: policy (fcnet_hiddens [32, 32, 32], diagonal Gaussian) -> action ->
    step -> reset of finished episodes, K = 500 steps per launch of the fused kernel (fs_policy_rollout_dev)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import train_vec
    from flow_amd.envs import VecFlowEnv
    from flow_amd.utils.device_policy import DevicePolicy
    out = {"unit": "env-steps/s", "replicas": R,
           "workload": "singleagent_ring: 21 IDM (noise 0.2) + 1 RL, WaveAttenuationPOEnv, ring length 220..270 per replica"}
    for label, precision, noise in (("f32_noise_0.2", "f32", 0.2), ("mixed_noise_0.2", "mixed", 0.2), ("mixed_quiet", "mixed", 0.0)):
        fp = train_vec.ring_flow_params(3000)
        fp["sim"].precision = precision
        fp["env"].additional_params["ring_length"] = [220, 270]
        if not noise:
            for t in fp["veh"].type_parameters.values():
                if "noise" in t["acceleration_controller"][1]:
                    t["acceleration_controller"][1]["noise"] = 0.0
        vec = VecFlowEnv(fp, num_replicas=R, device=device.index)
        gen = torch.Generator(device=device).manual_seed(2)
        tape = torch.rand((K, R, 1), device=device, generator=gen) * 2 - 1
        buf = (torch.empty((K, R, vec.obs_dim), device=device), torch.empty((K, R), device=device),
               torch.empty((K, R), dtype=torch.uint8, device=device))
        vec.reset()
        vec.rollout(K, tape, out=buf)
        torch.cuda.synchronize(device)
        ms = []
        for _ in range(4):
            vec.reset()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            vec.rollout(K, tape, out=buf)
            e1.record()
            torch.cuda.synchronize(device)
            ms.append(e0.elapsed_time(e1))
        t = float(np.mean(ms)) * 1e-3
        leg = {"open_loop": {"value": R * K / t, "avg_launch_ms": t * 1e3, "steps_per_launch": K,
                             "kernel": vec.sim.last_kernel,
                             # PO head: 3 observations + reward + done per env step, one action read
                             "algorithmic_GBs": R * K * (3 * 4 + 4 + 1 + 4) / t / 1e9}}
        hidden = [torch.nn.Linear(3, 32), torch.nn.Linear(32, 32), torch.nn.Linear(32, 32)]
        head = torch.nn.Linear(32, 2)
        for l in hidden + [head]:
            l.to(device)
        pol = DevicePolicy(hidden, head, seed=1)
        KP = 500
        vec.reset()
        res = vec.policy_rollout(pol, KP, reset_done=True)
        torch.cuda.synchronize(device)
        ms = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            vec.policy_rollout(pol, KP, reset_done=True, out=res)
            e1.record()
            torch.cuda.synchronize(device)
            ms.append(e0.elapsed_time(e1))
        t = float(np.mean(ms)) * 1e-3
        leg["closed_loop_fused_policy"] = {"value": R * KP / t, "avg_launch_ms": t * 1e3, "steps_per_launch": KP,
                                           "kernel": vec.sim.last_kernel, "model": "fcnet 3-32-32-32-2 tanh, diagonal Gaussian"}
        out[label] = leg
        vec.close()
    out["value"] = out["f32_noise_0.2"]["open_loop"]["value"]
    return out


KERNEL_NAMES = {"f32": "fs::k_rollout_pair<float, 16, true, true, false>",
                "mixed": "fs::k_rollout_pair<double, 16, true, true, false>",
                "f64": "fs::k_rollout_idm<double, 32, true, false, false>"}


def launch_bytes(R, N, K, precision):
    """Algorithmic HBM bytes of ONE K-step rollout launch: observation [K,R,2N] f32 + reward f32 + done u8 written
    every step; state (pos, vel) read and written once, time counter once."""
    obs_b = 2 * N * 4 + 4 + 1
    st = 4 if precision == "f32" else 8
    state_b = N * (st + st) * 2 + 4 * 2
    return R * (K * obs_b + state_b), obs_b + state_b / float(K)


PMC_STALE_TOLERANCE = 0.15


def pmc_summary(precision, R, K, avg_s=None):
    """The committed PMC summary of THIS kernel at THIS launch shape (newest profiles/rNN_pmc_summary_<dtype>.json), or
    (None, reason).  A summary is only used while it still describes the kernel that was just timed: same kernel name,
    same launch shape, and a launch duration within 15 % of the one measured in this run -- a counter file of an older
    kernel would otherwise pass for a measurement."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_summary_%s.json" % precision)))
    if not paths:
        return None, "no profiles/rNN_pmc_summary_%s.json" % precision
    path = paths[-1]
    try:
        tj = json.load(open(path))
    except Exception as e:
        return None, "%s unreadable: %s" % (os.path.basename(path), e)
    # rocprofv3 prints the kernel as "void fs::k_...<...>(args)", with or without the defaulted template arguments
    # (SM = false, NOISE = false): the bench line's name must be that instantiation
    want = KERNEL_NAMES[precision]
    printed = tj.get("kernel", "")
    same_kernel = want in printed or (want[:-1] + ", false, false>") in printed
    if not (tj.get("replicas") == R and tj.get("steps_per_launch") == K and same_kernel):
        return None, "%s describes another kernel / launch shape" % os.path.basename(path)
    if avg_s is not None:
        ref_s = float(tj.get("avg_launch_ns_kernel_trace", 0.0)) * 1e-9
        if not ref_s > 0 or abs(avg_s - ref_s) / ref_s > PMC_STALE_TOLERANCE:
            return None, "%s is stale: its launches took %.4f ms, this run's %.4f ms (> %d %% apart)" % (
                os.path.basename(path), ref_s * 1e3, avg_s * 1e3, int(PMC_STALE_TOLERANCE * 100))
    return tj, os.path.basename(path)


def pmc_traffic(precision, R, K, avg_s=None):
    """HBM bytes per launch from that summary, or (None, why not)."""
    tj, src = pmc_summary(precision, R, K, avg_s)
    return (tj.get("hbm_bytes_per_launch"), src) if tj else (None, src)


VALU_PEAK_GINST = 256 * 4 * 2.4 / 4.0     # wave-instructions per ns the chip can issue: 1024 SIMDs, one wave64 VALU
#                                           instruction per 4 cycles each, 2.4 GHz (MI355X_MICROARCH.md) = 614.4 G/s


def valu_issue(precision, R, K, avg_s):
    """The second roof of this kernel: it is instruction-bound (about 20 VALU instructions per env step, most of them
    packed float32; scripts/sweep_pair.sh: the same kernel on a FULL chip reaches 0.49 of the HBM roof, not more).
    Wave-instructions per launch from the PMC summary (SQ_INSTS_VALU) over the launch time measured here.  Two peaks:
    the nominal one (2.4 GHz) and the one at the clock the chip actually ran this kernel at (GRBM_GUI_ACTIVE / 8 /
    launch time of the counter run: ~2.04 GHz while 3 TB/s leave the chip) -- `frac` is against the latter."""
    tj, src = pmc_summary(precision, R, K, avg_s)
    try:
        insts = float(tj["counters_per_launch"]["SQ_INSTS_VALU"]["mean"])
    except Exception:
        return {"frac": None, "source": src}
    clock = float(tj.get("effective_clock_GHz") or 0.0)
    peak_eff = 256 * 4 * clock / 4.0 if clock > 0 else None
    ach = insts / avg_s / 1e9
    return {"valu_insts_per_launch": insts, "valu_insts_per_env_step": insts / (R * K), "achieved_Ginst_s": ach,
            "peak_Ginst_s_nominal_2.4GHz": VALU_PEAK_GINST, "frac_of_nominal": ach / VALU_PEAK_GINST,
            "effective_clock_GHz": clock or None, "peak_Ginst_s": peak_eff,
            "frac": ach / peak_eff if peak_eff else None, "source": src}


def rollout_1500_leg(device, precision, R, launches=50, check_parity=True):
    """50 full 1500-step fragments of C2 in one process (launch times: mean, median, p10 / p90), per-launch HIP events on the kernel's
    stream, roofline of that kernel, and the deviation of the final state from the float64 oracle (oracle/csim,
    the reference's arithmetic) after one 1500-step episode: the parity figure of this dtype."""
    import torch
    from flow_amd.sim import FlowSim
    K, N = 1500, 22
    spec = c2_spec(R, seed=1000)
    sim = FlowSim(spec, precision=precision, device=device.index)
    sim.set_stream(torch.cuda.current_stream(device).cuda_stream)
    obs = torch.empty((K, R, 2 * N), dtype=torch.float32, device=device)
    rew = torch.empty((K, R), dtype=torch.float32, device=device)
    done = torch.empty((K, R), dtype=torch.uint8, device=device)
    obs0 = torch.empty((R, 2 * N), dtype=torch.float32, device=device)
    sim.reset_dev(obs0)
    sim.rollout_dev(K, obs, rew, done)
    torch.cuda.synchronize(device)
    parity = None
    if check_parity:
        from oracle import cbuild
        ref = cbuild.CRingIDM(spec, np.float64, threads=host_cores())
        ref.rollout(K)
        L = 230.4
        dx = np.abs(sim.pos.astype(np.float64) - ref.x)
        dx = float(np.minimum(dx, L - dx).max())
        dv = float(np.abs(sim.vel.astype(np.float64) - ref.v).max())
        parity = {"max_abs_dx_m": dx, "max_abs_dv_mps": dv, "steps": K, "replicas": R,
                  "against": "oracle/csim float64 (the reference's arithmetic type), all replicas, step 1500",
                  "within_1e-4": bool(dx < 1e-4 and dv < 1e-4)}
    events = []
    t0 = time.perf_counter()
    for _ in range(launches):
        sim.reset_dev(obs0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sim.rollout_dev(K, obs, rew, done)
        e1.record()
        events.append((e0, e1))
    torch.cuda.synchronize(device)
    wall = time.perf_counter() - t0
    ms = [a.elapsed_time(b) for a, b in events]
    avg_s = float(np.mean(ms)) * 1e-3
    nbytes, per_step = launch_bytes(R, N, K, precision)
    traffic, src = pmc_traffic(precision, R, K, avg_s)
    sim.close()
    return {"value": R * K / avg_s, "unit": "env-steps/s (kernel time of a 1500-step launch)",
            "value_wall": R * K * launches / wall, "dtype": precision, "launches_timed": launches,
            "avg_launch_ms": avg_s * 1e3, "min_launch_ms": float(min(ms)), "max_launch_ms": float(max(ms)),
            "median_launch_ms": float(np.median(ms)), "p10_launch_ms": float(np.percentile(ms, 10)),
            "p90_launch_ms": float(np.percentile(ms, 90)), "value_median": R * K / (float(np.median(ms)) * 1e-3),
            "roofline": {"bound": "hbm", "kernel": KERNEL_NAMES[precision], "achieved": nbytes / avg_s / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / avg_s / 1e9 / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": src, "bytes_per_launch": nbytes,
                         "steps_per_launch": K, "avg_launch_ms": avg_s * 1e3, "bytes_per_env_step": per_step,
                         "valu_issue": valu_issue(precision, R, K, avg_s)},
            "parity": parity}


def step_graph_leg(device, precision, R, K=100, replays=40):
    """The closed-loop call pattern without the per-launch host cost: K single-step launches (fs_step_dev, state
    round-trips through HBM between them, a policy could sit between any two) captured ONCE as a HIP graph and
    replayed -- what VecFlowEnv.capture does for a learner (examples/train_vec.py)."""
    import torch
    from flow_amd.sim import FlowSim
    spec = c2_spec(R, seed=1000, horizon=10 ** 9)
    sim = FlowSim(spec, precision=precision, device=device.index)
    stream = torch.cuda.Stream(device)
    obs = torch.empty((K, R, sim.obs_dim), dtype=torch.float32, device=device)
    rew = torch.empty((K, R), dtype=torch.float32, device=device)
    done = torch.empty((K, R), dtype=torch.uint8, device=device)
    with torch.cuda.stream(stream):
        sim.set_stream(stream.cuda_stream)
        sim.reset_dev(obs[0])
        sim.step_dev(obs[0], rew[0], done[0])          # warm-up: lazy host work happens outside the capture
    stream.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        for k in range(K):
            sim.step_dev(obs[k], rew[k], done[k])
    with torch.cuda.stream(stream):                 # CUDAGraph.replay() goes to torch's CURRENT stream
        graph.replay()
        stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(replays):
            graph.replay()
        stream.synchronize()
    dt = time.perf_counter() - t0
    sim.use_own_stream()
    advanced = int(sim.time_counter[0])
    sim.close()
    if advanced != 1 + K * (replays + 1):
        raise RuntimeError("step graph: the simulator advanced %d steps, expected %d" % (advanced, 1 + K * (replays + 1)))
    return {"value": R * K * replays / dt, "unit": "env-steps/s", "steps_per_graph": K, "replays": replays,
            "us_per_step": dt / (K * replays) * 1e6, "dtype": precision,
            "note": "K fs_step_dev launches per HIP-graph replay (one launch per env step, no host in the loop)"}


def generic_kernel_leg(device, R=4096, K=1500):
    """The cliff's height: C2 forced onto the generic k_steps (what a ring with a fail-safe, an odd vehicle count,
    sort_vehicles, track_aux or a non-IDM controller steps on: FLOWSIM_FORCE_GENERIC=1), kernel time of a 1500-step launch."""
    import torch
    from flow_amd.sim import FlowSim
    os.environ["FLOWSIM_FORCE_GENERIC"] = "1"
    try:
        sim = FlowSim(c2_spec(R, seed=1000), precision="f32", device=device.index)
    finally:
        os.environ.pop("FLOWSIM_FORCE_GENERIC")
    sim.set_stream(torch.cuda.current_stream(device).cuda_stream)
    obs = torch.empty((K, R, 44), dtype=torch.float32, device=device)
    rew = torch.empty((K, R), dtype=torch.float32, device=device)
    done = torch.empty((K, R), dtype=torch.uint8, device=device)
    obs0 = torch.empty((R, 44), dtype=torch.float32, device=device)
    sim.reset_dev(obs0)
    sim.rollout_dev(K, obs, rew, done)
    torch.cuda.synchronize(device)
    ms = []
    for _ in range(3):
        sim.reset_dev(obs0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sim.rollout_dev(K, obs, rew, done)
        e1.record()
        torch.cuda.synchronize(device)
        ms.append(e0.elapsed_time(e1))
    kernel = sim.last_kernel
    sim.close()
    t = float(np.median(ms)) * 1e-3
    return {"value": R * K / t, "unit": "env-steps/s (kernel time of a 1500-step launch)", "kernel": kernel, "dtype": "f32",
            "median_launch_ms": t * 1e3, "note": "C2 on the generic step kernel (FLOWSIM_FORCE_GENERIC=1)"}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes (nothing in THIS
    process has touched the GPU), fail loudly when the node has fewer devices, exit with the worst child code."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()          # counting devices does not initialise the GPU
    if have < args.gpus:
        raise SystemExit("bench.py --gpus %d: this node exposes %d GPU(s)" % (args.gpus, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for pr in procs:
        rc = max(rc, abs(pr.wait()))
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30000)
    ap.add_argument("--warmup", type=int, default=3000)
    ap.add_argument("--replicas", type=int, default=4096, help="replicas per GPU (weak scaling) / in total (strong)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): --replicas per GPU; strong: --replicas in total, split over the ranks "
                         "(BASELINE's '4096 replicas, whole node')")
    ap.add_argument("--fragment", type=int, default=1500, help="env steps per rollout launch")
    ap.add_argument("--precision", default="mixed", choices=["f32", "f64", "mixed"],
                    help="mixed (default): float64 state, float32 controller -- the precision that holds the 1e-4 "
                         "trajectory bar; f32: the float32 bit-twin; f64: the reference's arithmetic")
    ap.add_argument("--no-extras", action="store_true", help="skip every leg but the headline and its roofline")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)

    # stdout carries exactly ONE line (the JSON); native libraries print banners there (RCCL's
    # version block), so file descriptor 1 points at stderr until the result is written
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the simulation path is HIP-only (no CPU fallback)")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: LOCAL_RANK %d but only %d GPU(s) visible" % (local_rank, torch.cuda.device_count()))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    # every launch of this process (simulator, events, collectives) goes to one explicit stream
    torch.cuda.set_stream(torch.cuda.Stream(device))
    # BENCH_FORCE_DIST=1 takes the N > 1 code path (RCCL init, per-fragment all-gather, MAX over ranks)
    # even with one rank, so the distributed plumbing can be exercised on a one-GPU box
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        world = dist.get_world_size()          # what RCCL actually formed

    if args.scaling == "strong":
        from flow_amd.dist import shard_range
        lo, hi = shard_range(args.replicas, rank, world)
        R, offset = hi - lo, lo
        if R < 1 or args.replicas % world:
            raise SystemExit("bench.py --scaling strong: %d replicas do not split evenly over %d ranks" % (args.replicas, world))
    else:
        R, offset = args.replicas, rank * args.replicas
    spec = c2_spec(R, seed=1000 + rank)
    spec["replica_offset"] = offset            # global replica ids (noise streams do not depend on the sharding)
    runner = Runner(spec, args.precision, device, args.fragment)

    gather = None
    if use_dist:
        from flow_amd.dist import ObservationGather
        gather = ObservationGather(R, runner.obs_dim, world, device)

    last_local = {}

    def after_fragment(o, r, d):
        if gather is not None:
            gather.launch(o, r, d)       # overlaps the next fragment; the learner reads gather.result() one fragment late
            last_local["row"] = (o, r, d)

    def barrier():
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(device)

    runner.run(args.warmup, after_fragment=after_fragment)
    if gather is not None:
        gather.wait()
    barrier()
    t0 = time.perf_counter()
    runner.run(args.steps, record=True, after_fragment=after_fragment)
    if gather is not None:
        gather.wait()
    barrier()
    elapsed = time.perf_counter() - t0
    gather_check = None
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the learner's view of this rank's block must be what the rank produced (the last fragment's final step)
        go, gr, gd = gather.result()
        lo_, ro_, do_ = last_local["row"]
        blk = slice(rank * R, (rank + 1) * R)
        ok = bool(torch.equal(go[blk], lo_) and torch.equal(gr[blk], ro_) and torch.equal(gd[blk], do_ != 0))
        tt = torch.tensor([1.0 if ok else 0.0], device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MIN)
        gather_check = {"gathered_rows": int(go.shape[0]), "rank_blocks_equal_local": bool(tt.item() > 0.5)}

    N = spec["num_vehicles"]
    total_R = args.replicas if args.scaling == "strong" else world * R
    out = {"metric": "env-steps/sec", "value": total_R * args.steps / elapsed, "unit": "env-steps/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
           "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
           "config": {"workload": "C2: RingNetwork 230 m, 22 IDM vehicles (speed_mode 'aggressive': the rollout "
                                  "kernels' configuration class), %d replicas per GPU, AccelEnv obs [R,44] + reward + "
                                  "done written every step, episodes of 1500 steps" % R,
                      "replicas_per_gpu": R, "vehicles": N, "sim_step": 0.1, "horizon": 1500,
                      "fragment_steps": runner.fragment, "speed_mode": "aggressive",
                      "precision": {"mixed": "float64 state, float32 controller (FS_MIXED)", "f32": "float32",
                                    "f64": "float64"}[args.precision],
                      "replicas_total": total_R,
                      "parallelism": "replica-sharded x%d, obs all-gather per fragment" % world},
           "fragment_latency": None, "gather_check": gather_check,
           "timed_region": {"steps": args.steps, "launches": runner.launches,
                            "note": "value = replicas x steps / wall time of exactly --steps env steps (barrier + "
                                    "synchronize on both sides); a region shorter than one 1500-step fragment is "
                                    "dominated by launch + synchronize latency -- the kernel's own rate is in "
                                    "`roofline` / `rollout_1500`, measured over >= 5 full fragments in this process"}}

    # the latency floor of a fragment: a launch is a chain of `fragment` dependent steps per wave, and 4096 replicas are
    # about one wave per SIMD already -- fewer replicas per GPU (strong scaling of BASELINE's "4096 replicas, whole node")
    # leave the chain as long as it is: the same milliseconds per fragment on every GPU count
    if runner.events:
        e0, e1, n_steps, n_full = runner.events[-1]
        ms = float(e0.elapsed_time(e1)) * runner.fragment / n_steps
        out["fragment_latency"] = {
            "steps": runner.fragment, "replicas_this_gpu": R, "avg_launch_ms": ms, "us_per_step": ms * 1e3 / runner.fragment,
            "full_fragments": n_full,
            "note": "HIP events around the timed region / its fragments (incl. the episode resets between them); one launch "
                    "= %d dependent steps per wave; %d replicas = %.2f waves per SIMD: the launch time is the chain's "
                    "latency, not throughput -- splitting a FIXED number of replicas over more GPUs (strong scaling) cannot "
                    "shorten it, adding replicas per GPU (weak scaling) is what scales"
                    % (runner.fragment, R, R / 4.0 / 1024.0)}

    # ---- roofline of the dominant kernel: ALWAYS from >= 5 full 1500-step launches of this process (HIP events
    # on the kernel's stream), whatever --steps was; plus the parity figure of the reported dtype
    legs = {}
    if rank == 0:
        legs[args.precision] = rollout_1500_leg(device, args.precision, R, check_parity=True)
        out["roofline"] = legs[args.precision]["roofline"]
        out["parity"] = legs[args.precision]["parity"]
    runner.sim.close()

    if world == 1 and rank == 0 and not args.no_extras:
        for other in ("mixed", "f32", "f64"):
            if other not in legs:
                legs[other] = rollout_1500_leg(device, other, R, check_parity=True)
        out["rollout_1500"] = {k: {kk: vv for kk, vv in v.items()} for k, v in legs.items()}
        # the Gym-faithful call pattern: one launch per env step (state round-trips through HBM)
        r1 = Runner(c2_spec(R, seed=1000), args.precision, device, args.fragment)
        n_api = 3000
        r1.run_step_api(300)
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t1 = time.perf_counter()
        e0.record()
        r1.run_step_api(n_api)
        e1.record()
        torch.cuda.synchronize(device)
        dt_api = time.perf_counter() - t1
        per_step_b = 533.0 if args.precision == "f32" else 533.0 + 352.0
        out["step_api"] = {"value": R * n_api / dt_api, "unit": "env-steps/s", "launches": n_api,
                           "us_per_launch_wall": dt_api / n_api * 1e6,
                           "us_per_launch_stream": e0.elapsed_time(e1) * 1e3 / n_api,
                           "achieved_GBs": per_step_b * R * n_api / dt_api / 1e9,
                           "note": "fs_step_dev: 1 launch per env step, 533 B/env-step algorithmic (SURVEY 8d)"}
        r1.sim.close()
        out["step_api_graph"] = step_graph_leg(device, args.precision, R)
        out["ring_default_speed_mode"] = ring_defaults_leg(device)
        out["rl_ring"] = rl_ring_legs(device)
        out["c3_figure_eight"] = c3_leg(device)
        out["c3_figure_eight_po"] = c3_leg(device, po=True)
        out["c3_figure_eight_mixed"] = c3_leg(device, precision="mixed")      # float64 state, float32 car-following models
        out["c3_closed_loop"] = c3_closed_loop_leg(device)                    # the policy in the loop (k_loop_policy)
        out["generic_kernel"] = generic_kernel_leg(device)
        out["c4_bottleneck"] = c4_leg(device)
        out["c4_bottleneck_f64"] = c4_leg(device, precision="f64")            # the reference's arithmetic type
        out["c4_bottleneck_lane_change"] = c4_leg(device, lane_change_mode=1621)   # flow/benchmarks/bottleneck1: M11 on
        out["c4_fullchip"] = c4_leg(device, R=1024)                           # throughput next to the latency of 128 replicas
        out["c5_merge"] = c5_leg(device)
        out["c5_fullchip"] = c5_leg(device, R=4096)
        out["c5_merge_fp16_state"] = c5_leg(device, precision="f16s")
        out["c5_merge_mixed"] = c5_leg(device, precision="mixed")             # float64 kernel, float32 car-following models
        out["c5_merge_f64"] = c5_leg(device, precision="f64")                 # the reference's arithmetic type
        out["cpu_baseline"] = cpu_baseline(lambda r: c2_spec(r, seed=1000))
    elif world == 1 and rank == 0:
        out["cpu_baseline"] = None

    if use_dist:
        # rank 0 has just spent seconds on the roofline / parity launches: the others wait for it here, so that no rank
        # tears its communicator down while a peer is still inside the job
        barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
