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

Prints ONE JSON line (rank 0).  Extra keys: `roofline` (dominant kernel =
k_steps, per-launch HIP-event timing), `cpu_baseline` (oracle C port on the host
cores, bounded sample, rank 0, N=1 only), `c4_bottleneck` / `c5_merge` (open-network kernel, informational), `step_api` (one launch per env step,
the Gym-faithful call pattern), `f64` (same workload in float64).
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

    def run(self, steps, record=False, after_fragment=None):
        torch = self.torch
        left = steps
        while left > 0:
            k = min(self.fragment, left, self.horizon - self.t_in_episode)
            if record:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            self.sim.rollout_dev(k, self.obs, self.rew, self.done, obs_every_step=True)
            if record:
                e1.record()
                self.events.append((e0, e1, k))
            if after_fragment is not None:
                after_fragment(self.obs[k - 1], self.rew[k - 1], self.done[k - 1])
            self.t_in_episode += k
            left -= k
            if self.t_in_episode >= self.horizon:          # every replica is done: Env.reset
                self.sim.reset_dev(self.obs0)
                self.t_in_episode = 0

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


def c3_leg(device, R=4096, steps=3000, po=False):
    """BASELINE configs[2] (informational, not the headline): FigureEightNetwork, 13 noisy IDM + 1 RL vehicle,
    random RL actions from a pre-generated tape, generic k_steps kernel.  ``po=False``: AccelEnv observation (28),
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
              simulator="traci", sim=SumoParams(sim_step=0.1, render=False, seed=7),
              env=EnvParams(horizon=1500, additional_params=add),
              net=NetParams(additional_params=dict(ADDITIONAL_NET_PARAMS)), veh=veh)
    vec = VecFlowEnv(fp, num_replicas=R, device=device.index)
    K = 1500
    gen = torch.Generator(device=device).manual_seed(1)
    tape = (torch.rand((K, R, 1), device=device, generator=gen) * 2 - 1) * 3.0
    out = (torch.empty((K, R, vec.obs_dim), dtype=torch.float32, device=device),
           torch.empty((K, R), dtype=torch.float32, device=device),
           torch.empty((K, R), dtype=torch.uint8, device=device))
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
    vec.close()
    return {"value": R * done_steps / dt, "unit": "env-steps/s", "steps": done_steps, "steps_per_launch": K, "replicas": R,
            "obs_dim": 3 if po else 28, "replicas_crashed_before_horizon": crashed,
            "workload": "C3: FigureEightNetwork r=30, 13 IDM (noise 0.2, obey_safe_speed) + 1 RL, %s, "
                        "random actions; generic kernel" % ("WaveAttenuationPOEnv" if po else "AccelEnv")}


def c5_leg(device, R=1024, env_steps=600):
    """BASELINE configs[4] (informational, not the headline): MergeNetwork pre_merge 500 m, 5 initial humans +
    inflows 1800 / 200 (RL) / 100 veh/h, sim_step 0.2, sims_per_step 5, horizon 600, MultiAgentMergePOEnv head
    (examples/exp_configs/rl/multiagent/multiagent_merge.py); 1024 replicas per GPU = 8192 over 8 GPUs; the
    open-network kernel k_steps_open, 64 vehicle slots per replica.  fp32 state (BASELINE names fp16 state /
    fp32 integrator: the state arrays here are fp32, nothing is stored in fp16)."""
    import torch
    from flow_amd import _lib as L
    from flow_amd.controllers import IDMController, RLController
    from flow_amd.core.params import (EnvParams, InFlows, NetParams, SumoCarFollowingParams, SumoParams,
                                      VehicleParams)
    from flow_amd.envs import VecFlowEnv
    from flow_amd.envs.multiagent import MultiAgentMergePOEnv
    from flow_amd.networks import MergeNetwork
    from flow_amd.networks.merge import ADDITIONAL_NET_PARAMS
    add = dict(ADDITIONAL_NET_PARAMS)
    add["pre_merge_length"] = 500
    veh = VehicleParams()
    veh.add(veh_id="human", acceleration_controller=(IDMController, {"noise": 0.2}),
            car_following_params=SumoCarFollowingParams(speed_mode="obey_safe_speed"), num_vehicles=5)
    veh.add(veh_id="rl", acceleration_controller=(RLController, {}),
            car_following_params=SumoCarFollowingParams(speed_mode="obey_safe_speed"), num_vehicles=0)
    inflow = InFlows()
    inflow.add(veh_type="human", edge="inflow_highway", vehs_per_hour=1800, depart_lane="free", depart_speed=10)
    inflow.add(veh_type="rl", edge="inflow_highway", vehs_per_hour=200, depart_lane="free", depart_speed=10)
    inflow.add(veh_type="human", edge="inflow_merge", vehs_per_hour=100, depart_lane="free", depart_speed=7.5)
    fp = dict(exp_tag="multiagent_merge", env_name=MultiAgentMergePOEnv, network=MergeNetwork, simulator="traci",
              sim=SumoParams(sim_step=0.2, render=False, restart_instance=True, seed=11),
              env=EnvParams(horizon=600, sims_per_step=5, warmup_steps=0,
                            additional_params={"max_accel": 1.5, "max_decel": 1.5, "target_velocity": 20}),
              net=NetParams(inflows=inflow, additional_params=add), veh=veh)
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
    res = {"value": R * K * 5 / dt, "unit": "env-steps/s (simulation sub-steps)", "env_steps": K, "sims_per_step": 5,
           "replicas": R, "gym_steps_per_s": R * K / dt, "obs_dim": vec.obs_dim,
           "vehicles_in_network_mean": float((route >= 0).sum(axis=1).mean()),
           "departed_mean": float(cnt[:, 6].mean()), "arrived_mean": float(cnt[:, 5].mean()),
           "workload": "C5: MergeNetwork pre_merge 500 m, inflows 1800 + 200 RL + 100 veh/h, 64 slots, "
                       "MultiAgentMergePOEnv head, sim_step 0.2 x 5 sub-steps, one 600-step episode; k_steps_open"}
    vec.close()
    return res


def c4_leg(device, R=128, env_steps=1000, slots=256):
    """BASELINE configs[3] (informational, not the headline): BottleneckNetwork scaling 1 (4 -> 2 -> 1 lanes at two
    zipper junctions), inflow 2300 veh/h (10 % RL) with random entry lanes, all vehicles on the SUMO car-following
    model, BottleneckDesiredVelocityEnv head (141 observations, 20 actions), sim_step 0.5, warm-up 40 + horizon 1000
    (examples/exp_configs/rl/singleagent/singleagent_bottleneck.py); 128 replicas per GPU = 1024 over 8 GPUs."""
    import torch
    from flow_amd import _lib as L
    from flow_amd.controllers import ContinuousRouter, RLController, SimLaneChangeController
    from flow_amd.core.params import (EnvParams, InFlows, InitialConfig, NetParams, SumoCarFollowingParams,
                                      SumoLaneChangeParams, SumoParams, VehicleParams)
    from flow_amd.envs import BottleneckDesiredVelocityEnv, VecFlowEnv
    from flow_amd.networks import BottleneckNetwork
    veh = VehicleParams()
    veh.add(veh_id="human", lane_change_controller=(SimLaneChangeController, {}),
            routing_controller=(ContinuousRouter, {}),
            car_following_params=SumoCarFollowingParams(speed_mode="all_checks"),
            lane_change_params=SumoLaneChangeParams(lane_change_mode=0), num_vehicles=1)
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
                                                max_vehicles=slots),
              env=EnvParams(warmup_steps=40, sims_per_step=1, horizon=1000, additional_params=add),
              net=NetParams(inflows=inflow, additional_params={"scaling": 1, "speed_limit": 23}), veh=veh,
              initial=InitialConfig(spacing="uniform", min_gap=5, lanes_distribution=float("inf"),
                                    edges_distribution=["2", "3", "4", "5"]))
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
    res = {"value": R * K / dt, "unit": "env-steps/s", "env_steps": K, "replicas": R, "obs_dim": vec.obs_dim,
           "act_dim": vec.act_dim, "vehicles_in_network_mean": float((route >= 0).sum(axis=1).mean()),
           "departed_mean": float(cnt[:, 6].mean()), "arrived_mean": float(cnt[:, 5].mean()),
           "dropped_at_insertion_mean": float(cnt[:, 7].mean()),
           "outflow_veh_per_hour_mean": float(out[1][-200:].mean().item() * 2000.0),
           "slots": slots,
           "workload": "C4: BottleneckNetwork 4->2->1 lanes, inflow 2300 veh/h (10 %% RL, random lanes), %d slots, "
                       "BottleneckDesiredVelocityEnv head (141 obs / 20 actions), sim_step 0.5, one 1000-step episode "
                       "after 40 warm-up steps, random actions; %s" %
                       (slots, "k_steps_wide<., %d> (one workgroup per replica)" % ((slots + 63) // 64 if slots > 128 else 2)
                        if slots > 64 else "k_steps_open<.,64,4>")}
    vec.close()
    return res


def cpu_baseline(spec_fn, seconds=12.0):
    """Oracle C port (oracle/csim) on the host cores, bounded sample of the same workload."""
    from oracle import cbuild
    cores = host_cores()
    R = 4096
    spec = spec_fn(R)
    sim = cbuild.CRingIDM(spec, np.float32, threads=cores)
    sim.rollout(20, obs_every_step=True)               # warm-up
    chunk, done_steps, t0 = 100, 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:          # ~10-30 s of CPU work; episodes of 1500 steps
        if done_steps % 1500 == 0:
            sim.reset()
        sim.rollout(chunk, obs_every_step=True)
        done_steps += chunk
    dt = time.perf_counter() - t0
    return {"value": R * done_steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "oracle/csim C port (float32, OpenMP over replicas, obs written every step), "
                      "%d replicas x 22 vehicles x %d steps in %.1f s" % (R, done_steps, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30000)
    ap.add_argument("--warmup", type=int, default=3000)
    ap.add_argument("--replicas", type=int, default=4096, help="replicas per GPU")
    ap.add_argument("--fragment", type=int, default=1500, help="env steps per rollout launch")
    ap.add_argument("--precision", default="f32", choices=["f32", "f64", "mixed"])
    ap.add_argument("--no-extras", action="store_true", help="skip cpu_baseline / step_api / f64 legs")
    args = ap.parse_args()

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
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the simulation path is HIP-only (no CPU fallback)")
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
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    R = args.replicas
    spec = c2_spec(R, seed=1000 + rank)
    spec["replica_offset"] = rank * R          # global replica ids (noise streams do not depend on the sharding)
    runner = Runner(spec, args.precision, device, args.fragment)

    gather = None
    if use_dist:
        from flow_amd.dist import ObservationGather
        gather = ObservationGather(R, runner.obs_dim, world, device)

    def after_fragment(o, r, d):
        if gather is not None:
            gather.launch(o, r, d)       # overlaps the next fragment; the learner reads gather.result() one fragment late

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
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel (k_steps), per launch, from HIP events on its stream
    full = [(e0.elapsed_time(e1) * 1e-3, k) for e0, e1, k in runner.events if k == runner.fragment]
    if not full:
        full = [(e0.elapsed_time(e1) * 1e-3, k) for e0, e1, k in runner.events]
    k_launch = full[0][1]
    avg_launch_s = float(np.mean([t for t, _ in full]))
    N = spec["num_vehicles"]
    obs_b = runner.obs_dim * 4 + 4 + 1                    # obs + reward + done, per env-step
    state_b = N * (4 + 4) * 2 + 4 * 2                     # pos+vel read and written once per launch, time counter
    if args.precision != "f32":
        state_b = N * (8 + 8) * 2 + 4 * 2
    bytes_per_launch = R * (k_launch * obs_b + state_b)
    achieved = bytes_per_launch / avg_launch_s / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
    if os.path.exists(tpath) and args.precision == "f32":
        try:
            tj = json.load(open(tpath))
            if tj.get("replicas") == R and tj.get("steps_per_launch") == k_launch:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm",
                "kernel": "fs::k_rollout_idm<%s, 32, true, %s>" % (("float", "true") if args.precision == "f32"
                                                                   else ("double", "false")),
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "bytes_per_launch": bytes_per_launch, "steps_per_launch": k_launch,
                "avg_launch_ms": avg_launch_s * 1e3, "launches_timed": len(full),
                "bytes_per_env_step": obs_b + state_b / k_launch,
                "survey_533B_equiv_GBs": 533.0 * R * k_launch / avg_launch_s / 1e9}

    out = {"metric": "env-steps/sec", "value": world * R * args.steps / elapsed, "unit": "env-steps/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
           "config": {"workload": "C2: RingNetwork 230 m, 22 IDM vehicles, %d replicas per GPU, AccelEnv obs "
                                  "[R,44] + reward + done written every step, episodes of 1500 steps" % R,
                      "replicas_per_gpu": R, "vehicles": N, "sim_step": 0.1, "horizon": 1500,
                      "fragment_steps": runner.fragment,
                      "parallelism": "replica-sharded x%d, obs all-gather per fragment" % world},
           "roofline": roofline}

    if world == 1 and rank == 0 and not args.no_extras:
        # the Gym-faithful call pattern: one launch per env step (state round-trips through HBM)
        n_api = 3000
        runner.run_step_api(300)
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t1 = time.perf_counter()
        e0.record()
        runner.run_step_api(n_api)
        e1.record()
        torch.cuda.synchronize(device)
        dt_api = time.perf_counter() - t1
        per_step_b = 533.0 if args.precision == "f32" else 533.0 + 352.0
        out["step_api"] = {"value": R * n_api / dt_api, "unit": "env-steps/s", "launches": n_api,
                           "us_per_launch_wall": dt_api / n_api * 1e6,
                           "us_per_launch_stream": e0.elapsed_time(e1) * 1e3 / n_api,
                           "achieved_GBs": per_step_b * R * n_api / dt_api / 1e9,
                           "note": "fs_step_dev: 1 launch per env step, 533 B/env-step algorithmic (SURVEY 8d)"}
        # same workload in float64 (the reference's arithmetic type)
        other = "f64" if args.precision == "f32" else "f32"
        r2 = Runner(c2_spec(R, seed=1000), other, device, args.fragment)
        r2.run(1500)
        torch.cuda.synchronize(device)
        t2 = time.perf_counter()
        r2.run(6000)
        torch.cuda.synchronize(device)
        out[other] = {"value": R * 6000 / (time.perf_counter() - t2), "unit": "env-steps/s", "steps": 6000}
        r2.sim.close()
        out["c3_figure_eight"] = c3_leg(device)
        out["c3_figure_eight_po"] = c3_leg(device, po=True)
        out["c4_bottleneck"] = c4_leg(device)
        out["c5_merge"] = c5_leg(device)
        out["cpu_baseline"] = cpu_baseline(lambda r: c2_spec(r, seed=1000))

    runner.sim.close()
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
