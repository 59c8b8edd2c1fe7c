"""Stabilising the ring: one RL vehicle among 21 noisy IDM vehicles on a ring whose length is redrawn from
[220, 270] m at every reset (the experiment of the reference's examples/exp_configs/rl/singleagent/
singleagent_ring.py:14-80, same parameter values).  python examples/train.py singleagent_ring"""
from flow.controllers import ContinuousRouter, IDMController, RLController
from flow.core.params import (EnvParams, InitialConfig, NetParams, SumoCarFollowingParams, SumoParams,
                              VehicleParams)
from flow.envs import WaveAttenuationPOEnv
from flow.networks import RingNetwork

HORIZON = 3000      # steps of one rollout
N_ROLLOUTS = 20     # rollouts per training batch (reference: RLlib train_batch_size = HORIZON * N_ROLLOUTS)
N_CPUS = 2          # rollout workers of the reference; here: unused, the replicas of one GPU handle take their place

vehicles = VehicleParams()
vehicles.add(veh_id="human", acceleration_controller=(IDMController, {"noise": 0.2}),
             car_following_params=SumoCarFollowingParams(min_gap=0), routing_controller=(ContinuousRouter, {}),
             num_vehicles=21)
vehicles.add(veh_id="rl", acceleration_controller=(RLController, {}), routing_controller=(ContinuousRouter, {}),
             num_vehicles=1)

flow_params = dict(
    exp_tag="stabilizing_the_ring",
    env_name=WaveAttenuationPOEnv,
    network=RingNetwork,
    simulator='traci',
    sim=SumoParams(sim_step=0.1, render=False, restart_instance=False),
    env=EnvParams(horizon=HORIZON, warmup_steps=750, clip_actions=False,
                  additional_params={"max_accel": 1, "max_decel": 1, "ring_length": [220, 270]}),
    net=NetParams(additional_params={"length": 260, "lanes": 1, "speed_limit": 30, "resolution": 40}),
    veh=vehicles,
    initial=InitialConfig(),
)
