"""Stabilising the ring with NUM_AUTOMATED autonomous vehicles spread evenly among 22 vehicles, each its own agent with
the 3-value observation of MultiAgentWaveAttenuationPOEnv; the ring length is redrawn from [220, 270] m at every reset
(the flow_params of the reference's examples/exp_configs/rl/multiagent/multiagent_ring.py, same parameter values, without
the RLlib policy boilerplate)."""
from flow.controllers import ContinuousRouter, IDMController, RLController
from flow.core.params import (EnvParams, InitialConfig, NetParams, SumoCarFollowingParams, SumoParams,
                              VehicleParams)
from flow.envs.multiagent import MultiAgentWaveAttenuationPOEnv
from flow.networks import RingNetwork

HORIZON = 3000
N_ROLLOUTS = 20
N_CPUS = 2
NUM_AUTOMATED = 2       # at most 22

# every autonomous vehicle is followed by its share of the humans
vehicles = VehicleParams()
humans_left = 22 - NUM_AUTOMATED
for i in range(NUM_AUTOMATED):
    vehicles.add(veh_id="rl_{}".format(i), acceleration_controller=(RLController, {}),
                 routing_controller=(ContinuousRouter, {}), num_vehicles=1)
    share = round(humans_left / (NUM_AUTOMATED - i))
    humans_left -= share
    vehicles.add(veh_id="human_{}".format(i), acceleration_controller=(IDMController, {"noise": 0.2}),
                 car_following_params=SumoCarFollowingParams(min_gap=0), routing_controller=(ContinuousRouter, {}),
                 num_vehicles=share)

flow_params = dict(
    exp_tag="multiagent_ring",
    env_name=MultiAgentWaveAttenuationPOEnv,
    network=RingNetwork,
    simulator="traci",
    sim=SumoParams(sim_step=0.1, render=False, restart_instance=False),
    env=EnvParams(horizon=HORIZON, warmup_steps=750, clip_actions=False,
                  additional_params={"max_accel": 1, "max_decel": 1, "ring_length": [220, 270]}),
    net=NetParams(additional_params={"length": 260, "lanes": 1, "speed_limit": 30, "resolution": 40}),
    veh=vehicles,
    initial=InitialConfig(),
)
