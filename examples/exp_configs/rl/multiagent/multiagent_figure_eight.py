"""Figure eight with NUM_AUTOMATED autonomous vehicles spread evenly among 14 vehicles, each its own agent with the
6-value observation of MultiAgentAccelPOEnv (the flow_params of the reference's
examples/exp_configs/rl/multiagent/multiagent_figure_eight.py, without the RLlib policy boilerplate)."""
from flow.controllers import ContinuousRouter, IDMController, RLController
from flow.core.params import (EnvParams, InitialConfig, NetParams, SumoCarFollowingParams, SumoParams,
                              VehicleParams)
from flow.envs.multiagent import MultiAgentAccelPOEnv
from flow.networks import FigureEightNetwork
from flow.networks.figure_eight import ADDITIONAL_NET_PARAMS

HORIZON = 1500
N_ROLLOUTS = 4
N_CPUS = 2
NUM_AUTOMATED = 2
TARGET_VELOCITY = 20
MAX_ACCEL = 3
MAX_DECEL = 3

assert NUM_AUTOMATED in [1, 2, 7, 14], "num_automated must be one of [1, 2, 7 14]"
human_per_automated = (14 - NUM_AUTOMATED) // NUM_AUTOMATED

# groups of humans, each followed by its autonomous vehicle
vehicles = VehicleParams()
for i in range(NUM_AUTOMATED):
    vehicles.add(veh_id="human_{}".format(i), acceleration_controller=(IDMController, {"noise": 0.2}),
                 routing_controller=(ContinuousRouter, {}),
                 car_following_params=SumoCarFollowingParams(speed_mode="obey_safe_speed", decel=1.5),
                 num_vehicles=human_per_automated)
    vehicles.add(veh_id="rl_{}".format(i), acceleration_controller=(RLController, {}),
                 routing_controller=(ContinuousRouter, {}),
                 car_following_params=SumoCarFollowingParams(speed_mode="obey_safe_speed", accel=MAX_ACCEL, decel=MAX_DECEL),
                 num_vehicles=1)

flow_params = dict(
    exp_tag="multiagent_figure_eight",
    env_name=MultiAgentAccelPOEnv,
    network=FigureEightNetwork,
    simulator="traci",
    sim=SumoParams(sim_step=0.1, render=False),
    env=EnvParams(horizon=HORIZON, additional_params={"target_velocity": TARGET_VELOCITY, "max_accel": MAX_ACCEL,
                                                      "max_decel": MAX_DECEL, "sort_vehicles": False}),
    net=NetParams(additional_params=ADDITIONAL_NET_PARAMS.copy()),
    veh=vehicles,
    initial=InitialConfig(),
)
