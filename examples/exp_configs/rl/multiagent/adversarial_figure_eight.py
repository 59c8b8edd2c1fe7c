"""Figure eight with one autonomous vehicle and an adversary that perturbs its accelerations (the flow_params of the
reference's examples/exp_configs/rl/multiagent/adversarial_figure_eight.py, without the RLlib policy boilerplate; the
adversary's reward is the negative of the AV's, so the expected total is zero).  Agents: 'av', 'adversary'."""
from copy import deepcopy

from flow.controllers import ContinuousRouter, IDMController, RLController
from flow.core.params import (EnvParams, InitialConfig, NetParams, SumoCarFollowingParams, SumoParams,
                              VehicleParams)
from flow.envs.multiagent import AdversarialAccelEnv
from flow.networks import FigureEightNetwork
from flow.networks.figure_eight import ADDITIONAL_NET_PARAMS

HORIZON = 1500
N_ROLLOUTS = 4
N_CPUS = 2
N_HUMANS = 13
N_AVS = 1

vehicles = VehicleParams()
for veh_id, controller, count in (("human", (IDMController, {"noise": 0.2}), N_HUMANS), ("rl", (RLController, {}), N_AVS)):
    vehicles.add(veh_id=veh_id, acceleration_controller=controller, routing_controller=(ContinuousRouter, {}),
                 car_following_params=SumoCarFollowingParams(speed_mode="obey_safe_speed"), num_vehicles=count)

flow_params = dict(
    exp_tag="adversarial_figure_eight",
    env_name=AdversarialAccelEnv,
    network=FigureEightNetwork,
    simulator="traci",
    sim=SumoParams(sim_step=0.1, render=False),
    env=EnvParams(horizon=HORIZON, additional_params={"target_velocity": 20, "max_accel": 3, "max_decel": 3,
                                                      "perturb_weight": 0.03, "sort_vehicles": False}),
    net=NetParams(additional_params=deepcopy(ADDITIONAL_NET_PARAMS)),
    veh=vehicles,
    initial=InitialConfig(),
)
