"""Figure eight with one RL vehicle among 13 noisy IDM vehicles (the experiment of the reference's
examples/exp_configs/rl/singleagent/singleagent_figure_eight.py:17-82, same parameter values; BASELINE's C3 population).
python examples/train.py singleagent_figure_eight"""
from flow.controllers import ContinuousRouter, IDMController, RLController
from flow.core.params import (EnvParams, InitialConfig, NetParams, SumoCarFollowingParams, SumoParams,
                              VehicleParams)
from flow.envs import AccelEnv
from flow.networks import FigureEightNetwork
from flow.networks.figure_eight import ADDITIONAL_NET_PARAMS

HORIZON = 1500
N_ROLLOUTS = 20
N_CPUS = 2

vehicles = VehicleParams()
vehicles.add(veh_id='human', acceleration_controller=(IDMController, {'noise': 0.2}),
             routing_controller=(ContinuousRouter, {}),
             car_following_params=SumoCarFollowingParams(speed_mode="obey_safe_speed", decel=1.5), num_vehicles=13)
vehicles.add(veh_id='rl', acceleration_controller=(RLController, {}), routing_controller=(ContinuousRouter, {}),
             car_following_params=SumoCarFollowingParams(speed_mode="obey_safe_speed", decel=1.5), num_vehicles=1)

flow_params = dict(
    exp_tag='singleagent_figure_eight',
    env_name=AccelEnv,
    network=FigureEightNetwork,
    simulator='traci',
    sim=SumoParams(sim_step=0.1, render=False),
    env=EnvParams(horizon=HORIZON,
                  additional_params={'target_velocity': 20, 'max_accel': 3, 'max_decel': 3, 'sort_vehicles': False}),
    net=NetParams(additional_params=ADDITIONAL_NET_PARAMS.copy()),
    veh=vehicles,
    initial=InitialConfig(),
)
