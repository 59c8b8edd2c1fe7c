"""14 IDM cars on the figure eight: right of way at the crossing makes them queue (the experiment of the
reference's examples/exp_configs/non_rl/figure_eight.py)."""
from flow.controllers import IDMController, StaticLaneChanger, ContinuousRouter
from flow.core.params import SumoParams, EnvParams, NetParams
from flow.core.params import VehicleParams, SumoCarFollowingParams
from flow.envs.ring.accel import ADDITIONAL_ENV_PARAMS
from flow.networks.figure_eight import ADDITIONAL_NET_PARAMS
from flow.envs import AccelEnv
from flow.networks import FigureEightNetwork

vehicles = VehicleParams()
vehicles.add(veh_id="idm", acceleration_controller=(IDMController, {}),
             lane_change_controller=(StaticLaneChanger, {}), routing_controller=(ContinuousRouter, {}),
             car_following_params=SumoCarFollowingParams(speed_mode="obey_safe_speed", decel=1.5),
             initial_speed=0, num_vehicles=14)

flow_params = dict(
    exp_tag='figure8', env_name=AccelEnv, network=FigureEightNetwork, simulator='traci',
    sim=SumoParams(render=False),
    env=EnvParams(horizon=1500, additional_params=ADDITIONAL_ENV_PARAMS.copy()),
    net=NetParams(additional_params=ADDITIONAL_NET_PARAMS.copy()),
    veh=vehicles,
)
