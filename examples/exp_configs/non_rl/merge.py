"""Open merge with human drivers only (the experiment of the reference's examples/exp_configs/non_rl/merge.py):
perturbations grow upstream of the on-ramp and leave the network with the traffic.  Written against the `flow`
names; examples/simulate.py maps them onto flow_amd."""
from flow.controllers import IDMController
from flow.core.params import (EnvParams, InFlows, InitialConfig, NetParams, SumoCarFollowingParams, SumoParams,
                              VehicleParams)
from flow.envs.merge import ADDITIONAL_ENV_PARAMS, MergePOEnv
from flow.networks import MergeNetwork

FLOW_RATE = 2000          # vehicles per hour entering on the highway

vehicles = VehicleParams()
vehicles.add(veh_id="human", acceleration_controller=(IDMController, {"noise": 0.2}),
             car_following_params=SumoCarFollowingParams(speed_mode="obey_safe_speed"), num_vehicles=5)

inflow = InFlows()
inflow.add(veh_type="human", edge="inflow_highway", vehs_per_hour=FLOW_RATE, departLane="free", departSpeed=10)
inflow.add(veh_type="human", edge="inflow_merge", vehs_per_hour=100, departLane="free", departSpeed=7.5)

flow_params = dict(
    exp_tag='merge-baseline',
    env_name=MergePOEnv,
    network=MergeNetwork,
    simulator='traci',
    sim=SumoParams(render=True, emission_path="./data/", sim_step=0.2, restart_instance=False),
    env=EnvParams(horizon=3600, additional_params=ADDITIONAL_ENV_PARAMS, sims_per_step=5, warmup_steps=0),
    net=NetParams(inflows=inflow,
                  additional_params={"merge_length": 100, "pre_merge_length": 500, "post_merge_length": 100,
                                     "merge_lanes": 1, "highway_lanes": 1, "speed_limit": 30}),
    veh=vehicles,
    initial=InitialConfig(spacing="uniform", perturbation=5.0),
)
