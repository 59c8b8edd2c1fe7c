"""Multi-agent open merge (the flow_params of the reference's examples/exp_configs/rl/multiagent/
multiagent_merge.py, without the RLlib policy boilerplate): 10 % of the highway inflow are RL vehicles, each an
agent with a 5-value observation.  Use with flow_amd.envs.VecFlowEnv for batched rollouts."""
from flow.controllers import IDMController, RLController
from flow.core.params import (EnvParams, InFlows, InitialConfig, NetParams, SumoCarFollowingParams, SumoParams,
                              VehicleParams)
from flow.envs.multiagent import MultiAgentMergePOEnv
from flow.networks import MergeNetwork
from flow.networks.merge import ADDITIONAL_NET_PARAMS

HORIZON = 600
FLOW_RATE = 2000
RL_PENETRATION = 0.1

additional_net_params = ADDITIONAL_NET_PARAMS.copy()
additional_net_params.update(merge_lanes=1, highway_lanes=1, pre_merge_length=500)

vehicles = VehicleParams()
vehicles.add(veh_id="human", acceleration_controller=(IDMController, {"noise": 0.2}),
             car_following_params=SumoCarFollowingParams(speed_mode="obey_safe_speed"), num_vehicles=5)
vehicles.add(veh_id="rl", acceleration_controller=(RLController, {}),
             car_following_params=SumoCarFollowingParams(speed_mode="obey_safe_speed"), num_vehicles=0)

inflow = InFlows()
inflow.add(veh_type="human", edge="inflow_highway", vehs_per_hour=(1 - RL_PENETRATION) * FLOW_RATE,
           departLane="free", departSpeed=10)
inflow.add(veh_type="rl", edge="inflow_highway", vehs_per_hour=RL_PENETRATION * FLOW_RATE,
           departLane="free", departSpeed=10)
inflow.add(veh_type="human", edge="inflow_merge", vehs_per_hour=100, departLane="free", departSpeed=7.5)

flow_params = dict(
    exp_tag="multiagent_merge",
    env_name=MultiAgentMergePOEnv,
    network=MergeNetwork,
    simulator='traci',
    sim=SumoParams(sim_step=0.2, render=False, restart_instance=True),
    env=EnvParams(horizon=HORIZON, sims_per_step=5, warmup_steps=0,
                  additional_params={"max_accel": 1.5, "max_decel": 1.5, "target_velocity": 20}),
    net=NetParams(inflows=inflow, additional_params=additional_net_params),
    veh=vehicles,
    initial=InitialConfig(),
)
