"""Congestion forming at a lane drop (the experiment of the reference's examples/exp_configs/non_rl/bottleneck.py):
2300 veh/h enter four lanes that narrow to two and then to one; toll booth and ramp meter are off.  The vehicles'
lane_change_mode (1621) lets the simulator change lanes: flow_amd runs its simplified lane-change model for them
(docs/HISTORY.md M11 -- not SUMO's LC2013).  Written against the `flow` names; examples/simulate.py maps them."""
from flow.controllers import ContinuousRouter, SimLaneChangeController
from flow.core.params import (EnvParams, InFlows, InitialConfig, NetParams, SumoCarFollowingParams,
                              SumoLaneChangeParams, SumoParams, TrafficLightParams, VehicleParams)
from flow.envs import BottleneckEnv
from flow.networks import BottleneckNetwork

SCALING = 1
DISABLE_TB = True
DISABLE_RAMP_METER = True
INFLOW = 2300
HORIZON = 1000

vehicles = VehicleParams()
vehicles.add(veh_id="human", lane_change_controller=(SimLaneChangeController, {}),
             routing_controller=(ContinuousRouter, {}), car_following_params=SumoCarFollowingParams(speed_mode=25),
             lane_change_params=SumoLaneChangeParams(lane_change_mode=1621), num_vehicles=1)

inflow = InFlows()
inflow.add(veh_type="human", edge="1", vehsPerHour=INFLOW, departLane="random", departSpeed=10)

traffic_lights = TrafficLightParams()

flow_params = dict(
    exp_tag='bay_bridge_toll',
    env_name=BottleneckEnv,
    network=BottleneckNetwork,
    simulator='traci',
    sim=SumoParams(sim_step=0.5, render=False, overtake_right=False, restart_instance=False),
    env=EnvParams(horizon=HORIZON,
                  additional_params={"target_velocity": 40, "max_accel": 1, "max_decel": 1, "lane_change_duration": 5,
                                     "add_rl_if_exit": False, "disable_tb": DISABLE_TB,
                                     "disable_ramp_metering": DISABLE_RAMP_METER}),
    net=NetParams(inflows=inflow, additional_params={"scaling": SCALING, "speed_limit": 23}),
    veh=vehicles,
    initial=InitialConfig(spacing="random", min_gap=5, lanes_distribution=float("inf"),
                          edges_distribution=["2", "3", "4", "5"]),
    tls=traffic_lights,
)
