import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_bottleneck_env_gpu as T
from flow_amd.controllers import RLController
from flow_amd.core.params import (EnvParams, InFlows, InitialConfig, NetParams, SumoLaneChangeParams, SumoParams, VehicleParams)
from flow_amd.envs import BottleneckAccelEnv
from flow_amd.networks import BottleneckNetwork
vehicles = VehicleParams()
vehicles.add(veh_id="human", num_vehicles=6)
vehicles.add(veh_id="rl", acceleration_controller=(RLController, {}), lane_change_params=SumoLaneChangeParams(lane_change_mode="no_lc_safe"), num_vehicles=3)
inflow = InFlows()
inflow.add(veh_type="human", edge="1", vehs_per_hour=1800, departLane="random", departSpeed=10)
add = {"max_accel": 3, "max_decel": 3, "lane_change_duration": 5, "disable_tb": True, "disable_ramp_metering": True, "target_velocity": 30, "add_rl_if_exit": True}
net = BottleneckNetwork(name="bottleneck", vehicles=vehicles, initial_config=InitialConfig(spacing="uniform", edges_distribution=["2", "3"]),
                        net_params=NetParams(inflows=inflow, additional_params={"scaling": 1, "speed_limit": 23}))
env = BottleneckAccelEnv(EnvParams(horizon=600, additional_params=add), SumoParams(sim_step=0.5, seed=7), net)
ora = T.accel_oracle(env)
obs = env.reset(); ora.reset()
ref = ora.accel_state(0)
for k in np.flatnonzero(obs != ref):
    print(k, repr(obs[k]), repr(ref[k]), obs[k] - ref[k])
print("x", env.sim.get_state(0)[0][:12], ora.x[0][:12])
print("starts", env.k.vehicle._drop_geometry()[1], [ora.e_start[e] for e in ora.path])
print(env.k.vehicle.lane_neighbour_table("rl_0"))
print(ora._multi_lane(0, ora.slot_of["rl_0"], T.__dict__.get("_x") or {}) if False else "")
