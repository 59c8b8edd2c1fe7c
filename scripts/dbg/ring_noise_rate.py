"""Rollout rate of a noisy all-IDM ring (22 x IDMController(noise=0.2)) through VecFlowEnv, both speed modes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flow_amd.controllers import ContinuousRouter, IDMController
from flow_amd.core.params import EnvParams, InitialConfig, NetParams, SumoCarFollowingParams, SumoParams, VehicleParams
from flow_amd.envs import AccelEnv, VecFlowEnv
from flow_amd.networks import RingNetwork
from flow_amd.networks.ring import ADDITIONAL_NET_PARAMS
R, K = 4096, 1500
dev = torch.device("cuda", 0)
for mode in ("aggressive", "right_of_way"):
    veh = VehicleParams()
    veh.add(veh_id="idm", acceleration_controller=(IDMController, {"noise": 0.2}), routing_controller=(ContinuousRouter, {}),
            car_following_params=SumoCarFollowingParams(speed_mode=mode), num_vehicles=22)
    fp = dict(exp_tag="ring", env_name=AccelEnv, network=RingNetwork, simulator="traci", sim=SumoParams(render=False, sim_step=0.1),
              env=EnvParams(horizon=1500, additional_params={"max_accel": 3, "max_decel": 3, "target_velocity": 10, "sort_vehicles": False}),
              net=NetParams(additional_params=dict(ADDITIONAL_NET_PARAMS)), veh=veh, initial=InitialConfig(bunching=20))
    vec = VecFlowEnv(fp, num_replicas=R, device=0)
    out = (torch.empty((K, R, vec.obs_dim), device=dev), torch.empty((K, R), device=dev), torch.empty((K, R), dtype=torch.uint8, device=dev))
    vec.reset(); vec.rollout(K, None, out=out); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        vec.reset(); vec.rollout(K, None, out=out)
    torch.cuda.synchronize()
    print(mode, "%.2f G env-steps/s" % (3 * R * K / (time.perf_counter() - t0) / 1e9), vec.sim.last_kernel)
    vec.close()
