"""A user-defined acceleration controller (the reference's extension point is a BaseController subclass with a Python
get_accel, flow/controllers/base_controller.py:42-118; here get_accel is a device function compiled into a copy of the
library: flow_amd.controllers.CompiledController).  ``TimeGap`` is a constant-time-gap follower with a damping term on the
follower's speed, written once as C++ (what the step kernel runs) and once as numpy (what a test's oracle runs).

    python examples/compiled_controller.py          # 4096 replicas of a 22-vehicle ring, 1500 steps, on the device
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from flow_amd.controllers import CompiledController  # noqa: E402


class TimeGap(CompiledController):
    SOURCE = """
        const T gap_err = h - p[0] * v - p[1];
        const T a = p[2] * gap_err + p[3] * (v_lead - v) + p[4] * (v - v_follow);
        const T lim = tmin(tmax(a, T(0) - max_accel), max_accel);
        return has_lead ? lim : max_accel;
    """

    def __init__(self, veh_id, car_following_params, t_gap=1.2, s0=2.0, k_gap=0.3, k_speed=0.6, k_follow=0.05, **kw):
        CompiledController.__init__(self, veh_id, car_following_params, params=[t_gap, s0, k_gap, k_speed, k_follow], **kw)


def time_gap_numpy(v, v_lead, h, has_lead, v_follow, h_follow, dt, max_accel, p, dtype):
    """The same law in numpy, operation by operation (for an oracle: spec['user_controller_numpy'])."""
    import numpy as np
    T = np.dtype(dtype).type
    gap_err = h - p[0] * v - p[1]
    a = p[2] * gap_err + p[3] * (v_lead - v) + p[4] * (v - v_follow)
    lim = np.minimum(np.maximum(a, T(0) - max_accel), max_accel)
    return np.where(has_lead, lim, max_accel).astype(dtype)


if __name__ == "__main__":
    import time
    import torch
    from flow_amd.controllers import ContinuousRouter
    from flow_amd.core.params import EnvParams, InitialConfig, NetParams, SumoParams, VehicleParams
    from flow_amd.envs import AccelEnv, VecFlowEnv
    from flow_amd.networks import RingNetwork
    veh = VehicleParams()
    veh.add("gap", acceleration_controller=(TimeGap, {"t_gap": 1.0, "noise": 0.1}), routing_controller=(ContinuousRouter, {}),
            num_vehicles=22)
    fp = dict(exp_tag="time_gap_ring", env_name=AccelEnv, network=RingNetwork, simulator="traci",
              sim=SumoParams(sim_step=0.1), initial=InitialConfig(bunching=20), veh=veh,
              env=EnvParams(horizon=1500, additional_params={"max_accel": 1, "max_decel": 1, "target_velocity": 10,
                                                             "sort_vehicles": False}),
              net=NetParams(additional_params={"length": 230, "lanes": 1, "speed_limit": 30, "resolution": 40}))
    vec = VecFlowEnv(fp, num_replicas=4096, device=0)          # builds (once) and loads the library copy with TimeGap
    vec.reset()
    vec.rollout(1500)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    obs, rew, done = vec.rollout(1500)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("TimeGap ring: %.2f G env-steps/s on %s, mean reward %.3f" % (4096 * 1500 / dt / 1e9, vec.sim.last_kernel,
                                                                           float(rew.mean())))
    vec.close()
