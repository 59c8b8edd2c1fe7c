"""Velocity-controller descriptors (flow/controllers/velocity_controllers.py)."""
from flow_amd import _lib as L
from flow_amd.controllers.base_controller import BaseController


class FollowerStopper(BaseController):
    """velocity_controllers.py:7-116; always safe_velocity fail-safe with delay 1.0 (:31-33)."""
    FS_ID = L.FS_CTRL_FOLLOWER_STOPPER

    def __init__(self, veh_id, car_following_params, v_des=15, danger_edges=None):
        BaseController.__init__(self, veh_id, car_following_params, delay=1.0, fail_safe='safe_velocity')
        if danger_edges:
            raise NotImplementedError("FollowerStopper(danger_edges=...) is not built")
        if v_des is None:
            raise NotImplementedError("FollowerStopper(v_des=None) is not built")
        self.v_des = v_des
        self.danger_edges = {}

    def fs_params(self):
        return [self.v_des]


class NonLocalFollowerStopper(FollowerStopper):
    """velocity_controllers.py:119-164: v_des is the mean speed of the replica."""
    FS_ID = L.FS_CTRL_NONLOCAL_FOLLOWER_STOPPER


class PISaturation(BaseController):
    """velocity_controllers.py:167-240: PI controller with saturation; the kernel keeps the last
    int(38 / sim_step) - 1 speeds of the vehicle in HBM and the previous command as controller state."""
    FS_ID = L.FS_CTRL_PISATURATION

    def __init__(self, veh_id, car_following_params):
        BaseController.__init__(self, veh_id, car_following_params, delay=1.0)
        self.max_accel = car_following_params.controller_params['accel']
        self.gamma, self.g_l, self.g_u, self.v_catch = 2, 7, 30, 1
