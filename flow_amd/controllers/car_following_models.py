"""Car-following controller descriptors (flow/controllers/car_following_models.py)."""
from flow_amd import _lib as L
from flow_amd.controllers.base_controller import BaseController


class CFMController(BaseController):
    """car_following_models.py:17-88."""
    FS_ID = L.FS_CTRL_CFM

    def __init__(self, veh_id, car_following_params, k_d=1, k_v=1, k_c=1, d_des=1, v_des=8, time_delay=0.0,
                 noise=0, fail_safe=None):
        BaseController.__init__(self, veh_id, car_following_params, delay=time_delay, fail_safe=fail_safe,
                                noise=noise)
        self.k_d, self.k_v, self.k_c, self.d_des, self.v_des = k_d, k_v, k_c, d_des, v_des

    def fs_params(self):
        return [self.k_d, self.k_v, self.k_c, self.d_des, self.v_des]


class BCMController(CFMController):
    """car_following_models.py:91-176 (same parameters as CFM, reads the follower too)."""
    FS_ID = L.FS_CTRL_BCM


class LACController(BaseController):
    """car_following_models.py:179-245."""
    FS_ID = L.FS_CTRL_LAC

    def __init__(self, veh_id, car_following_params, k_1=0.3, k_2=0.4, h=1, tau=0.1, a=0, time_delay=0.0,
                 noise=0, fail_safe=None):
        BaseController.__init__(self, veh_id, car_following_params, delay=time_delay, fail_safe=fail_safe,
                                noise=noise)
        self.k_1, self.k_2, self.h, self.tau, self.a = k_1, k_2, h, tau, a
        if a != 0:
            raise NotImplementedError("LACController: a non-zero initial a is not supported")

    def fs_params(self):
        return [self.k_1, self.k_2, self.h, self.tau]


class OVMController(BaseController):
    """car_following_models.py:248-328."""
    FS_ID = L.FS_CTRL_OVM

    def __init__(self, veh_id, car_following_params, alpha=1, beta=1, h_st=2, h_go=15, v_max=30, time_delay=0,
                 noise=0, fail_safe=None):
        BaseController.__init__(self, veh_id, car_following_params, delay=time_delay, fail_safe=fail_safe,
                                noise=noise)
        self.v_max, self.alpha, self.beta, self.h_st, self.h_go = v_max, alpha, beta, h_st, h_go

    def fs_params(self):
        return [self.alpha, self.beta, self.h_st, self.h_go, self.v_max]


class LinearOVM(BaseController):
    """car_following_models.py:331-397."""
    FS_ID = L.FS_CTRL_LINEAR_OVM

    def __init__(self, veh_id, car_following_params, v_max=30, adaptation=0.65, h_st=5, time_delay=0.0, noise=0,
                 fail_safe=None):
        BaseController.__init__(self, veh_id, car_following_params, delay=time_delay, fail_safe=fail_safe,
                                noise=noise)
        self.v_max, self.adaptation, self.h_st = v_max, adaptation, h_st

    def fs_params(self):
        return [self.v_max, self.adaptation, self.h_st]


class IDMController(BaseController):
    """car_following_models.py:400-482."""
    FS_ID = L.FS_CTRL_IDM

    def __init__(self, veh_id, v0=30, T=1, a=1, b=1.5, delta=4, s0=2, time_delay=0.0, noise=0, fail_safe=None,
                 car_following_params=None):
        BaseController.__init__(self, veh_id, car_following_params, delay=time_delay, fail_safe=fail_safe,
                                noise=noise)
        self.v0, self.T, self.a, self.b, self.delta, self.s0 = v0, T, a, b, delta, s0

    def fs_params(self):
        return [self.v0, self.T, self.a, self.b, self.delta, self.s0]


class SimCarFollowingController(BaseController):
    """car_following_models.py:485-497: never commanded; the simulator's own model drives."""
    FS_ID = L.FS_CTRL_SIM


class GippsController(BaseController):
    """car_following_models.py:500-582."""
    FS_ID = L.FS_CTRL_GIPPS

    def __init__(self, veh_id, car_following_params=None, v0=30, acc=1.5, b=-1, b_l=-1, s0=2, tau=1, delay=0,
                 noise=0, fail_safe=None):
        BaseController.__init__(self, veh_id, car_following_params, delay=delay, fail_safe=fail_safe, noise=noise)
        self.v_desired, self.acc, self.b, self.b_l, self.s0, self.tau = v0, acc, b, b_l, s0, tau

    def fs_params(self):
        return [self.v_desired, self.acc, self.b, self.b_l, self.s0, self.tau]
