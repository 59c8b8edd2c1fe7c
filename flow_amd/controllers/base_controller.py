"""Acceleration-controller descriptors.

In the reference a controller is a Python object whose ``get_action(env)`` is
called once per vehicle per step (flow/controllers/base_controller.py:70-118).
Here a controller class is a *descriptor*: it keeps the reference's constructor
signature and attributes, and names the in-kernel model (``FS_ID``) plus its
packed parameters (``fs_params``).  The arithmetic runs in the HIP step kernel
(flow_amd/csrc/flowsim_kernels.h), never in Python.
"""
from flow_amd import _lib as L

FAIL_SAFES = {None: L.FS_FAILSAFE_NONE, 'instantaneous': L.FS_FAILSAFE_INSTANTANEOUS,
              'safe_velocity': L.FS_FAILSAFE_SAFE_VELOCITY}


class BaseController:
    """Constructor contract of flow/controllers/base_controller.py:42-64."""

    FS_ID = None

    def __init__(self, veh_id, car_following_params, delay=0, fail_safe=None, noise=0):
        self.veh_id = veh_id
        self.accel_noise = noise
        self.delay = delay
        if fail_safe not in FAIL_SAFES:
            raise ValueError("fail_safe must be None, 'instantaneous' or 'safe_velocity'")
        self.fail_safe = fail_safe
        self.max_accel = car_following_params.controller_params['accel']
        self.max_deaccel = abs(car_following_params.controller_params['decel'])
        self.car_following_params = car_following_params

    def fs_params(self):
        """Controller parameters in the order documented in include/flowsim.h."""
        return []

    def get_accel(self, env):
        raise NotImplementedError(
            "flow_amd controllers are evaluated inside the HIP step kernel; read the applied "
            "acceleration with env.k.vehicle.get_accel(veh_id)")

    def get_action(self, env):
        """Last acceleration the kernel applied for this vehicle (None if it was not commanded)."""
        return env.k.vehicle.get_accel(self.veh_id)
