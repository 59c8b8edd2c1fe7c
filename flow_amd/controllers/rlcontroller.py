"""RLController descriptor (flow/controllers/rlcontroller.py:6-39)."""
from flow_amd import _lib as L
from flow_amd.controllers.base_controller import BaseController


class RLController(BaseController):
    """Marks a vehicle as commanded by the action vector passed to ``step``."""
    FS_ID = L.FS_CTRL_RL

    def __init__(self, veh_id, car_following_params):
        BaseController.__init__(self, veh_id, car_following_params)
