"""Lane-change controller descriptors (flow/controllers/lane_change_controllers.py).
The built networks are single-lane: both are no-ops there."""


class BaseLaneChangeController:
    """flow/controllers/base_lane_changing_controller.py:20-37."""

    def __init__(self, veh_id, lane_change_params=None):
        self.veh_id = veh_id
        self.lane_change_params = lane_change_params or {}


class SimLaneChangeController(BaseLaneChangeController):
    """The simulator decides (lane_change_controllers.py:6-14)."""


class StaticLaneChanger(BaseLaneChangeController):
    """Never changes lane (lane_change_controllers.py:17-26)."""
