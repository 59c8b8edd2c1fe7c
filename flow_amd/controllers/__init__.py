"""Controller descriptors with the names of flow/controllers/__init__.py."""
from flow_amd.controllers.rlcontroller import RLController
from flow_amd.controllers.base_controller import BaseController
from flow_amd.controllers.car_following_models import CFMController, BCMController, OVMController, \
    LinearOVM, IDMController, SimCarFollowingController, LACController, GippsController
from flow_amd.controllers.velocity_controllers import FollowerStopper, PISaturation, NonLocalFollowerStopper
from flow_amd.controllers.lane_change_controllers import BaseLaneChangeController, StaticLaneChanger, \
    SimLaneChangeController
from flow_amd.controllers.routing_controllers import BaseRouter, ContinuousRouter
from flow_amd.controllers.compiled import CompiledController

__all__ = [
    "RLController", "BaseController", "BaseLaneChangeController", "BaseRouter", "CFMController",
    "BCMController", "OVMController", "LinearOVM", "IDMController", "SimCarFollowingController",
    "FollowerStopper", "PISaturation", "StaticLaneChanger", "SimLaneChangeController", "ContinuousRouter",
    "LACController", "GippsController", "NonLocalFollowerStopper", "CompiledController"
]
