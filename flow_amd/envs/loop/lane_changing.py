"""Pending-deprecation module (flow/envs/loop/lane_changing.py): the old import path of flow/envs/ring/lane_change_accel.py."""
from flow_amd.utils.flow_warnings import deprecated
from flow_amd.envs.ring.lane_change_accel import LaneChangeAccelEnv as _LaneChangeAccelEnv
from flow_amd.envs.ring.lane_change_accel import LaneChangeAccelPOEnv as _LaneChangeAccelPOEnv


@deprecated('flow.envs.loop.lane_changing', 'flow.envs.ring.lane_change_accel.LaneChangeAccelEnv')
class LaneChangeAccelEnv(_LaneChangeAccelEnv):
    """See parent class."""

    pass


@deprecated('flow.envs.loop.lane_changing', 'flow.envs.ring.lane_change_accel.LaneChangeAccelPOEnv')
class LaneChangeAccelPOEnv(_LaneChangeAccelPOEnv):
    """See parent class."""

    pass
