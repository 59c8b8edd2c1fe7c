"""Pending-deprecation module (flow/envs/loop/loop_accel.py): the old import path of flow/envs/ring/accel.py."""
from flow_amd.utils.flow_warnings import deprecated
from flow_amd.envs.ring.accel import AccelEnv as _AccelEnv


@deprecated('flow.envs.loop.loop_accel', 'flow.envs.ring.accel.AccelEnv')
class AccelEnv(_AccelEnv):
    """See parent class."""

    pass
