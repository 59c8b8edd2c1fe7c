"""Pending-deprecation module (flow/envs/bottleneck_env.py): the old import path of flow/envs/bottleneck.py."""
from flow_amd.utils.flow_warnings import deprecated
from flow_amd.envs.bottleneck import BottleneckEnv as _BottleneckEnv
from flow_amd.envs.bottleneck import BottleneckAccelEnv as _BottleneckAccelEnv
from flow_amd.envs.bottleneck import BottleneckDesiredVelocityEnv as _BottleneckDesiredVelocityEnv


@deprecated('flow.envs.bottleneck_env', 'flow.envs.bottleneck.BottleneckEnv')
class BottleneckEnv(_BottleneckEnv):
    """See parent class."""

    pass


@deprecated('flow.envs.bottleneck_env', 'flow.envs.bottleneck.BottleneckAccelEnv')
class BottleNeckAccelEnv(_BottleneckAccelEnv):
    """See parent class."""

    pass


@deprecated('flow.envs.bottleneck_env', 'flow.envs.bottleneck.BottleneckDesiredVelocityEnv')
class DesiredVelocityEnv(_BottleneckDesiredVelocityEnv):
    """See parent class."""

    pass
