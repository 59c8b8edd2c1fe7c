"""Pending-deprecation module (flow/multiagent_envs/loop/loop_accel.py): the old import path of flow/envs/multiagent/ring/accel.py."""
from flow_amd.utils.flow_warnings import deprecated
from flow_amd.envs.multiagent.ring.accel import AdversarialAccelEnv as _AdversarialAccelEnv


@deprecated('flow.multiagent_envs.loop.loop_accel', 'flow.envs.multiagent.ring.accel.AdversarialAccelEnv')
class AdversarialAccelEnv(_AdversarialAccelEnv):
    """See parent class."""

    pass
