"""Pending-deprecation module (flow/multiagent_envs/multiagent_env.py): the old import path of flow/envs/multiagent/base.py."""
from flow_amd.utils.flow_warnings import deprecated
from flow_amd.envs.multiagent.base import MultiEnv as _MultiEnv


@deprecated('flow.multiagent_envs.multiagent_env', 'flow.envs.multiagent.base.MultiEnv')
class MultiEnv(_MultiEnv):
    """See parent class."""

    pass
