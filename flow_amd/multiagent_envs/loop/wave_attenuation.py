"""Pending-deprecation module (flow/multiagent_envs/loop/wave_attenuation.py): the old import path of flow/envs/multiagent/ring/wave_attenuation.py."""
from flow_amd.utils.flow_warnings import deprecated
from flow_amd.envs.multiagent.ring.wave_attenuation import MultiWaveAttenuationPOEnv as _MultiWaveAttenuationPOEnv
from flow_amd.envs.multiagent.ring.wave_attenuation import ADDITIONAL_ENV_PARAMS  # noqa: F401


@deprecated('flow.multiagent_envs.loop.wave_attenuation', 'flow.envs.multiagent.ring.wave_attenuation.MultiWaveAttenuationPOEnv')
class MultiWaveAttenuationPOEnv(_MultiWaveAttenuationPOEnv):
    """See parent class."""

    pass
