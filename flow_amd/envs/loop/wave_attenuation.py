"""Pending-deprecation module (flow/envs/loop/wave_attenuation.py): the old import path of flow/envs/ring/wave_attenuation.py."""
from flow_amd.utils.flow_warnings import deprecated
from flow_amd.envs.ring.wave_attenuation import WaveAttenuationEnv as _WaveAttenuationEnv
from flow_amd.envs.ring.wave_attenuation import WaveAttenuationPOEnv as _WaveAttenuationPOEnv


@deprecated('flow.envs.loop.wave_attenuation', 'flow.envs.ring.wave_attenuation.WaveAttenuationEnv')
class WaveAttenuationEnv(_WaveAttenuationEnv):
    """See parent class."""

    pass


@deprecated('flow.envs.loop.wave_attenuation', 'flow.envs.ring.wave_attenuation.WaveAttenuationPOEnv')
class WaveAttenuationPOEnv(_WaveAttenuationPOEnv):
    """See parent class."""

    pass
