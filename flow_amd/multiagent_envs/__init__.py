"""Old package path flow.multiagent_envs (now flow.envs.multiagent), flow/multiagent_envs/__init__.py; the
traffic-light-grid and highway environments are not part of this package."""
from flow_amd.multiagent_envs.multiagent_env import MultiEnv
from flow_amd.multiagent_envs.loop.wave_attenuation import MultiWaveAttenuationPOEnv
from flow_amd.multiagent_envs.loop.loop_accel import AdversarialAccelEnv

__all__ = ['MultiEnv', 'AdversarialAccelEnv', 'MultiWaveAttenuationPOEnv']
