"""Multi-agent ring environments (flow/envs/multiagent/ring/): per-agent observation blocks written by the step kernel."""
from flow_amd.envs.multiagent.ring.accel import AdversarialAccelEnv, MultiAgentAccelPOEnv
from flow_amd.envs.multiagent.ring.wave_attenuation import MultiAgentWaveAttenuationPOEnv

__all__ = ["AdversarialAccelEnv", "MultiAgentAccelPOEnv", "MultiAgentWaveAttenuationPOEnv"]
