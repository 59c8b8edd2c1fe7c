"""Environments built on the GPU step loop (names as in flow/envs/__init__.py)."""
from flow_amd.envs.base import Env
from flow_amd.envs.ring.accel import AccelEnv
from flow_amd.envs.ring.lane_change_accel import LaneChangeAccelEnv, LaneChangeAccelPOEnv
from flow_amd.envs.ring.wave_attenuation import WaveAttenuationEnv, WaveAttenuationPOEnv
from flow_amd.envs.bottleneck import BottleneckAccelEnv, BottleneckDesiredVelocityEnv, BottleneckEnv
from flow_amd.envs.merge import MergePOEnv
from flow_amd.envs.test import TestEnv
from flow_amd.envs.vec import VecFlowEnv

__all__ = ['Env', 'AccelEnv', 'LaneChangeAccelEnv', 'LaneChangeAccelPOEnv', 'WaveAttenuationEnv', 'WaveAttenuationPOEnv', 'MergePOEnv',
           'BottleneckEnv', 'BottleneckAccelEnv', 'BottleneckDesiredVelocityEnv',
           'TestEnv', 'VecFlowEnv']
