"""MultiAgentWaveAttenuationPOEnv (flow/envs/multiagent/ring/wave_attenuation.py:128-290): several autonomous vehicles
on ONE ring, each an agent with WaveAttenuationPOEnv's three values (its third one the bumper-to-bumper headway); the
step kernel writes one block per RL vehicle (head ``FS_ENV_WAVE_ATTENUATION_PO_MA``) and WaveAttenuationEnv's reward,
shared by the agents.  (``MultiWaveAttenuationPOEnv`` needs the multi-ring network, which is not built.)"""
import random

import numpy as np

from flow_amd import _lib as L
from flow_amd.envs.base import redraw_ring
from flow_amd.envs.multiagent.base import MultiEnv
from flow_amd.envs.multiagent.ring.accel import _RLColumns
from flow_amd.utils.spaces import Box

ADDITIONAL_ENV_PARAMS = {
    # maximum acceleration of autonomous vehicles
    'max_accel': 1,
    # maximum deceleration of autonomous vehicles
    'max_decel': 1,
    # bounds on the ranges of ring road lengths the autonomous vehicle is trained on
    'ring_length': [220, 270],
}


class MultiAgentWaveAttenuationPOEnv(_RLColumns, MultiEnv):
    FS_ENV = L.FS_ENV_WAVE_ATTENUATION_PO_MA

    def __init__(self, env_params, sim_params, network, simulator='traci'):
        for p in ADDITIONAL_ENV_PARAMS.keys():
            if p not in env_params.additional_params:
                raise KeyError('Environment parameter \'{}\' not supplied'.format(p))
        self._ma_actions = None
        super().__init__(env_params, sim_params, network, simulator)

    def _po_max_length(self):
        return self.env_params.additional_params['ring_length'][1]      # :197

    @property
    def observation_space(self):
        return Box(low=-5, high=5, shape=(3, ), dtype=np.float32)

    @property
    def action_space(self):
        return Box(low=-np.abs(self.env_params.additional_params['max_decel']),
                   high=self.env_params.additional_params['max_accel'], shape=(1, ), dtype=np.float32)

    def _apply_rl_actions(self, rl_actions):
        if rl_actions:
            self._ma_actions = self._row_from_dict(rl_actions)

    def _action_vector(self):
        a, self._ma_actions = self._ma_actions, None
        return a

    def get_state(self):
        row = np.asarray(self._last_obs, dtype=np.float64)
        return {rl_id: row[3 * col:3 * col + 3].copy() for rl_id, col in self._rl_columns().items()}

    def compute_reward(self, rl_actions, **kwargs):
        if rl_actions is None:                                          # the warm-up steps (:216-218)
            return 0
        return {key: float(self._last_reward) for key in self.k.vehicle.get_rl_ids()}

    def additional_command(self):
        for rl_id in self.k.vehicle.get_rl_ids():
            self.k.vehicle.set_observed(self.k.vehicle.get_leader(rl_id) or rl_id)

    def reset(self, new_inflow_rate=None):
        """:253-290: a new ring length per episode (as WaveAttenuationEnv.reset), then the generic reset."""
        bounds = self.env_params.additional_params['ring_length']
        if bounds is not None:
            self.step_counter = 0
            length = random.randint(bounds[0], bounds[1])
            X = redraw_ring(self, length)
            self.sim.set_state(L.FS_FIELD_RING_LENGTH, np.full(1, float(length)))
            self.sim.set_state(L.FS_FIELD_INIT_POS, X)
        return super().reset()
