"""WaveAttenuationEnv / WaveAttenuationPOEnv (flow/envs/ring/wave_attenuation.py)."""
import random

import numpy as np
from scipy.optimize import fsolve

from flow_amd import _lib as L
from flow_amd.envs.base import Env, redraw_ring
from flow_amd.utils.spaces import Box

ADDITIONAL_ENV_PARAMS = {
    # maximum acceleration of autonomous vehicles
    'max_accel': 1,
    # maximum deceleration of autonomous vehicles
    'max_decel': 1,
    # bounds on the ranges of ring road lengths the autonomous vehicle is trained on
    'ring_length': [220, 270],
}


def v_eq_max_function(v, *args):
    """Error between the desired and actual equilibrium gap (wave_attenuation.py:33-47)."""
    num_vehicles, length = args
    s_eq_max = (length - num_vehicles * 5) / (num_vehicles - 1)
    v0, s0, tau, gamma = 30, 2, 1, 4
    return s_eq_max - (s0 + v * tau) * (1 - (v / v0) ** gamma) ** -0.5


class WaveAttenuationEnv(Env):
    """Fully observed wave-attenuation environment on a variable-length ring
    (wave_attenuation.py:50-210)."""

    FS_ENV = L.FS_ENV_WAVE_ATTENUATION

    def __init__(self, env_params, sim_params, network, simulator='traci'):
        for p in ADDITIONAL_ENV_PARAMS.keys():
            if p not in env_params.additional_params:
                raise KeyError('Environment parameter \'{}\' not supplied'.format(p))
        super().__init__(env_params, sim_params, network, simulator)

    def _rl_action_order(self):
        return list(self.k.vehicle.get_rl_ids())                 # wave_attenuation.py:110-111

    @property
    def action_space(self):
        return Box(low=-np.abs(self.env_params.additional_params['max_decel']),
                   high=self.env_params.additional_params['max_accel'],
                   shape=(self.initial_vehicles.num_rl_vehicles, ), dtype=np.float32)

    @property
    def observation_space(self):
        self.obs_var_labels = ["Velocity", "Absolute_pos"]
        return Box(low=0, high=1, shape=(2 * self.initial_vehicles.num_vehicles, ), dtype=np.float32)

    def _apply_rl_actions(self, rl_actions):
        self.k.vehicle.apply_acceleration(self.k.vehicle.get_rl_ids(), rl_actions)

    def compute_reward(self, rl_actions, **kwargs):
        """Computed in the step kernel (wave_attenuation.py:113-139)."""
        if rl_actions is None:
            return 0
        return float(self._last_reward)

    def get_state(self):
        return np.array(self._last_obs, dtype=np.float64)

    def additional_command(self):
        if self.k.vehicle.num_rl_vehicles > 0:
            for veh_id in self.k.vehicle.get_human_ids():
                self.k.vehicle.set_observed(veh_id)

    def reset(self):
        """wave_attenuation.py:157-210: draw a ring length, re-place the vehicles with
        InitialConfig(bunching=50, min_gap=0), then the generic reset (incl. warm-up)."""
        if self.env_params.additional_params['ring_length'] is None:
            return super().reset()
        self.step_counter = 0
        length = random.randint(self.env_params.additional_params['ring_length'][0],
                                self.env_params.additional_params['ring_length'][1])
        X = redraw_ring(self, length)
        self.v_eq_max = fsolve(v_eq_max_function, np.array(4), args=(len(self.initial_ids), length))[0]
        self.sim.set_state(L.FS_FIELD_RING_LENGTH, np.full(1, float(length)))
        self.sim.set_state(L.FS_FIELD_INIT_POS, X)
        for i, veh_id in enumerate(self.initial_ids):
            edge, pos = self.k.network.get_edge(float(X[0, i]))
            self.initial_state[veh_id] = (self.k.vehicle.get_type(veh_id), edge, 0, pos,
                                          self.k.vehicle.get_initial_speed(veh_id))
        return super().reset()


class WaveAttenuationPOEnv(WaveAttenuationEnv):
    """Partially observed version: [v_rl/15, (v_lead - v_rl)/15, headway distance / max_length]
    (wave_attenuation.py:213-276); needs exactly one RL vehicle."""

    FS_ENV = L.FS_ENV_WAVE_ATTENUATION_PO

    def _po_max_length(self):
        if self.env_params.additional_params['ring_length'] is not None:
            return self.env_params.additional_params['ring_length'][1]
        return self.k.network.length()

    @property
    def observation_space(self):
        return Box(low=-float('inf'), high=float('inf'), shape=(3, ), dtype=np.float32)

    def additional_command(self):
        rl_id = self.k.vehicle.get_rl_ids()[0]
        lead_id = self.k.vehicle.get_leader(rl_id) or rl_id
        self.k.vehicle.set_observed(lead_id)
