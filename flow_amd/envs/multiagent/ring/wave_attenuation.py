"""Multi-agent wave-attenuation environments (flow/envs/multiagent/ring/wave_attenuation.py), host-assembled
observations and rewards over the GPU step loop (see multiagent/ring/accel.py):

* MultiAgentWaveAttenuationPOEnv (:130-312): every RL vehicle on ONE ring is an agent with the 3-value
  WaveAttenuationPOEnv observation; shared reward.
* MultiWaveAttenuationPOEnv (:34-127): the "lord of the rings" version on MultiRingNetwork -- one RL vehicle per
  ring, each ring an agent with its own desired-velocity reward.  Ring r runs as replica r of the ring kernel."""
import random

import numpy as np
from scipy.optimize import fsolve

from flow_amd import _lib as L
from flow_amd.envs.base import redraw_ring
from flow_amd.envs.multiagent.base import MultiEnv
from flow_amd.envs.multiagent.ring.accel import _PendingActions
from flow_amd.envs.ring.wave_attenuation import v_eq_max_function
from flow_amd.utils.spaces import Box

ADDITIONAL_ENV_PARAMS = {
    # maximum acceleration of autonomous vehicles
    'max_accel': 1,
    # maximum deceleration of autonomous vehicles
    'max_decel': 1,
    # bounds on the ranges of ring road lengths the autonomous vehicle is trained on
    'ring_length': [220, 270],
}


class MultiWaveAttenuationPOEnv(_PendingActions, MultiEnv):
    """flow/envs/multiagent/ring/wave_attenuation.py:34-127."""

    FS_ENV = L.FS_ENV_ACCEL

    def __init__(self, env_params, sim_params, network=None, simulator='traci', scenario=None):
        for p in ADDITIONAL_ENV_PARAMS.keys():
            if p not in env_params.additional_params:
                raise KeyError('Environment parameter \'{}\' not supplied'.format(p))
        super().__init__(env_params, sim_params, network, simulator, scenario)

    @property
    def observation_space(self):
        return Box(low=-1, high=1, shape=(3,), dtype=np.float32)

    @property
    def action_space(self):
        num_rings = self.net_params.additional_params['num_rings']
        return Box(low=-np.abs(self.env_params.additional_params['max_decel']),
                   high=self.env_params.additional_params['max_accel'],
                   shape=(int(self.initial_vehicles.num_rl_vehicles / num_rings), ), dtype=np.float32)

    def _action_vector(self):
        """One row of RL commands per ring (= per replica of the handle), rings in id order."""
        pend = self.k.vehicle._pending
        if not pend:
            return None
        rings = int(self.net_params.additional_params['num_rings'])
        return np.array([pend.get(v, 0.0) for v in self._rl_action_order()], dtype=np.float32).reshape(rings, -1)

    def get_state(self):
        obs = {}
        veh = self.k.vehicle
        for rl_id in veh.get_rl_ids():
            lead_id = veh.get_leader(rl_id) or rl_id
            max_speed = 15.
            max_length = self.env_params.additional_params['ring_length'][1]
            obs.update({rl_id: np.array([veh.get_speed(rl_id) / max_speed,
                                         (veh.get_speed(lead_id) - veh.get_speed(rl_id)) / max_speed,
                                         veh.get_headway(rl_id) / max_length])})
        return obs

    def _apply_rl_actions(self, rl_actions):
        if rl_actions:
            self.k.vehicle.apply_acceleration(list(rl_actions.keys()), list(rl_actions.values()))

    def compute_reward(self, rl_actions, **kwargs):
        """:98-124: per agent, the desired-velocity reward over the vehicles on the four EDGES of its ring (a vehicle
        inside one of the 0.1 m junctions is on none of them); the ring number is read out of the vehicle id."""
        if rl_actions is None:
            return {}
        rew = {}
        for rl_id in rl_actions.keys():
            edge_id = rl_id.split('_')[1]
            vehs_on_edge = self.k.vehicle.get_ids_by_edge(self.gen_edges(edge_id))
            vel = np.array([self.k.vehicle.get_speed(veh_id) for veh_id in vehs_on_edge])
            if any(vel < -100) or kwargs['fail']:
                return 0.
            target_vel = self.env_params.additional_params['target_velocity']
            max_cost = np.linalg.norm(np.array([target_vel] * len(vehs_on_edge)))
            cost = np.linalg.norm(vel - target_vel)
            rew[rl_id] = max(max_cost - cost, 0) / max_cost
        return rew

    def additional_command(self):
        for rl_id in self.k.vehicle.get_rl_ids():
            lead_id = self.k.vehicle.get_leader(rl_id) or rl_id
            self.k.vehicle.set_observed(lead_id)

    @staticmethod
    def gen_edges(i):
        """Return the edges corresponding to the rl id."""
        return ['top_{}'.format(i), 'left_{}'.format(i), 'right_{}'.format(i), 'bottom_{}'.format(i)]


class MultiAgentWaveAttenuationPOEnv(_PendingActions, MultiEnv):
    """See module docstring."""

    FS_ENV = L.FS_ENV_ACCEL

    def __init__(self, env_params, sim_params, network=None, simulator='traci', scenario=None):
        for p in ADDITIONAL_ENV_PARAMS.keys():
            if p not in env_params.additional_params:
                raise KeyError('Environment parameter \'{}\' not supplied'.format(p))
        super().__init__(env_params, sim_params, network, simulator, scenario)

    @property
    def observation_space(self):
        return Box(low=-5, high=5, shape=(3,), dtype=np.float32)

    @property
    def action_space(self):
        return Box(low=-np.abs(self.env_params.additional_params['max_decel']),
                   high=self.env_params.additional_params['max_accel'], shape=(1,), dtype=np.float32)

    def get_state(self):
        obs = {}
        veh = self.k.vehicle
        for rl_id in veh.get_rl_ids():
            lead_id = veh.get_leader(rl_id) or rl_id
            max_speed = 15.
            max_length = self.env_params.additional_params['ring_length'][1]
            obs[rl_id] = np.array([veh.get_speed(rl_id) / max_speed,
                                   (veh.get_speed(lead_id) - veh.get_speed(rl_id)) / max_speed,
                                   veh.get_headway(rl_id) / max_length])
        return obs

    def _apply_rl_actions(self, rl_actions):
        if rl_actions:
            self.k.vehicle.apply_acceleration(list(rl_actions.keys()), list(rl_actions.values()))

    def compute_reward(self, rl_actions, **kwargs):
        if rl_actions is None:
            return 0
        vel = np.array([self.k.vehicle.get_speed(veh_id) for veh_id in self.k.vehicle.get_ids()])
        if any(vel < -100) or kwargs['fail']:
            return 0.
        eta_2 = 4.
        reward = eta_2 * np.mean(vel) / 20
        eta = 4
        mean_actions = np.mean(np.abs(list(rl_actions.values())))
        accel_threshold = 0
        if mean_actions > accel_threshold:
            reward += eta * (accel_threshold - mean_actions)
        return {key: reward for key in self.k.vehicle.get_rl_ids()}

    def additional_command(self):
        for rl_id in self.k.vehicle.get_rl_ids():
            lead_id = self.k.vehicle.get_leader(rl_id) or rl_id
            self.k.vehicle.set_observed(lead_id)

    def reset(self, new_inflow_rate=None):
        """wave_attenuation.py:262-312: a new ring length per rollout, vehicles re-placed with bunching 50."""
        if self.env_params.additional_params['ring_length'] is None:
            return super().reset()
        self.step_counter = 0
        length = random.randint(self.env_params.additional_params['ring_length'][0],
                                self.env_params.additional_params['ring_length'][1])
        X = redraw_ring(self, length)
        self.v_eq_max = fsolve(v_eq_max_function, np.array(4), args=(len(self.initial_ids), length))[0]
        self.sim.set_state(L.FS_FIELD_RING_LENGTH, np.full(1, float(length)))
        self.sim.set_state(L.FS_FIELD_INIT_POS, X)
        return super().reset()
