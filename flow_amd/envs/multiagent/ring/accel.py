"""AdversarialAccelEnv / MultiAgentAccelPOEnv (flow/envs/multiagent/ring/accel.py) over the kernel's heads.

* ``MultiAgentAccelPOEnv``: every RL vehicle is an agent with a 6-value observation; the step kernel writes one block
  per RL vehicle (head ``FS_ENV_ACCEL_PO_MA``, column = the vehicle's RL slot) and the shared ``desired_velocity``
  reward; this class only slices the row into the dicts RLlib's ``MultiAgentEnv`` speaks.
* ``AdversarialAccelEnv``: ``AccelEnv``'s observation and reward for two agents ('av', 'adversary'); the adversary's
  action perturbs the AV's (``av + perturb_weight * adversary``) before it reaches the kernel's action row."""
import numpy as np

from flow_amd import _lib as L
from flow_amd.envs.multiagent.base import MultiEnv
from flow_amd.envs.ring.accel import AccelEnv
from flow_amd.utils.spaces import Box

ADDITIONAL_ENV_PARAMS = {
    # maximum acceleration of autonomous vehicles
    "max_accel": 1,
    # maximum deceleration of autonomous vehicles
    "max_decel": 1,
    # desired velocity for all vehicles in the network, in m/s
    "target_velocity": 20,
}


class _RLColumns(object):
    """{rl_id: column of the kernel's action row / observation blocks} = the RL slot the vehicle sits in."""

    def _rl_columns(self):
        spec, veh = self._spec, self.k.vehicle
        return {v: spec["vehicles"][veh._slot[v]]["rl_index"] for v in veh.get_rl_ids()}

    def _row_from_dict(self, rl_actions):
        cols = self._rl_columns()
        row = np.zeros((1, max(self.sim.act_dim, 1)), dtype=np.float32)
        for rl_id, col in cols.items():
            if rl_id in rl_actions:
                row[0, col] = float(np.asarray(rl_actions[rl_id]).reshape(-1)[0])
        return row


class AdversarialAccelEnv(AccelEnv, MultiEnv):
    """multiagent/ring/accel.py:18-81.  ``perturb_weight`` comes from ``env_params.additional_params``."""

    def _apply_rl_actions(self, rl_actions):
        weight = self.env_params.additional_params['perturb_weight']
        combined = np.asarray(rl_actions['av'], dtype=np.float64) + weight * np.asarray(rl_actions['adversary'],
                                                                                         dtype=np.float64)
        AccelEnv._apply_rl_actions(self, combined)

    def compute_reward(self, rl_actions, **kwargs):
        reward = float(self._last_reward)            # desired_velocity, or the mean speed with ``evaluate`` (the kernel's)
        return {'av': reward, 'adversary': -reward}

    def get_state(self, **kwargs):
        state = np.array(self._last_obs, dtype=np.float64)          # AccelEnv's row
        # (the reference interleaves [v_i, x_i] per vehicle, :76-81; AccelEnv's kernel row is [v..., x...])
        n = state.shape[0] // 2
        state = np.stack([state[:n], state[n:]], axis=1).reshape(-1)
        return {'av': state, 'adversary': state}


class MultiAgentAccelPOEnv(_RLColumns, MultiEnv):
    """multiagent/ring/accel.py:84-227."""

    FS_ENV = L.FS_ENV_ACCEL_PO_MA

    def __init__(self, env_params, sim_params, network, simulator='traci'):
        for p in ADDITIONAL_ENV_PARAMS.keys():
            if p not in env_params.additional_params:
                raise KeyError('Environment parameter "{}" not supplied'.format(p))
        self.leader, self.follower = [], []
        self._ma_actions = None
        super().__init__(env_params, sim_params, network, simulator)

    @property
    def action_space(self):
        return Box(low=-abs(self.env_params.additional_params["max_decel"]),
                   high=self.env_params.additional_params["max_accel"], shape=(1, ), dtype=np.float32)

    @property
    def observation_space(self):
        return Box(low=-5, high=5, shape=(6, ), dtype=np.float32)

    def _apply_rl_actions(self, rl_actions):
        self._ma_actions = self._row_from_dict(rl_actions)

    def _action_vector(self):
        a, self._ma_actions = self._ma_actions, None
        return a

    def compute_reward(self, rl_actions, **kwargs):
        return {key: float(self._last_reward) for key in self.k.vehicle.get_rl_ids()}

    def get_state(self, **kwargs):
        veh = self.k.vehicle
        row = np.asarray(self._last_obs, dtype=np.float64)
        self.leader = [l for l in (veh.get_leader(r) for r in veh.get_rl_ids()) if l not in ("", None)]
        self.follower = [f for f in (veh.get_follower(r) for r in veh.get_rl_ids()) if f not in ("", None)]
        return {rl_id: row[6 * col:6 * col + 6].copy() for rl_id, col in self._rl_columns().items()}

    def additional_command(self):
        for veh_id in self.leader + self.follower:
            self.k.vehicle.set_observed(veh_id)

    def reset(self, new_inflow_rate=None):
        self.leader, self.follower = [], []
        return super().reset()
