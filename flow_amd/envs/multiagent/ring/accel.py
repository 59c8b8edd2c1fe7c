"""Multi-agent acceleration environments (flow/envs/multiagent/ring/accel.py).

Dict-in / dict-out wrappers of the reference's per-agent observation and reward code over the GPU step loop: the
physics of a step is one HIP launch (``FS_ENV_ACCEL`` head); the per-agent observations are assembled on the host from
``fs_get_state`` through the same ``k.vehicle`` accessors the reference calls.  One replica per environment; the
batched ``VecFlowEnv`` needs an in-kernel head and does not take these classes."""
import numpy as np

from flow_amd import _lib as L
from flow_amd.core import rewards
from flow_amd.envs.multiagent.base import MultiEnv
from flow_amd.envs.ring.accel import AccelEnv
from flow_amd.utils.spaces import Box

ADDITIONAL_ENV_PARAMS = {
    # maximum acceleration for autonomous vehicles, in m/s^2
    "max_accel": 1,
    # maximum deceleration for autonomous vehicles, in m/s^2
    "max_decel": 1,
    # desired velocity for all vehicles in the network, in m/s
    "target_velocity": 20,
}


class _PendingActions(object):
    """MultiEnv over the closed-loop kernels: RL commands travel through k.vehicle.apply_acceleration."""

    HOST_HEADS = True          # observation / reward are computed on the host (no in-kernel head)

    def _action_vector(self):
        pend = self.k.vehicle._pending
        if not pend:
            return None
        return np.array([[pend.get(v, 0.0) for v in self._rl_action_order()]], dtype=np.float32)


class AdversarialAccelEnv(_PendingActions, AccelEnv, MultiEnv):
    """flow/envs/multiagent/ring/accel.py:20-97: an 'av' agent commands the RL vehicles, an 'adversary' perturbs the
    commands (``perturb_weight``); both see [v / v_max, x / L] of every vehicle; rewards are +/- the AccelEnv reward."""

    def _apply_rl_actions(self, rl_actions):
        sorted_rl_ids = [veh_id for veh_id in self.sorted_ids if veh_id in self.k.vehicle.get_rl_ids()]
        av_action = rl_actions['av']
        adv_action = rl_actions['adversary']
        perturb_weight = self.env_params.additional_params['perturb_weight']
        rl_action = av_action + perturb_weight * adv_action
        self.k.vehicle.apply_acceleration(sorted_rl_ids, rl_action)

    def compute_reward(self, rl_actions, **kwargs):
        if self.env_params.evaluate:
            reward = np.mean(self.k.vehicle.get_speed(self.k.vehicle.get_ids()))
        else:
            reward = rewards.desired_velocity(self, fail=kwargs['fail'])
        return {'av': reward, 'adversary': -reward}

    def get_state(self, **kwargs):
        state = np.array([[self.k.vehicle.get_speed(veh_id) / self.k.network.max_speed(),
                           self.k.vehicle.get_x_by_id(veh_id) / self.k.network.length()]
                          for veh_id in self.sorted_ids])
        state = np.ndarray.flatten(state)
        return {'av': state, 'adversary': state}

    def reset(self, new_inflow_rate=None):
        MultiEnv.reset(self)
        for veh_id in self.k.vehicle.get_ids():
            self.absolute_position[veh_id] = self.k.vehicle.get_x_by_id(veh_id)
            self.prev_pos[veh_id] = self.k.vehicle.get_x_by_id(veh_id)
        return self.get_state()


class MultiAgentAccelPOEnv(_PendingActions, MultiEnv):
    """flow/envs/multiagent/ring/accel.py:100-229: every RL vehicle is an agent observing [x / L, v / v_max,
    (v_lead - v) / v_max, gap_lead / L, (v - v_follow) / v_max, gap_follow / L]; shared desired-velocity reward."""

    FS_ENV = L.FS_ENV_ACCEL

    def __init__(self, env_params, sim_params, network=None, simulator='traci', scenario=None):
        for p in ADDITIONAL_ENV_PARAMS.keys():
            if p not in env_params.additional_params:
                raise KeyError('Environment parameter "{}" not supplied'.format(p))
        self.leader = []
        self.follower = []
        super().__init__(env_params, sim_params, network, simulator, scenario)

    @property
    def action_space(self):
        return Box(low=-abs(self.env_params.additional_params["max_decel"]),
                   high=self.env_params.additional_params["max_accel"], shape=(1,), dtype=np.float32)

    @property
    def observation_space(self):
        return Box(low=-5, high=5, shape=(6,), dtype=np.float32)

    def _apply_rl_actions(self, rl_actions):
        for veh_id in self.k.vehicle.get_rl_ids():
            self.k.vehicle.apply_acceleration(veh_id, rl_actions[veh_id])

    def compute_reward(self, rl_actions, **kwargs):
        reward = rewards.desired_velocity(self, fail=kwargs['fail'])
        return {key: reward for key in self.k.vehicle.get_rl_ids()}

    def get_state(self, **kwargs):
        self.leader = []
        self.follower = []
        obs = {}
        max_speed = self.k.network.max_speed()
        max_length = self.k.network.length()
        veh = self.k.vehicle
        for rl_id in veh.get_rl_ids():
            this_pos = veh.get_x_by_id(rl_id)
            this_speed = veh.get_speed(rl_id)
            lead_id = veh.get_leader(rl_id)
            follower = veh.get_follower(rl_id)
            if lead_id in ["", None]:
                lead_speed = max_speed
                lead_head = max_length
            else:
                self.leader.append(lead_id)
                lead_speed = veh.get_speed(lead_id)
                lead_head = veh.get_x_by_id(lead_id) - veh.get_x_by_id(rl_id) - veh.get_length(rl_id)
            if follower in ["", None]:
                follow_speed = 0
                follow_head = max_length
            else:
                self.follower.append(follower)
                follow_speed = veh.get_speed(follower)
                follow_head = veh.get_headway(follower)
            obs[rl_id] = np.array([this_pos / max_length, this_speed / max_speed,
                                   (lead_speed - this_speed) / max_speed, lead_head / max_length,
                                   (this_speed - follow_speed) / max_speed, follow_head / max_length])
        return obs

    def additional_command(self):
        for veh_id in self.leader + self.follower:
            self.k.vehicle.set_observed(veh_id)

    def reset(self, new_inflow_rate=None):
        self.leader = []
        self.follower = []
        return super().reset()
