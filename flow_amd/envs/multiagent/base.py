"""MultiEnv: dict-in / dict-out version of Env (flow/envs/multiagent/base.py:20-415).

``step`` is the reference's ``_step_helper`` (:116-226).  The fork wraps it in action-repeat helpers that
need ``self._action_repeat`` / ``self._agents`` (its traffic-light environments define them); environments
without those attributes step once per call here."""
import random

import numpy as np

from flow_amd.envs.base import Env

try:                                             # pragma: no cover - ray is optional
    from ray.rllib.env import MultiAgentEnv as _MA
except Exception:
    class _MA(object):
        pass


class MultiEnv(_MA, Env):
    """See module docstring."""

    UNCLIPPED_ACTIONS = True      # multiagent/base.py:366-391: this fork's clip_actions returns the dict as it is

    def step(self, rl_actions):
        if getattr(self, "_action_repeat", False):
            raise NotImplementedError("action repeat is defined by the fork's traffic-light environments only")
        return self._step_helper(rl_actions)

    def _step_helper(self, rl_actions):
        n_sub = self.env_params.sims_per_step
        self.time_counter += n_sub
        self.step_counter += n_sub
        self.apply_rl_actions(rl_actions)
        self.additional_command()
        obs, rew, done = self.sim.step(self._action_vector())
        self.k.update(reset=False)
        self.time_counter = int(self.sim.time_counter[0])
        self._last_obs, self._last_reward = obs[0], float(rew[0])
        self.k.simulation.crashed = False
        crash = 0                                                    # multiagent/base.py:188-190
        states = self.get_state()
        arrived = self.k.vehicle.get_arrived_ids() or []
        done = {key: key in arrived for key in states.keys()}
        limit = n_sub * (self.env_params.warmup_steps + self.env_params.horizon)
        done['__all__'] = bool(crash or self.time_counter >= limit)
        infos = {key: {} for key in states.keys()}
        if self.env_params.clip_actions:
            reward = self.compute_reward(self.clip_actions(rl_actions), fail=crash)
        else:
            reward = self.compute_reward(rl_actions, fail=crash)
        for rl_id in (self.k.vehicle.get_arrived_rl_ids() or []):    # :221-224
            done[rl_id] = True
            reward[rl_id] = 0
            states[rl_id] = None
        return states, reward, done, infos

    def reset(self, new_inflow_rate=None, perform_extra_work=None):
        self.time_counter = 0
        if self.sim_params.restart_instance or self.step_counter > 2e6:
            self.step_counter = 0
            self.sim_params.seed = random.randint(0, int(1e5))
        obs = self.sim.reset()
        self.k.update(reset=True)
        self.time_counter = int(self.sim.time_counter[0])
        self._last_obs, self._last_reward = obs[0], 0.0
        return self.get_state()

    def clip_actions(self, rl_actions=None):
        """multiagent/base.py:366-391: the fork returns the dict unclipped."""
        if rl_actions is None:
            return None
        return rl_actions

    def apply_rl_actions(self, rl_actions=None):
        if rl_actions is None:
            return
        self._apply_rl_actions(self.clip_actions(rl_actions))

