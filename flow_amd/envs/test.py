"""TestEnv (flow/envs/test.py:8-54): empty observation, reward 0 or a user callable."""
import numpy as np

from flow_amd.envs.base import Env
from flow_amd.utils.spaces import Box


class TestEnv(Env):
    @property
    def action_space(self):
        return Box(low=0, high=0, shape=(0,), dtype=np.float32)

    @property
    def observation_space(self):
        return Box(low=0, high=0, shape=(0,), dtype=np.float32)

    def _apply_rl_actions(self, rl_actions):
        return

    def compute_reward(self, rl_actions, **kwargs):
        if "reward_fn" in self.env_params.additional_params:
            return self.env_params.additional_params["reward_fn"](self)
        return 0

    def get_state(self, **kwargs):
        return np.array([])
