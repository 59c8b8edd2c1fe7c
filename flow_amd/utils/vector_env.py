"""FlowVectorEnv: the batched simulator behind RLlib's ``VectorEnv`` interface (ray/rllib/env/vector_env.py:
``vector_reset``, ``reset_at``, ``vector_step``, ``get_sub_environments``) -- what ``examples/train.py:110-212`` of the
reference obtains by giving every rollout worker its own SUMO process becomes ONE handle with ``num_envs`` replicas.

ray is not a dependency: when it is importable the class derives from ``ray.rllib.env.VectorEnv`` so that
``register_env(name, lambda cfg: FlowVectorEnv(flow_params, cfg["num_envs"]))`` is accepted as it is; without ray
the same methods exist on a plain object (used by the tests and by ``examples/train_vec.py``).

Host copies happen only at this boundary (RLlib's sampler wants numpy): one [R, obs] device-to-host copy per
``vector_step``.  A learner that lives on the GPU should use ``VecFlowEnv`` / ``VecFlowEnv.capture`` directly.
"""
import numpy as np

try:                                                             # pragma: no cover - ray absent in this image
    from ray.rllib.env.vector_env import VectorEnv as _Base
except ImportError:
    _Base = object


class FlowVectorEnv(_Base):
    def __init__(self, flow_params, num_envs, device=0, seed=None):
        import torch
        from flow_amd.envs.vec import VecFlowEnv
        self.torch = torch
        self.vec = VecFlowEnv(flow_params, num_replicas=int(num_envs), device=device, seed=seed)
        self.observation_space = self.vec.observation_space
        self.action_space = self.vec.action_space
        self.num_envs = int(num_envs)
        if _Base is not object:                                  # pragma: no cover
            _Base.__init__(self, self.observation_space, self.action_space, self.num_envs)
        self._mask = torch.zeros(self.num_envs, dtype=torch.uint8, device=self.vec.device)

    # ---- VectorEnv interface
    def vector_reset(self):
        """List of ``num_envs`` initial observations."""
        return list(self.vec.reset().cpu().numpy())

    def reset_at(self, index=None):
        """Reset ONE sub-environment (RLlib calls it when that episode ended); returns its observation."""
        index = 0 if index is None else int(index)
        self._mask.zero_()
        self._mask[index] = 1
        return self.vec.reset(self._mask)[index].cpu().numpy()

    def vector_step(self, actions):
        """``actions``: list of ``num_envs`` actions -> (obs list, reward list, done list, info list)."""
        a = None
        if self.vec.act_dim:
            a = self.torch.as_tensor(np.asarray(actions, dtype=np.float32).reshape(self.num_envs, self.vec.act_dim),
                                     device=self.vec.device)
        obs, rew, done = self.vec.step(a)
        return (list(obs.cpu().numpy()), list(rew.cpu().numpy().astype(float)),
                list(done.cpu().numpy().astype(bool)), [{} for _ in range(self.num_envs)])

    def get_sub_environments(self):
        """The replicas are rows of one simulator, not objects: the scalar environment whose accessors look at one
        replica at a time (``vec.vehicle_view(i)``) stands for all of them."""
        return [self.vec.env]

    get_unwrapped = get_sub_environments

    def close(self):
        self.vec.close()
