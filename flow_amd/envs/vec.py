"""VecFlowEnv: R replicas of one Flow environment stepped together on one GPU.

The reference parallelises rollouts as one OS process + one SUMO process per
environment (examples/train.py:149; SubprocVecEnv, examples/train.py:102-103).
Here the R replicas are rows of [R, N] arrays in HBM advanced by one launch;
observations, rewards and done flags stay on the device as torch tensors so a
learner on the same GPU consumes them without a host copy.
"""
import os
from copy import deepcopy

from flow_amd import _lib as L


class VecFlowEnv(object):
    """Parameters: either ``flow_params`` (the reference's dict: env_name, network, env, sim, net,
    veh, initial; flow/utils/registry.py:29-46) or the four objects ``env_class, env_params,
    sim_params, network``.  ``device`` is the HIP ordinal of this process' GPU; ``replica_offset`` the global index
    of this process' first replica when a job is sharded over GPUs (noise streams follow global replica ids).
    ``track_aux=True`` keeps ``k.vehicle.get_previous_speed`` / ``get_accel`` of the wrapped env current (fields the
    specialised rollout kernels do not write: such a handle steps on the generic kernels)."""

    def __init__(self, flow_params=None, num_replicas=4096, device=0, env_class=None, env_params=None,
                 sim_params=None, network=None, seed=None, replica_offset=0, track_aux=False):
        import torch
        self.torch = torch
        if flow_params is not None:
            from flow_amd.core.params import InitialConfig, TrafficLightParams
            env_class = flow_params["env_name"]
            env_params, sim_params = flow_params["env"], deepcopy(flow_params["sim"])
            network = flow_params["network"](
                name=flow_params["exp_tag"], vehicles=deepcopy(flow_params["veh"]), net_params=flow_params["net"],
                initial_config=flow_params.get("initial", InitialConfig()),
                traffic_lights=flow_params.get("tls", TrafficLightParams()))
        if seed is not None:
            sim_params = deepcopy(sim_params)
            sim_params.seed = seed
        self.num_envs = int(num_replicas)
        self.device = torch.device("cuda", int(device))
        env = env_class.__new__(env_class)
        env.num_replicas = self.num_envs
        env._device_index = int(device)
        env._replica_offset = int(replica_offset)     # global index of replica 0 (flow_amd.dist.shard_range)
        # per-vehicle previous speed / realised acceleration (k.vehicle.get_previous_speed / get_accel) are scalar-Env
        # accessors; keeping them costs the specialised rollout kernels (they do not write those fields)
        env._track_aux = bool(track_aux)
        env.__init__(env_params, sim_params, network)
        if env.FS_ENV is None or getattr(env, "HOST_HEADS", False):
            env.terminate()
            raise NotImplementedError("VecFlowEnv needs an env with an in-kernel observation/reward head")
        if getattr(env, "symmetric", False) or (env.FS_ENV in (L.FS_ENV_BOTTLENECK_DV, L.FS_ENV_BOTTLENECK)
                                                and env_params.evaluate):
            env.terminate()
            raise NotImplementedError("BottleneckDesiredVelocityEnv(symmetric=True) / evaluate=True gather their actions / "
                                      "reward on the host: scalar Env only")
        self.env = env
        self.sim = env.sim
        self.k = env.k
        self.observation_space = env.observation_space
        self.action_space = env.action_space
        self.obs_dim, self.num_rl, self.act_dim = self.sim.obs_dim, self.sim.num_rl, self.sim.act_dim
        R = self.num_envs
        self._obs = torch.empty((R, self.obs_dim), dtype=torch.float32, device=self.device)
        self._rew = torch.empty((R,), dtype=torch.float32, device=self.device)
        self._done = torch.zeros((R,), dtype=torch.uint8, device=self.device)
        import numpy as np
        self._resample = env.env_params.additional_params.get('ring_length', None) is not None \
            and env.FS_ENV in (L.FS_ENV_WAVE_ATTENUATION, L.FS_ENV_WAVE_ATTENUATION_PO, L.FS_ENV_WAVE_ATTENUATION_PO_MA)
        self._rng = np.random.default_rng(sim_params.seed)
        self._placement_cache = {}
        self.use_current_stream()

    def use_current_stream(self):
        """Enqueue the simulator's launches on torch's current stream (ordering with learner kernels).  Called by
        every reset / step / rollout, so a caller that switches streams (``with torch.cuda.stream(s):``) is followed:
        the launches, the temporaries they read and the caller's kernels are then all ordered on one stream."""
        st = self.torch.cuda.current_stream(self.device).cuda_stream
        if st != getattr(self, "_bound_stream", None):
            self.sim.set_stream(st)
            self._bound_stream = st

    def _check(self, t, shape, dtype):
        if t is None:
            return None
        if t.device != self.device or t.dtype != dtype or tuple(t.shape) != tuple(shape) or not t.is_contiguous():
            raise ValueError("expected a contiguous %s tensor of shape %s on %s" % (dtype, shape, self.device))
        return t

    def _resample_ring_lengths(self, mask_host):
        """WaveAttenuationEnv.reset (flow/envs/ring/wave_attenuation.py:157-210) for the replicas being reset:
        each draws its own ring length in ``ring_length`` and is re-placed with InitialConfig(bunching=50,
        min_gap=0).  Host-side (placement runs once per distinct length), then uploaded."""
        import numpy as np
        from flow_amd.core.kernel.network import NetworkKernel
        from flow_amd.core.params import InitialConfig, NetParams
        from flow_amd.envs.spec import check_placement, initial_positions
        lo, hi = self.env.env_params.additional_params['ring_length']
        idx = np.flatnonzero(mask_host)
        if idx.size == 0:
            return
        lengths = self.sim.get_state(L.FS_FIELD_RING_LENGTH).astype(np.float64)
        pending = self.sim.get_state(L.FS_FIELD_INIT_RING_LENGTH).astype(np.float64)
        init = self.sim.get_state(L.FS_FIELD_INIT_POS).astype(np.float64)
        self._draw_placements(idx, lengths, init)
        pending[idx] = lengths[idx]
        self.sim.set_state(L.FS_FIELD_RING_LENGTH, lengths)         # (sets the pending lengths of ALL replicas too ...)
        self.sim.set_state(L.FS_FIELD_INIT_RING_LENGTH, pending)    # ... so those of the replicas not reset go back
        self.sim.set_state(L.FS_FIELD_INIT_POS, init)

    def _draw_placements(self, idx, lengths, init):
        """random.randint(lo, hi) per replica in ``idx``; InitialConfig(bunching=50, min_gap=0) placement for the drawn
        length (once per distinct length), written into ``lengths`` / ``init``."""
        import numpy as np
        from flow_amd.core.kernel.network import NetworkKernel
        from flow_amd.core.params import InitialConfig, NetParams
        from flow_amd.envs.spec import check_placement, initial_positions
        lo, hi = self.env.env_params.additional_params['ring_length']
        draw = self._rng.integers(lo, hi + 1, idx.size)
        veh_len = np.array([v["length"] for v in self.env._spec["vehicles"]])
        ic = InitialConfig(bunching=50, min_gap=0)
        for length in np.unique(draw):
            if length not in self._placement_cache:
                add = dict(self.env.net_params.additional_params)
                add["length"] = int(length)
                net = self.env.network.__class__(self.env.network.orig_name, self.env.network.vehicles,
                                                 NetParams(additional_params=add), ic)
                nk = NetworkKernel(net, junction_length=self.k.network.junction_length)
                X, lanes = initial_positions(nk, ic, self.sim.N, 1)
                check_placement(X, veh_len, nk.length(), lanes)
                self._placement_cache[length] = X[0]
            sel = idx[draw == length]
            lengths[sel] = float(length)
            init[sel] = self._placement_cache[length][None, :]

    def redraw_ring_lengths(self):
        """Give EVERY replica a fresh pending ring length and start placement for its NEXT reset without touching the
        episode it is in (FS_FIELD_INIT_RING_LENGTH / FS_FIELD_INIT_POS): call between replays of a fragment captured
        with ``reset_done=True``, so that the resets inside the graph redraw the length per episode as
        WaveAttenuationEnv.reset does (flow/envs/ring/wave_attenuation.py:157-210).  Host-side draw + two uploads; the
        uploads synchronise the handle's stream."""
        import numpy as np
        if not self._resample:
            return
        lengths = self.sim.get_state(L.FS_FIELD_INIT_RING_LENGTH).astype(np.float64)
        init = self.sim.get_state(L.FS_FIELD_INIT_POS).astype(np.float64)
        self._draw_placements(np.arange(self.num_envs), lengths, init)
        self.sim.set_state(L.FS_FIELD_INIT_RING_LENGTH, lengths)
        self.sim.set_state(L.FS_FIELD_INIT_POS, init)

    def reset(self, mask=None):
        """Reset all replicas (or those where ``mask`` [R] uint8/bool tensor is set); returns obs [R, obs_dim]."""
        self.use_current_stream()
        if mask is not None:
            mask = self._check(mask.to(self.torch.uint8), (self.num_envs,), self.torch.uint8)
        if self._resample:
            import numpy as np
            self._resample_ring_lengths(np.ones(self.num_envs, bool) if mask is None else mask.cpu().numpy() != 0)
        self.sim.reset_dev(self._obs, mask)
        return self._obs

    def reset_done(self):
        """Reset exactly the replicas whose last ``done`` flag is set.  No host synchronisation, unless the
        environment redraws its network per episode (WaveAttenuationEnv with ``ring_length``)."""
        self.use_current_stream()
        if self._resample:
            self._resample_ring_lengths(self._done.cpu().numpy() != 0)
        self.sim.reset_dev(self._obs, self._done)
        return self._obs

    # ---- performance cliffs: say so, once -----------------------------------------------------------------------
    GENERIC_KERNELS = ("k_steps", "k_steps<CSET>", "k_steps_ml")

    def why_generic(self):
        """What sends this handle's rollouts to the generic step kernel (4-7x slower than the specialised rollout
        kernels of closed loops: flow_amd/csrc/flowsim_launch.h), as a list of the configuration fields responsible;
        empty for open networks (they have kernels of their own)."""
        sp = self.env._spec
        if sp.get("network") in ("merge", "bottleneck"):
            return []
        veh, why = sp["vehicles"], []
        CTRL_SIM, CTRL_RL, CTRL_IDM = 0, 1, 2                # (include/flowsim.h FS_CTRL_*)
        if any(v["controller"] not in (CTRL_IDM, CTRL_RL, CTRL_SIM) for v in veh):
            why.append("an acceleration controller other than IDMController / RLController / SimCarFollowingController")
        if any(v.get("fail_safe", 0) for v in veh):
            why.append("fail_safe")
        if len(veh) % 2 and not sp.get("segments"):
            why.append("an odd number of vehicles (the ring kernels hold two per lane)")
        if sp.get("sort_vehicles"):
            why.append("sort_vehicles=True")
        if sp.get("evaluate"):
            why.append("EnvParams(evaluate=True)")
        if sp.get("track_aux"):
            why.append("track_aux=True")
        if int(sp.get("num_lanes", 1)) > 1:
            why.append("a multi-lane ring (k_steps_ml)")
        if int(sp.get("sims_per_step", 1)) != 1:
            why.append("sims_per_step > 1")
        if sp.get("integrator", "euler") != "euler":
            why.append("use_ballistic=True")
        if sp.get("obs_perm") is not None:
            why.append("InitialConfig(shuffle=True) / a placement that is not in driving order")
        heads = [L.FS_ENV_ACCEL, L.FS_ENV_WAVE_ATTENUATION_PO, L.FS_ENV_ACCEL_PO_MA]
        if not sp.get("segments"):
            heads.append(L.FS_ENV_WAVE_ATTENUATION_PO_MA)
        if self.env.FS_ENV not in heads:
            why.append("the environment head of %s" % type(self.env).__name__)
        if os.environ.get("FLOWSIM_FORCE_GENERIC") == "1":
            why.append("FLOWSIM_FORCE_GENERIC=1")
        return why

    def _warn_if_generic(self):
        if getattr(self, "_warned_generic", False) or self.num_envs < 1024:
            return
        if self.sim.last_kernel in self.GENERIC_KERNELS:
            import warnings
            self._warned_generic = True
            why = self.why_generic()
            warnings.warn("VecFlowEnv: %d replicas step on the generic kernel %s (several times slower than the rollout "
                          "kernels of this network) because of: %s" %
                          (self.num_envs, self.sim.last_kernel, "; ".join(why) if why else "this configuration"),
                          stacklevel=3)

    def step(self, actions=None):
        """One Env.step of every replica.  ``actions``: float32 [R, action_dim] device tensor or None."""
        self.use_current_stream()
        a = self._check(actions, (self.num_envs, self.act_dim), self.torch.float32) if self.act_dim else None
        self.sim.step_dev(self._obs, self._rew, self._done, a)
        self._warn_if_generic()
        return self._obs, self._rew, self._done

    def rollout(self, num_steps, actions=None, obs_every_step=True, out=None):
        """``num_steps`` consecutive steps in one launch.  ``actions``: None, [R, num_rl] (held) or
        [K, R, num_rl] (one per step).  Returns (obs, rew, done) with a leading K axis when
        ``obs_every_step``."""
        torch, R, K = self.torch, self.num_envs, int(num_steps)
        self.use_current_stream()
        if out is None:
            lead = (K,) if obs_every_step else ()
            out = (torch.empty(lead + (R, self.obs_dim), dtype=torch.float32, device=self.device),
                   torch.empty(lead + (R,), dtype=torch.float32, device=self.device),
                   torch.empty(lead + (R,), dtype=torch.uint8, device=self.device))
        stride = 0
        if actions is not None and self.act_dim:
            if actions.dim() == 3:
                self._check(actions, (K, R, self.act_dim), torch.float32)
                stride = R * self.act_dim
            else:
                self._check(actions, (R, self.act_dim), torch.float32)
        else:
            actions = None
        self.sim.rollout_dev(K, out[0], out[1], out[2], actions, stride, obs_every_step)
        self._warn_if_generic()
        return out

    def capture(self, num_steps, policy=None, reset_done=False):
        """A closed-loop fragment of ``num_steps`` steps as ONE replayable HIP graph (StepGraph): per step the
        ``policy`` (a callable on the [R, obs_dim] observation tensor returning [R, action_dim] actions, e.g. a torch
        module) and one fs_step_dev launch, captured back to back on one stream.  What the reference's rollout workers
        do with one Python call + socket round trips per step (examples/train.py:110-212) costs one graph launch per
        fragment here; observations and actions never leave HBM.

        ``done`` is the simulator's flag byte (bit 0: horizon reached, bit 1: collision), not a 0/1 value."""
        if reset_done and self._resample:
            import warnings
            warnings.warn("VecFlowEnv.capture(reset_done=True): this environment redraws its ring length on reset "
                          "(WaveAttenuationEnv ring_length, flow/envs/ring/wave_attenuation.py:157-210); a reset inside "
                          "the captured graph re-places a replica on the length it drew at the last vec.reset() / "
                          "vec.reset_done() instead.  Call vec.redraw_ring_lengths() between replays to give every "
                          "replica a fresh pending length for its next in-graph reset.", stacklevel=2)
        return StepGraph(self, num_steps, policy, reset_done)

    def policy_rollout(self, policy, num_steps, reset_done=False, out=None):
        """``num_steps`` x (policy -> action -> Env.step [-> Env.reset of finished episodes]) in ONE kernel launch
        (``fs_policy_rollout_dev``; ``policy``: a ``flow_amd.utils.device_policy.DevicePolicy``).  Returns device tensors
        ``obs [K+1, R, obs_dim]`` (obs[0]: the state the fragment starts from), ``actions [K, R]``, ``logp [K, R]``,
        ``rew [K, R]``, ``done [K, R]`` (flag byte: bit 0 horizon, bit 1 collision).  Built for the reference's RL ring
        experiments (one RL vehicle, WaveAttenuationPOEnv); ``capture`` serves every other environment / model."""
        torch, R, K = self.torch, self.num_envs, int(num_steps)
        self.use_current_stream()
        if reset_done and self._resample and not getattr(self, "_warned_pending_length", False):
            import warnings
            self._warned_pending_length = True
            warnings.warn("VecFlowEnv.policy_rollout(reset_done=True): this environment redraws its ring length on reset "
                          "(flow/envs/ring/wave_attenuation.py:157-210); every reset inside ONE fragment takes the replica's "
                          "pending length (FS_FIELD_INIT_RING_LENGTH) -- call vec.redraw_ring_lengths() between fragments, "
                          "and keep fragments shorter than an episode if each episode must draw its own.", stacklevel=2)
        if out is None:
            out = (torch.empty((K + 1, R, self.obs_dim), dtype=torch.float32, device=self.device),
                   torch.empty((K, R), dtype=torch.float32, device=self.device),
                   torch.empty((K, R), dtype=torch.float32, device=self.device),
                   torch.empty((K, R), dtype=torch.float32, device=self.device),
                   torch.empty((K, R), dtype=torch.uint8, device=self.device))
        self.sim.policy_rollout_dev(policy.struct, K, out[0], out[1], out[2], out[3], out[4], reset_done=reset_done)
        return out

    # ---- host-side inspection
    def get_state(self, field):
        return self.sim.get_state(field)

    @property
    def positions(self):
        return self.sim.get_state(L.FS_FIELD_POS)

    @property
    def speeds(self):
        return self.sim.get_state(L.FS_FIELD_VEL)

    def vehicle_view(self, replica):
        """The k.vehicle accessor object looking at ``replica``."""
        self.k.vehicle.attach(self.sim, int(replica))
        return self.k.vehicle

    def close(self):
        self.env.terminate()

    terminate = close


class StepGraph(object):
    """``vec.capture(K, policy)``: K x (policy -> actions -> Env.step of every replica) recorded once with HIP stream
    capture (``torch.cuda.CUDAGraph``; libflowsim's launches go to the capturing stream) and replayed per fragment.

    Buffers (device, owned by the graph): ``obs [K+1, R, obs_dim]`` (``obs[0]`` = observation before the first step:
    what ``begin`` set, then the last observation of the previous replay), ``actions [K, R, action_dim]``, ``rew [K, R]``,
    ``done [K, R]`` uint8.  With ``reset_done`` every replica whose episode ended is reset inside the graph
    (masked fs_reset_dev) and ``obs[k+1]`` holds the first observation of its next episode, as a vectorised
    Gym / RLlib VectorEnv does.  The policy must be capturable (no host synchronisation, static shapes).
    Building the graph runs up to two eager warm-up steps: call ``vec.reset()`` and ``begin(obs)`` afterwards."""

    def __init__(self, vec, num_steps, policy=None, reset_done=False):
        torch = vec.torch
        self.vec, self.K, self.policy, self.reset_done = vec, int(num_steps), policy, bool(reset_done)
        R, K = vec.num_envs, self.K
        dev = vec.device
        self.obs = torch.zeros((K + 1, R, vec.obs_dim), dtype=torch.float32, device=dev)
        self.rew = torch.zeros((K, R), dtype=torch.float32, device=dev)
        self.done = torch.zeros((K, R), dtype=torch.uint8, device=dev)
        self.actions = torch.zeros((K, R, max(vec.act_dim, 1)), dtype=torch.float32, device=dev)
        self._carry = torch.zeros((R, vec.obs_dim), dtype=torch.float32, device=dev)
        self.stream = torch.cuda.Stream(dev)
        with torch.cuda.stream(self.stream):
            vec.use_current_stream()                 # bind BEFORE the capture: fs_set_stream synchronises
            self.obs[0].copy_(vec._obs)
            self._body(2 if K > 1 else 1)            # eager warm-up: lazy host work (divisor proofs, allocator) happens here
            vec.sim.sync()
            self.obs[0].copy_(vec._obs)
        self.stream.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream):
            self.obs[0].copy_(self._carry)
            self._body(K)
            self._carry.copy_(self.obs[K])
        # NOTE: the eager warm-up advanced the simulator by up to two steps: vec.reset() + begin(obs) start a rollout

    def _body(self, steps):
        vec, sim = self.vec, self.vec.sim
        for k in range(steps):
            a = None
            if vec.act_dim:
                if self.policy is not None:
                    self.actions[k].copy_(self.policy(self.obs[k]))
                a = self.actions[k]
            sim.step_dev(self.obs[k + 1], self.rew[k], self.done[k], a)
            if self.reset_done:
                sim.reset_dev(self.obs[k + 1], self.done[k])

    def _after_caller(self):
        # the graph's stream is non-blocking: what the caller enqueued on ITS stream so far (vec.reset()'s kernels, the
        # optimiser step that rewrote the policy weights) must have finished before anything here reads it
        self.stream.wait_stream(self.vec.torch.cuda.current_stream(self.vec.device))

    def begin(self, obs0):
        """Set the observation the first step's policy call sees (after ``vec.reset()``)."""
        self._after_caller()
        with self.vec.torch.cuda.stream(self.stream):
            self._carry.copy_(obs0)

    def replay(self):
        """One fragment: returns (obs [K+1,R,D], actions [K,R,A], rew [K,R], done [K,R]) views, valid until the next
        replay.  Ordered on both sides: the replay starts after the work the caller's current stream holds so far, and
        the caller's current stream waits for the fragment before it runs anything enqueued after this call (no host
        synchronisation either way).  ``done`` is a flag byte (bit 0 horizon, bit 1 collision)."""
        self._after_caller()
        with self.vec.torch.cuda.stream(self.stream):      # CUDAGraph.replay() goes to torch's CURRENT stream
            self.graph.replay()
        self.vec.torch.cuda.current_stream(self.vec.device).wait_stream(self.stream)
        return self.obs, self.actions, self.rew, self.done

    def synchronize(self):
        self.stream.synchronize()
