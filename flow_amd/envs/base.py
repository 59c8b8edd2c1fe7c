"""Env: the Gym-style environment contract of flow/envs/base.py:28-799 over the GPU simulator.

Same constructor, attributes, ``step`` / ``reset`` / ``terminate`` / ``clip_actions`` /
``apply_rl_actions`` contract and the same abstract hooks (``action_space``,
``observation_space``, ``_apply_rl_actions``, ``get_state``, ``compute_reward``,
``additional_command``).  What used to be N controller calls + N TraCI round trips +
a SUMO step + a Python state rebuild per sub-step (envs/base.py:324-382) is one fused
HIP launch per ``step`` through libflowsim.

One ``Env`` is one replica (R = 1), as in the reference; the throughput path is
``flow_amd.envs.vec.VecFlowEnv`` (R replicas, device tensors in and out).
"""
import atexit
import random
from copy import deepcopy

import numpy as np

from flow_amd import _lib as L
from flow_amd.core.kernel import Kernel
from flow_amd.envs.spec import build_spec, initial_positions, check_placement
from flow_amd.sim import FlowSim
from flow_amd.utils.exceptions import FatalFlowError
from flow_amd.utils.spaces import Box, Tuple

try:                                             # pragma: no cover
    import gym
    _Base = gym.Env
except Exception:
    _Base = object


class Env(_Base):
    """See module docstring.  ``simulator`` accepts 'traci' (the reference default, so that
    existing flow_params run unchanged) and 'hip'; both select the GPU step loop."""

    FS_ENV = None            # built-in observation/reward head (enum fs_env); None = Python hooks
    num_replicas = 1

    def __init__(self, env_params, sim_params, network=None, simulator='traci', scenario=None):
        self.env_params = env_params
        self.network = scenario if scenario is not None else network
        self.net_params = self.network.net_params
        self.initial_config = self.network.initial_config
        self.sim_params = deepcopy(sim_params)
        self.should_render = self.sim_params.render
        self.sim_params.render = False
        if self.should_render not in [True, False, 'gray', 'dgray', 'rgb', 'drgb']:
            raise FatalFlowError('Mode %s is not supported!' % self.should_render)   # envs/base.py:227
        self.time_counter = 0
        self.step_counter = 0
        self.initial_state = {}
        self.state = None
        self.obs_var_labels = []
        self.sim_step = sim_params.sim_step
        self.simulator = simulator
        self.k = Kernel(simulator=self.simulator, sim_params=self.sim_params)
        self.k.generate_network(self.network)
        self.k.vehicle.initialize(deepcopy(self.network.vehicles))
        self.available_routes = self.k.network.rts
        self.initial_ids = deepcopy(self.network.vehicles.ids)
        self.initial_vehicles = self.k.vehicle
        self.sim = None
        self._last_obs = None
        self._last_reward = 0.0
        self._crash = False
        self._recorder = None
        self._sim_time = 0.0
        self.setup_initial_state()
        if self.sim_params.emission_path is not None:
            from flow_amd.core.util import TrajectoryRecorder, ensure_dir
            ensure_dir(self.sim_params.emission_path)
            self._recorder = TrajectoryRecorder(self)
        atexit.register(self.terminate)

    # ------------------------------------------------------------------ construction
    def _rl_action_order(self):
        """Vehicle ids in the order of the action vector (envs/ring/accel.py:103-107 uses id
        order; WaveAttenuationEnv uses the sorted rl ids, wave_attenuation.py:108-111)."""
        return [v for v in self.k.vehicle.get_ids() if v in self.k.vehicle.get_rl_ids()]

    def _po_max_length(self):
        return self.k.network.length()

    def _precision(self):
        return getattr(self.sim_params, "precision", "f32")

    def setup_initial_state(self):
        """envs/base.py:268-292: compute the start placement and (re)create the simulator."""
        if self.sim is not None:
            self.sim.close()
        if self.initial_config.shuffle:                 # envs/base.py:276-277
            random.shuffle(self.initial_ids)
        spec = build_spec(self, self.num_replicas)
        if self.FS_ENV is None:
            spec["env"] = L.FS_ENV_ACCEL
        self._spec = spec
        self.sim = FlowSim(spec, precision=self._precision(), device=getattr(self, "_device_index", 0))
        self.k.vehicle.attach(self.sim, 0)
        x0 = spec["init_pos"][0]
        open_net = spec.get("network") in ("merge", "bottleneck")
        # closed loops: slot i of the simulator holds k.vehicle._order[i] (id order unless shuffled / re-ordered)
        for i, veh_id in enumerate(self.initial_ids if open_net else self.k.vehicle._order):
            if open_net:
                slot = spec["init_slot"][veh_id]
                edge, pos = self.k.network.open_locate(int(spec["init_route"][0][slot]), float(x0[slot]))
            else:
                edge, pos = self.k.network.get_edge(float(x0[i]))
            self.initial_state[veh_id] = (self.k.vehicle.get_type(veh_id), edge, 0, pos,
                                          self.k.vehicle.get_initial_speed(veh_id))

    def restart_simulation(self, sim_params, render=None):
        """envs/base.py:231-266: rebuild geometry + simulator (e.g. after the network changed)."""
        self.k.generate_network(self.network)
        self.k.vehicle.initialize(deepcopy(self.network.vehicles))
        self.initial_vehicles = self.k.vehicle
        self.setup_initial_state()

    # ------------------------------------------------------------------ step / reset
    def _action_vector(self):
        pend = self.k.vehicle._pending
        if not pend:
            return None
        order = self._rl_action_order()
        if self.FS_ENV in (L.FS_ENV_LANE_CHANGE_ACCEL, L.FS_ENV_LANE_CHANGE_ACCEL_PO):          # [acc_0, dir_0, acc_1, dir_1, ...]
            lc = self.k.vehicle._pending_lc or {}
            row = []
            for v in order:
                row += [pend.get(v, 0.0), float(lc.get(v, 0))]
            return np.array([row], dtype=np.float32)
        return np.array([[pend.get(v, 0.0) for v in order]], dtype=np.float32)

    def step(self, rl_actions):
        """envs/base.py:294-412."""
        n_sub = self.env_params.sims_per_step
        self.time_counter += n_sub
        self.step_counter += n_sub
        self.apply_rl_actions(rl_actions)
        self.additional_command()
        obs, rew, done = self.sim.step(self._action_vector())
        self._after_sim_step()
        self.k.update(reset=False)
        limit = n_sub * (self.env_params.warmup_steps + self.env_params.horizon)
        tc = int(self.sim.time_counter[0])
        crash = bool(int(self.sim.last_done_flags[0]) & 2)      # the kernel reports a collision separately from the horizon
        self.time_counter = tc
        self._crash = crash
        self.k.simulation.crashed = crash
        self._last_obs, self._last_reward = obs[0], float(rew[0])
        if self._recorder is not None:
            self._sim_time += n_sub * self.sim_step
            self._recorder.record(self._sim_time)
        states = self.get_state()
        self.state = np.asarray(states).T
        next_observation = np.copy(states)
        done = (self.time_counter >= limit or crash)
        if self.env_params.clip_actions:
            reward = self.compute_reward(self.clip_actions(rl_actions), fail=crash)
        else:
            reward = self.compute_reward(rl_actions, fail=crash)
        return next_observation, reward, bool(done), {}

    def reset(self):
        """envs/base.py:414-560: initial placement, then warmup_steps steps with no RL action."""
        self.time_counter = 0
        if self.sim_params.restart_instance or self.step_counter > 2e6:
            self.step_counter = 0
            self.sim_params.seed = random.randint(0, int(1e5))
        elif self.initial_config.shuffle:               # envs/base.py:445-446: new placement, hence a new simulator
            self.setup_initial_state()
        obs = self.sim.reset()
        self.k.update(reset=True)
        self.time_counter = int(self.sim.time_counter[0])
        self._last_obs, self._last_reward, self._crash = obs[0], 0.0, False
        if self._recorder is not None:
            # SUMO inserts the vehicles during the reset step: first emission row is t = sim_step
            self._sim_time = self.sim_step * (1 + self.time_counter)
            self._recorder.record(self._sim_time)
        states = self.get_state()
        self.state = np.asarray(states).T
        return np.copy(states)

    def additional_command(self):
        pass

    def _after_sim_step(self):
        """What the reference's TraCI commands of this step do once the simulator has moved the vehicles (lane changes,
        insertions): nothing for most environments."""
        pass

    def clip_actions(self, rl_actions=None):
        """envs/base.py:566-597."""
        if rl_actions is None:
            return
        if isinstance(self.action_space, Box):
            rl_actions = np.clip(rl_actions, a_min=self.action_space.low, a_max=self.action_space.high)
        elif isinstance(self.action_space, Tuple):
            for idx, action in enumerate(rl_actions):
                subspace = self.action_space[idx]
                if isinstance(subspace, Box):
                    rl_actions[idx] = np.clip(action, a_min=subspace.low, a_max=subspace.high)
        return rl_actions

    def apply_rl_actions(self, rl_actions=None):
        """envs/base.py:599-615."""
        if rl_actions is None:
            return
        self._apply_rl_actions(self.clip_actions(rl_actions))

    def _apply_rl_actions(self, rl_actions):
        raise NotImplementedError

    def get_state(self):
        raise NotImplementedError

    def compute_reward(self, rl_actions, **kwargs):
        return 0

    def terminate(self):
        """envs/base.py:680-703."""
        if getattr(self, "sim", None) is not None:
            self.sim.close()
            self.sim = None

    def write_emission(self, path=None):
        """Write the recorded trajectory as ``{network.name}-emission.csv`` under
        ``sim_params.emission_path`` (the file Experiment.run(convert_to_csv=True) leaves behind,
        flow/core/experiment.py:182-196)."""
        if self._recorder is None:
            return None
        if path is None:
            import os
            path = os.path.join(self.sim_params.emission_path, "{0}-emission.csv".format(self.network.name))
        return self._recorder.write(path)

    def render(self, reset=False, buffer_length=5):
        pass

    def close(self):
        self.terminate()


def redraw_ring(env, length, bunching=None, min_gap=None):
    """Re-place env's vehicles on a ring of ``length`` (WaveAttenuationEnv.reset,
    flow/envs/ring/wave_attenuation.py:170-207) without re-creating the simulator."""
    from flow_amd.core.params import InitialConfig, NetParams
    add = dict(env.net_params.additional_params)
    add["length"] = length
    net_params = NetParams(additional_params=add)
    ic = InitialConfig(bunching=50 if bunching is None else bunching, min_gap=0 if min_gap is None else min_gap)
    env.network = env.network.__class__(env.network.orig_name, env.network.vehicles, net_params, ic)
    env.net_params = net_params
    env.initial_config = ic
    env.k.generate_network(env.network)
    N = env.k.vehicle.num_vehicles
    X, lanes = initial_positions(env.k.network, ic, N, 1)
    check_placement(X, np.array([s["length"] for s in env._spec["vehicles"]]), env.k.network.length(), lanes)
    return X
