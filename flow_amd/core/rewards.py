"""Host-side reward helpers under the names and signatures of flow/core/rewards.py (cited per function), for
user-defined environments whose ``compute_reward`` is a Python hook.  The built-in environments compute their reward
inside the step kernel; nothing on the simulation path calls these.

Each helper is ONE reduction over the replica's speed array (``env.k.vehicle.speeds()``: a view of the simulator's state
row fetched once per step), not a walk over vehicle ids."""
import numpy as np

_EPS = float(np.finfo(np.float32).eps)


def _speeds(env, edge_list=None):
    vk = env.k.vehicle
    return vk.speeds() if edge_list is None else vk.speeds(vk.get_ids_by_edge(edge_list))


def _valid(v, fail):
    return not (fail or v.size == 0 or (v < -100).any())


def desired_velocity(env, fail=False, edge_list=None):
    """1 - ||v - v_target|| / ||v_target 1||, floored at 0 (flow/core/rewards.py:6-59)."""
    v = _speeds(env, edge_list)
    if not _valid(v, fail):
        return 0.
    target = float(env.env_params.additional_params['target_velocity'])
    top = np.sqrt(v.size * target * target)
    return max(top - np.sqrt(np.square(v - target).sum()), 0.) / (top + _EPS)


def average_velocity(env, fail=False):
    """Mean speed (flow/core/rewards.py:62-88)."""
    v = _speeds(env)
    return float(v.mean()) if _valid(v, fail) else 0.


def rl_forward_progress(env, gain=0.1):
    """gain * sum |v| over the RL vehicles (flow/core/rewards.py:91-109)."""
    vk = env.k.vehicle
    return gain * float(np.abs(vk.speeds(vk.get_rl_ids())).sum())


def boolean_action_penalty(discrete_actions, gain=1.0):
    """gain * number of non-zero actions (flow/core/rewards.py:112-114)."""
    return gain * np.sum(discrete_actions)


def min_delay(env):
    """1 - mean relative delay against the network's top speed limit (flow/core/rewards.py:117-148)."""
    v = _speeds(env)
    v = v[v >= -1e-6]
    net = env.k.network
    v_top = max(net.speed_limit(e) for e in net.get_edge_list())
    dt = env.sim_step
    top = dt * v.size
    return max((top - dt * ((v_top - v) / v_top).sum()) / (top + _EPS), 0)


def penalize_standstill(env, gain=1):
    """-gain * number of standing vehicles (flow/core/rewards.py:208-232)."""
    return -gain * int(np.count_nonzero(_speeds(env) == 0))


def penalize_near_standstill(env, thresh=0.3, gain=1):
    """-gain * number of vehicles slower than ``thresh`` (flow/core/rewards.py:235-256)."""
    return -gain * int(np.count_nonzero(_speeds(env) < thresh))


def energy_consumption(env, gain=.001):
    """-gain * sum of the per-vehicle power model (flow/core/rewards.py:309-332): inertia M v |dv/dt|, rolling
    resistance M g Cr v, drag rho A Ca v^3 / 2."""
    M, g, Cr, Ca, rho, A = 1200, 9.81, 0.005, 0.3, 1.225, 2.6
    vk = env.k.vehicle
    v, v_prev = vk.speeds(), vk.previous_speeds()
    accel = np.abs(v - v_prev) / env.sim_step
    return -gain * float((M * v * accel + M * g * Cr * v + 0.5 * rho * A * Ca * v ** 3).sum())
