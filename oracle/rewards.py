"""Oracle restatement of flow/core/rewards.py and the env reward functions.

TEST INFRASTRUCTURE ONLY.  Vectorised over a leading replica axis: ``vel`` has
shape [..., N]; results have shape [...].
"""
import numpy as np

EPS_F32 = float(np.finfo(np.float32).eps)      # rewards.py:57


def tree_sum(a):
    """Sum over the last axis in the kernels' order: pad to a power of two with
    zeros, then add neighbours pairwise (the xor-butterfly's tree).  For
    float64 inputs this differs from numpy's own summation by rounding only."""
    a = np.asarray(a)
    n = a.shape[-1]
    seg = 1
    while seg < n:
        seg *= 2
    if seg != n:
        pad = np.zeros(a.shape[:-1] + (seg - n,), dtype=a.dtype)
        a = np.concatenate([a, pad], axis=-1)
    while a.shape[-1] > 1:
        a = a[..., 0::2] + a[..., 1::2]
    return a[..., 0]


def desired_velocity(vel, target_velocity, fail=False):
    """rewards.desired_velocity, flow/core/rewards.py:6-59 (edge_list=None).

    ``fail`` may be a boolean array over replicas.
    """
    vel = np.asarray(vel)
    dt = vel.dtype
    n = vel.shape[-1]
    if n == 0:
        return np.zeros(vel.shape[:-1], dtype=dt)
    # np.linalg.norm([target_vel] * n), evaluated in float64 on the host
    max_cost = np.asarray(np.linalg.norm(np.array([target_velocity] * n, dtype=np.float64)),
                          dtype=dt)                                 # :50-51
    d = vel - np.asarray(target_velocity, dtype=dt)                 # :53
    cost = np.sqrt(tree_sum(d * d))                                 # :54
    r = np.maximum(max_cost - cost, np.asarray(0, dt)) / (max_cost + np.asarray(EPS_F32, dt))  # :59
    bad = np.any(vel < -100, axis=-1) | np.asarray(fail)            # :46
    return np.where(bad, np.asarray(0, dt), r)


def average_velocity(vel, fail=False):
    """rewards.average_velocity, flow/core/rewards.py:62-88."""
    vel = np.asarray(vel)
    dt = vel.dtype
    if vel.shape[-1] == 0:
        return np.zeros(vel.shape[:-1], dtype=dt)
    m = tree_sum(vel) / np.asarray(vel.shape[-1], dt)
    bad = np.any(vel < -100, axis=-1) | np.asarray(fail)
    return np.where(bad, np.asarray(0, dt), m)


def wave_attenuation_reward(vel, rl_actions, fail=False):
    """WaveAttenuationEnv.compute_reward, flow/envs/ring/wave_attenuation.py:113-139.

    ``rl_actions`` is None (warm-up: reward 0) or an array [..., n_rl].
    """
    vel = np.asarray(vel)
    dt = vel.dtype
    if rl_actions is None:                                          # :116-117
        return np.zeros(vel.shape[:-1], dtype=dt)
    rl_actions = np.asarray(rl_actions, dtype=dt)
    mean_v = tree_sum(vel) / np.asarray(vel.shape[-1], dt)
    reward = np.asarray(4.0, dt) * mean_v / np.asarray(20, dt)      # :128-129
    mean_a = tree_sum(np.abs(rl_actions)) / np.asarray(rl_actions.shape[-1], dt)  # :133
    reward = np.where(mean_a > 0, reward + np.asarray(4, dt) * (np.asarray(0, dt) - mean_a),
                      reward)                                       # :136-137
    bad = np.any(vel < -100, axis=-1) | np.asarray(fail)            # :124-125
    return np.where(bad, np.asarray(0, dt), reward)


def v_eq_max_function(v, num_vehicles, length):
    """flow/envs/ring/wave_attenuation.py:33-47."""
    s_eq_max = (length - num_vehicles * 5) / (num_vehicles - 1)
    v0, s0, tau, gamma = 30, 2, 1, 4
    return s_eq_max - (s0 + v * tau) * (1 - (v / v0) ** gamma) ** -0.5


def min_delay(vel, v_top, sim_step):
    """rewards.min_delay, flow/core/rewards.py:117-148 (single replica)."""
    vel = np.asarray(vel, dtype=np.float64)
    vel = vel[vel >= -1e-6]
    max_cost = sim_step * sum(vel.shape)
    cost = sim_step * sum((v_top - vel) / v_top)
    return max((max_cost - cost) / (max_cost + EPS_F32), 0)


def penalize_standstill(vel, gain=1):
    """rewards.penalize_standstill, flow/core/rewards.py:208-232."""
    vel = np.asarray(vel)
    return -gain * int(np.sum(vel == 0))


def penalize_near_standstill(vel, thresh=0.3, gain=1):
    """rewards.penalize_near_standstill, flow/core/rewards.py:235-256."""
    vel = np.asarray(vel)
    return -gain * int(np.sum(vel < thresh))


def energy_consumption(speed, prev_speed, sim_step, gain=.001):
    """rewards.energy_consumption, flow/core/rewards.py:309-332."""
    M, g, Cr, Ca, rho, A = 1200, 9.81, 0.005, 0.3, 1.225, 2.6
    power = 0
    for s, p in zip(speed, prev_speed):
        accel = abs(s - p) / sim_step
        power += M * s * accel + M * g * Cr * s + 0.5 * rho * A * Ca * s ** 3
    return -gain * power
