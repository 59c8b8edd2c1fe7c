"""Reward functions with the signatures of flow/core/rewards.py, for user-defined (Python-hook)
environments.  The built-in environments compute their reward inside the step kernel; these
helpers evaluate on the host from the k.vehicle view and exist so custom ``compute_reward``
implementations written against the reference keep working."""
import numpy as np


def desired_velocity(env, fail=False, edge_list=None):
    """flow/core/rewards.py:6-59."""
    veh_ids = env.k.vehicle.get_ids() if edge_list is None else env.k.vehicle.get_ids_by_edge(edge_list)
    vel = np.array(env.k.vehicle.get_speed(veh_ids))
    num_vehicles = len(veh_ids)
    if any(vel < -100) or fail or num_vehicles == 0:
        return 0.
    target_vel = env.env_params.additional_params['target_velocity']
    max_cost = np.linalg.norm(np.array([target_vel] * num_vehicles))
    cost = np.linalg.norm(vel - target_vel)
    eps = np.finfo(np.float32).eps
    return max(max_cost - cost, 0) / (max_cost + eps)


def average_velocity(env, fail=False):
    """flow/core/rewards.py:62-88."""
    vel = np.array(env.k.vehicle.get_speed(env.k.vehicle.get_ids()))
    if any(vel < -100) or fail or len(vel) == 0:
        return 0.
    return np.mean(vel)


def rl_forward_progress(env, gain=0.1):
    """flow/core/rewards.py:91-109."""
    return np.linalg.norm(env.k.vehicle.get_speed(env.k.vehicle.get_rl_ids()), 1) * gain


def boolean_action_penalty(discrete_actions, gain=1.0):
    """flow/core/rewards.py:112-114."""
    return gain * np.sum(discrete_actions)


def min_delay(env):
    """flow/core/rewards.py:117-148."""
    vel = np.array(env.k.vehicle.get_speed(env.k.vehicle.get_ids()))
    vel = vel[vel >= -1e-6]
    v_top = max(env.k.network.speed_limit(edge) for edge in env.k.network.get_edge_list())
    time_step = env.sim_step
    max_cost = time_step * sum(vel.shape)
    eps = np.finfo(np.float32).eps
    cost = time_step * sum((v_top - vel) / v_top)
    return max((max_cost - cost) / (max_cost + eps), 0)


def penalize_standstill(env, gain=1):
    """flow/core/rewards.py:208-232."""
    vel = np.array(env.k.vehicle.get_speed(env.k.vehicle.get_ids()))
    return -gain * len(vel[vel == 0])


def penalize_near_standstill(env, thresh=0.3, gain=1):
    """flow/core/rewards.py:235-256."""
    vel = np.array(env.k.vehicle.get_speed(env.k.vehicle.get_ids()))
    return -gain * len(vel[vel < thresh])


def energy_consumption(env, gain=.001):
    """flow/core/rewards.py:309-332."""
    M, g, Cr, Ca, rho, A = 1200, 9.81, 0.005, 0.3, 1.225, 2.6
    power = 0
    for veh_id in env.k.vehicle.get_ids():
        speed = env.k.vehicle.get_speed(veh_id)
        accel = abs(speed - env.k.vehicle.get_previous_speed(veh_id)) / env.sim_step
        power += M * speed * accel + M * g * Cr * speed + 0.5 * rho * A * Ca * speed ** 3
    return -gain * power
