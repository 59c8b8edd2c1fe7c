"""Oracle restatement of flow/controllers (TEST INFRASTRUCTURE ONLY).

Every function is a numpy restatement of one reference controller's
``get_accel`` (or of a ``BaseController`` fail-safe), vectorised over vehicles.
Inputs are arrays of the quantities the reference reads through
``env.k.vehicle``:

    v         own speed                      get_speed(veh_id)
    v_lead    leader speed                   get_speed(get_leader(veh_id))
    h         bumper-to-bumper headway       get_headway(veh_id)
    has_lead  leader is not None             get_leader(veh_id)
    v_follow  follower speed                 get_speed(get_follower(veh_id))
    h_follow  follower's headway             get_headway(get_follower(veh_id))

The dtype of the inputs is the dtype of the arithmetic: float64 restates the
reference (Python floats); float32 is the bit-twin of the HIP kernels (numpy
never contracts a*b+c into an fma, and neither do the kernels, which are built
with -ffp-contract=off).
"""
import numpy as np


def _c(x, like):
    """Constant cast to the working dtype (scalar parameters enter rounded)."""
    return np.asarray(x, dtype=like.dtype)


def pow_delta(x, delta):
    """x**delta.  Integer exponents 1..8 by repeated squaring/multiplication in
    a fixed order shared with the kernels; anything else through pow()."""
    d = float(delta)
    if d == 4.0:
        x2 = x * x
        return x2 * x2
    if d == 2.0:
        return x * x
    if d == 1.0:
        return x
    if d == 3.0:
        return (x * x) * x
    if d == 8.0:
        x2 = x * x
        x4 = x2 * x2
        return x4 * x4
    return np.power(x, _c(d, x))


# ---------------------------------------------------------------------------
# car-following models  (reference: flow/controllers/car_following_models.py)
# ---------------------------------------------------------------------------

def idm(v, v_lead, h, has_lead, v0=30, T=1, a=1, b=1.5, delta=4, s0=2):
    """IDMController.get_accel, car_following_models.py:464-482."""
    v = np.asarray(v)
    h = np.where(np.abs(h) < _c(1e-3, v), _c(1e-3, v), h)          # :471-472
    two_sqrt_ab = _c(2, v) * np.sqrt(_c(a, v) * _c(b, v))          # :480
    dyn = v * _c(T, v) + v * (v - v_lead) / two_sqrt_ab            # :479-480
    s_star = np.where(has_lead,
                      _c(s0, v) + np.maximum(_c(0, v), dyn),       # :478
                      _c(0, v))                                    # :475
    q = s_star / h
    return _c(a, v) * (_c(1, v) - pow_delta(v / _c(v0, v), delta) - q * q)  # :482


def cfm(v, v_lead, h, has_lead, max_accel, k_d=1, k_v=1, k_c=1, d_des=1, v_des=8):
    """CFMController.get_accel, car_following_models.py:76-88."""
    v = np.asarray(v)
    acc = (_c(k_d, v) * (h - _c(d_des, v)) + _c(k_v, v) * (v_lead - v)
           + _c(k_c, v) * (_c(v_des, v) - v))
    return np.where(has_lead, acc, _c(max_accel, v))               # :79-80


def bcm(v, v_lead, h, has_lead, v_follow, h_follow, max_accel,
        k_d=1, k_v=1, k_c=1, d_des=1, v_des=8):
    """BCMController.get_accel, car_following_models.py:152-176."""
    v = np.asarray(v)
    acc = (_c(k_d, v) * (h - h_follow)
           + _c(k_v, v) * ((v_lead - v) - (v - v_follow))
           + _c(k_c, v) * (_c(v_des, v) - v))
    return np.where(has_lead, acc, _c(max_accel, v))


def lac(v, v_lead, h, veh_len, a_prev, dt, k_1=0.3, k_2=0.4, h_gap=1, tau=0.1):
    """LACController.get_accel, car_following_models.py:232-245.

    Stateful: returns the new ``a`` (which is both the state and the command).
    """
    v = np.asarray(v)
    ex = h - veh_len - _c(h_gap, v) * v                            # :239
    ev = v_lead - v                                                # :240
    u = _c(k_1, v) * ex + _c(k_2, v) * ev                          # :241
    a_dot = -(a_prev / _c(tau, v)) + (u / _c(tau, v))              # :242
    return a_dot * _c(dt, v) + a_prev                              # :243


def ovm(v, v_lead, h, has_lead, max_accel, alpha=1, beta=1, h_st=2, h_go=15, v_max=30):
    """OVMController.get_accel, car_following_models.py:308-328."""
    v = np.asarray(v)
    h_dot = v_lead - v
    mid = (_c(v_max, v) / _c(2, v)
           * (_c(1, v) - np.cos(_c(np.pi, v) * (h - _c(h_st, v))
                                / (_c(h_go, v) - _c(h_st, v)))))    # :323-324
    v_h = np.where(h <= _c(h_st, v), _c(0, v),
                   np.where(h < _c(h_go, v), mid, _c(v_max, v)))   # :320-326
    acc = _c(alpha, v) * (v_h - v) + _c(beta, v) * h_dot
    return np.where(has_lead, acc, _c(max_accel, v))


def linear_ovm(v, h, v_max=30, adaptation=0.65, h_st=5):
    """LinearOVM.get_accel, car_following_models.py:383-397."""
    v = np.asarray(v)
    alpha = _c(1.689, v)                                           # :389
    upper = _c(h_st, v) + _c(v_max, v) / alpha
    v_h = np.where(h < _c(h_st, v), _c(0, v),
                   np.where(h <= upper, alpha * (h - _c(h_st, v)), _c(v_max, v)))
    return (v_h - v) / _c(adaptation, v)


def gipps(v, v_lead, h, dt, v0=30, acc=1.5, b=-1, b_l=-1, s0=2, tau=1):
    """GippsController.get_accel, car_following_models.py:567-582."""
    v = np.asarray(v)
    one = _c(1, v)
    r = v / _c(v0, v)
    v_acc = v + (_c(2.5, v) * _c(acc, v) * _c(tau, v) * (one - r)
                 * np.sqrt(_c(0.025, v) + r))                      # :575-576
    tb = _c(tau, v) * _c(b, v)
    disc = (_c(tau, v) * _c(tau, v)) * (_c(b, v) * _c(b, v)) - (
        _c(b, v) * ((_c(2, v) * (h - _c(s0, v))) - (_c(tau, v) * v)
                    - ((v_lead * v_lead) / _c(b_l, v))))           # :577-578
    with np.errstate(invalid="ignore"):
        v_safe = tb + np.sqrt(disc)
    # Python's min(v_acc, v_safe, v0) skips a NaN v_safe (NaN < x is False): fmin
    v_next = np.fmin(np.fmin(v_acc, v_safe), _c(v0, v))            # :580
    return (v_next - v) / _c(dt, v)                                # :582


# ---------------------------------------------------------------------------
# velocity controllers  (reference: flow/controllers/velocity_controllers.py)
# ---------------------------------------------------------------------------

def follower_stopper_vcmd(v, v_lead, h, has_lead, v_des):
    """The v_cmd part of FollowerStopper.get_accel, velocity_controllers.py:84-103.

    ``v_des`` may be an array (NonLocalFollowerStopper uses the replica mean
    speed, velocity_controllers.py:127).
    """
    v = np.asarray(v)
    v_des = np.asarray(v_des, dtype=v.dtype)
    dv_minus = np.minimum(v_lead - v, _c(0, v))                    # :88
    dv2 = dv_minus * dv_minus
    # 1 / (2 * d_k) evaluated in double by the reference, then multiplied
    dx_1 = _c(4.5, v) + _c(1 / (2 * 1.5), v) * dv2                 # :90
    dx_2 = _c(5.25, v) + _c(1 / (2 * 1.0), v) * dv2                # :91
    dx_3 = _c(6.0, v) + _c(1 / (2 * 0.5), v) * dv2                 # :92
    vv = np.minimum(np.maximum(v_lead, _c(0, v)), v_des)           # :93
    with np.errstate(invalid="ignore", divide="ignore"):
        c2 = vv * (h - dx_1) / (dx_2 - dx_1)                       # :98
        c3 = vv + (v_des - v) * (h - dx_2) / (dx_3 - dx_2)         # :100-101
    v_cmd = np.where(h <= dx_1, _c(0, v),
                     np.where(h <= dx_2, c2,
                              np.where(h <= dx_3, c3, v_des)))
    return np.where(has_lead, v_cmd, v_des)                        # :84-85


def follower_stopper(v, v_lead, h, has_lead, dt, v_des=15):
    """FollowerStopper.get_accel, velocity_controllers.py:75-116 (no danger
    edges; the junction / empty-edge ``None`` cases are handled by the caller)."""
    v = np.asarray(v)
    return (follower_stopper_vcmd(v, v_lead, h, has_lead, v_des) - v) / _c(dt, v)  # :116


class PISaturationState:
    """PISaturation controller state, velocity_controllers.py:185-206."""

    def __init__(self, shape, dtype=np.float64):
        self.history = []            # list of arrays, one per step  (:193)
        self.v_cmd = np.zeros(shape, dtype=dtype)                  # :206


def pi_saturation(state, v, v_lead, h, dt, max_accel):
    """PISaturation.get_accel, velocity_controllers.py:208-240."""
    v = np.asarray(v)
    dv = v_lead - v
    dx_s = np.maximum(_c(2, v) * dv, _c(4, v))                     # :216
    state.history.append(np.array(v, copy=True))                   # :219
    if len(state.history) == int(38 / dt):                         # :221-222
        del state.history[0]
    v_des = np.mean(np.stack(state.history, 0), axis=0)            # :225
    g_l, g_u, gamma, v_catch = 7, 30, 2, 1
    v_target = v_des + _c(v_catch, v) * np.minimum(
        np.maximum((h - _c(g_l, v)) / _c(g_u - g_l, v), _c(0, v)), _c(1, v))  # :226-227
    alpha = np.minimum(np.maximum((h - dx_s) / _c(gamma, v), _c(0, v)), _c(1, v))  # :230
    beta = _c(1, v) - _c(0.5, v) * alpha                           # :231
    state.v_cmd = beta * (alpha * v_target + (_c(1, v) - alpha) * v_lead) \
        + (_c(1, v) - beta) * state.v_cmd                          # :234-235
    accel = (state.v_cmd - v) / _c(dt, v)                          # :238
    return np.minimum(accel, _c(max_accel, v))                     # :240


# ---------------------------------------------------------------------------
# fail-safes  (reference: flow/controllers/base_controller.py)
# ---------------------------------------------------------------------------

def failsafe_instantaneous(acc, v, h, has_lead, dt, num_vehicles):
    """get_safe_action_instantaneous, base_controller.py:120-169."""
    v = np.asarray(v)
    if num_vehicles == 1:                                          # :141-142
        return acc
    dt_ = _c(dt, v)
    next_vel = v + acc * dt_                                       # :152
    thresh = dt_ * next_vel + v * _c(1e-3, v) + _c(0.5, v) * v * dt_  # :158-159
    stop = has_lead & (next_vel > 0) & (h < thresh)                # :147,155,158
    return np.where(stop, -v / dt_, acc)                           # :163


def failsafe_safe_velocity(acc, v, v_lead, h, dt, delay, num_vehicles):
    """get_safe_velocity_action + safe_velocity, base_controller.py:171-236."""
    v = np.asarray(v)
    if num_vehicles == 1:                                          # :191-193
        return acc
    dt_ = _c(dt, v)
    dv = v_lead - v                                                # :232
    v_safe = _c(2, v) * h / dt_ + dv - v * (_c(2, v) * _c(delay, v))   # :234
    over = (v + acc * dt_) > v_safe                                # :200
    clipped = np.where(v_safe > 0, (v_safe - v) / dt_, -v / dt_)   # :201-204
    return np.where(over, clipped, acc)


# ---------------------------------------------------------------------------
# SUMO-side car following for *uncommanded* vehicles
# ---------------------------------------------------------------------------

def sumo_idm_speed(v, v_lead, h, has_lead, dt, accel=2.6, decel=4.5, tau=1.0,
                   min_gap=2.5, max_speed=30.0, delta=4):
    """Speed SUMO gives a vehicle that received no command this step.

    NOT reference code: the reference hands such vehicles to SUMO
    (base_controller.py:93-106; vehicle/traci.py:960), whose vType is
    carFollowModel="IDM" with the parameters of SumoCarFollowingParams
    (core/params.py:839-891).  This restates the published IDM as SUMO's
    MSCFModel_IDM applies it for one iteration per step (sim_step <= 0.25 s):
    gap is bumper-to-bumper, desired gap s* = minGap + max(0, v*tau +
    v*dv/(2*sqrt(accel*decel))), v' = max(0, v + a_idm*dt).  PARITY UNPINNED
    (SUMO not available); see docs/HISTORY.md S7.
    """
    v = np.asarray(v)
    gap = np.maximum(h, _c(1e-3, v))
    two_sqrt = _c(2, v) * np.sqrt(_c(accel, v) * _c(decel, v))
    s = _c(min_gap, v) + np.maximum(_c(0, v), v * _c(tau, v) + v * (v - v_lead) / two_sqrt)
    q = np.where(has_lead, s / gap, _c(0, v))
    acc = _c(accel, v) * (_c(1, v) - pow_delta(v / _c(max_speed, v), delta) - q * q)
    return np.maximum(_c(0, v), v + acc * _c(dt, v))
