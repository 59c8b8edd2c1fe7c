"""ctypes binding of libflowsim.so (include/flowsim.h).

The library is the product: if it is missing or cannot be loaded this module
raises -- there is no Python / CPU fallback for the simulation path.
"""
import ctypes as C
import os

from .utils.exceptions import FatalFlowError

PKG = os.path.dirname(os.path.abspath(__file__))
# FLOWSIM_LIB: a development build of the same sources (e.g. the phase-timer build of scripts/phase_open.py)
LIB_PATH = os.environ.get("FLOWSIM_LIB") or os.path.join(PKG, "libflowsim.so")

FS_ABI_VERSION = 8
FS_MAX_CTRL_PARAMS = 8

# error codes
FS_OK, FS_ERR_INVALID, FS_ERR_UNSUPPORTED, FS_ERR_HIP, FS_ERR_NOSPACE = 0, -1, -2, -3, -4
# enum fs_precision
FS_F32, FS_F64, FS_MIXED, FS_F16S = 0, 1, 2, 3
# enum fs_controller
(FS_CTRL_SIM, FS_CTRL_RL, FS_CTRL_IDM, FS_CTRL_CFM, FS_CTRL_BCM, FS_CTRL_LAC, FS_CTRL_OVM,
 FS_CTRL_LINEAR_OVM, FS_CTRL_GIPPS, FS_CTRL_FOLLOWER_STOPPER, FS_CTRL_NONLOCAL_FOLLOWER_STOPPER,
 FS_CTRL_PISATURATION) = range(12)
FS_CTRL_USER = 12
# enum fs_failsafe
FS_FAILSAFE_NONE, FS_FAILSAFE_INSTANTANEOUS, FS_FAILSAFE_SAFE_VELOCITY = range(3)
# enum fs_env
(FS_ENV_ACCEL, FS_ENV_WAVE_ATTENUATION, FS_ENV_WAVE_ATTENUATION_PO, FS_ENV_LANE_CHANGE_ACCEL, FS_ENV_MERGE_PO,
 FS_ENV_MERGE_MA, FS_ENV_BOTTLENECK_DV, FS_ENV_BOTTLENECK, FS_ENV_WAVE_ATTENUATION_PO_MA,
 FS_ENV_ACCEL_PO_MA, FS_ENV_LANE_CHANGE_ACCEL_PO) = range(11)
# enum fs_network / fs_integrator
FS_NET_RING, FS_NET_FIGURE_EIGHT, FS_NET_MERGE, FS_NET_BOTTLENECK = 0, 1, 2, 3
FS_MAX_SEGMENTS = 16
FS_MAX_INFLOWS = 8
FS_EULER, FS_BALLISTIC = 0, 1
# enum fs_field
(FS_FIELD_POS, FS_FIELD_VEL, FS_FIELD_HEADWAY, FS_FIELD_PREV_VEL, FS_FIELD_ACCEL, FS_FIELD_TIME,
 FS_FIELD_RING_LENGTH, FS_FIELD_INIT_POS, FS_FIELD_INIT_VEL, FS_FIELD_CTRL_STATE, FS_FIELD_LANE,
 FS_FIELD_LAST_LC, FS_FIELD_LEADER, FS_FIELD_INIT_LANE, FS_FIELD_ROUTE, FS_FIELD_SEQ, FS_FIELD_ORIGIN,
 FS_FIELD_FOLLOWER, FS_FIELD_CTL_SEQ, FS_FIELD_COUNTERS, FS_FIELD_ARRIVED_RL, FS_FIELD_MAX_SPEED,
 FS_FIELD_INIT_RING_LENGTH, FS_FIELD_SORT_KEY) = range(24)

EXPORTS = ["fs_create", "fs_destroy", "fs_last_error", "fs_abi_version", "fs_obs_dim", "fs_action_dim", "fs_set_stream",
           "fs_use_own_stream", "fs_sync", "fs_reset", "fs_reset_dev", "fs_step", "fs_step_dev", "fs_rollout_dev",
           "fs_get_state", "fs_set_state", "fs_add_vehicle", "fs_dump_trajectory", "fs_last_kernel", "fs_policy_act_dev",
           "fs_policy_rollout_dev"]


class fs_vehicle_spec(C.Structure):
    _fields_ = [("controller", C.c_int32), ("fail_safe", C.c_int32), ("speed_mode", C.c_int32),
                ("rl_index", C.c_int32), ("type", C.c_int32), ("lane_change_mode", C.c_int32),
                ("p", C.c_double * FS_MAX_CTRL_PARAMS), ("noise", C.c_double),
                ("delay", C.c_double), ("max_accel", C.c_double), ("max_decel", C.c_double),
                ("length", C.c_double), ("sumo_tau", C.c_double), ("sumo_min_gap", C.c_double),
                ("sumo_max_speed", C.c_double), ("initial_speed", C.c_double)]


class fs_segment(C.Structure):
    _fields_ = [("start", C.c_double), ("flow_start", C.c_double), ("flow_slope", C.c_double),
                ("internal", C.c_int32), ("route", C.c_int32)]


class fs_inflow(C.Structure):
    _fields_ = [("type", C.c_int32), ("route", C.c_int32), ("number", C.c_int32), ("reserved", C.c_int32),
                ("period", C.c_double), ("begin", C.c_double), ("end", C.c_double), ("depart_speed", C.c_double),
                ("depart_pos", C.c_double), ("probability", C.c_double)]


class fs_cell(C.Structure):
    _fields_ = [("edge_start", C.c_double), ("lo", C.c_double), ("hi", C.c_double), ("lane", C.c_int32),
                ("last_segment", C.c_int32)]


class fs_junction(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("reserved", C.c_int32), ("a_in", C.c_double), ("a_out", C.c_double),
                ("b_in", C.c_double), ("b_out", C.c_double), ("lookahead", C.c_double), ("time_gap", C.c_double),
                ("za_lo", C.c_double), ("za_hi", C.c_double), ("zb_lo", C.c_double), ("zb_hi", C.c_double)]


class fs_config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("abi_version", C.c_uint32), ("precision", C.c_int32),
                ("network", C.c_int32), ("env", C.c_int32), ("integrator", C.c_int32),
                ("num_replicas", C.c_int32), ("num_vehicles", C.c_int32), ("num_rl", C.c_int32),
                ("horizon", C.c_int32), ("warmup_steps", C.c_int32), ("sims_per_step", C.c_int32),
                ("junction_mode", C.c_int32), ("clip_actions", C.c_int32), ("evaluate", C.c_int32),
                ("device", C.c_int32), ("track_aux", C.c_int32), ("num_lanes", C.c_int32),
                ("lane_change_mode", C.c_int32), ("last_lc_quirk", C.c_int32),
                ("seed", C.c_uint64), ("sim_step", C.c_double), ("slowdown_ramp", C.c_double),
                ("junction_length", C.c_double), ("crash_gap", C.c_double), ("max_speed", C.c_double),
                ("target_velocity", C.c_double), ("action_low", C.c_double), ("action_high", C.c_double),
                ("po_max_length", C.c_double), ("lane_change_duration", C.c_double),
                ("vehicles", C.POINTER(fs_vehicle_spec)),
                ("ring_length", C.POINTER(C.c_double)), ("init_pos", C.POINTER(C.c_double)),
                ("init_vel", C.POINTER(C.c_double)), ("init_lane", C.POINTER(C.c_int32)),
                ("segments", C.POINTER(fs_segment)), ("num_segments", C.c_int32), ("num_inflows", C.c_int32),
                ("junction", fs_junction),
                ("inflows", C.POINTER(fs_inflow)), ("init_alive", C.POINTER(C.c_uint8)),
                ("route_start", C.c_double * 2), ("merge_x", C.c_double), ("box_in", C.c_double),
                ("end_x", C.c_double), ("net_length", C.c_double), ("ma_apply_actions", C.c_int32),
                ("num_obs_cells", C.c_int32), ("merge1_x", C.c_double), ("merge2_x", C.c_double),
                ("zipper_distance", C.c_double), ("speed_limit", C.c_double), ("outflow_norm", C.c_double),
                ("obs_cells", C.POINTER(fs_cell)), ("act_cells", C.POINTER(fs_cell)),
                ("obs_outflow_window", C.c_int32), ("reward_outflow_window", C.c_int32),
                ("track_followers", C.c_int32), ("num_paths", C.c_int32),
                ("lane_change_cooldown_steps", C.c_int32), ("reserved6", C.c_int32), ("lane_change_min_gain", C.c_double),
                ("sort_vehicles", C.c_int32),
                ("noise_exact", C.c_int32), ("obs_perm", C.POINTER(C.c_int32)), ("replica_offset", C.c_int64)]


class fs_policy(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("obs_dim", C.c_int32), ("num_hidden", C.c_int32),
                ("hidden_width", C.c_int32), ("activation", C.c_int32), ("weights_dev", C.c_void_p),
                ("log_std_dev", C.c_void_p), ("seed", C.c_uint64)]


_libs = {}


def load(path=None):
    """Load libflowsim.so (or a copy of it built with a user controller: flow_amd.build.build_user) once; raise if it is
    absent (the HIP path is the only path)."""
    path = LIB_PATH if path is None else os.path.abspath(path)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise FatalFlowError(
            "libflowsim.so not found at %s: build it with `python -m flow_amd.build` "
            "(hipcc, gfx950).  flow_amd has no CPU fallback." % path)
    try:                       # share torch's HIP runtime (same SONAME) when torch is in the process
        import torch  # noqa: F401
    except Exception:          # pragma: no cover - torch is optional for the C ABI itself
        pass
    lib = C.CDLL(path)
    h = C.c_void_p
    f32p, u8p = C.c_void_p, C.c_void_p       # raw addresses: host numpy or device pointers
    lib.fs_create.argtypes = [C.POINTER(fs_config), C.POINTER(h)]
    lib.fs_create.restype = C.c_int
    lib.fs_destroy.argtypes = [h]
    lib.fs_destroy.restype = None
    lib.fs_last_error.argtypes = []
    lib.fs_last_error.restype = C.c_char_p
    lib.fs_abi_version.argtypes = []
    lib.fs_abi_version.restype = C.c_int
    lib.fs_obs_dim.argtypes = [h]
    lib.fs_obs_dim.restype = C.c_int
    lib.fs_action_dim.argtypes = [h]
    lib.fs_action_dim.restype = C.c_int
    lib.fs_set_stream.argtypes = [h, C.c_void_p]
    lib.fs_set_stream.restype = C.c_int
    lib.fs_use_own_stream.argtypes = [h]
    lib.fs_use_own_stream.restype = C.c_int
    lib.fs_sync.argtypes = [h]
    lib.fs_sync.restype = C.c_int
    lib.fs_reset.argtypes = [h, u8p, f32p]
    lib.fs_reset.restype = C.c_int
    lib.fs_reset_dev.argtypes = [h, u8p, f32p]
    lib.fs_reset_dev.restype = C.c_int
    lib.fs_step.argtypes = [h, f32p, f32p, f32p, u8p]
    lib.fs_step.restype = C.c_int
    lib.fs_step_dev.argtypes = [h, f32p, f32p, f32p, u8p]
    lib.fs_step_dev.restype = C.c_int
    lib.fs_rollout_dev.argtypes = [h, C.c_int, f32p, C.c_size_t, f32p, f32p, u8p, C.c_int]
    lib.fs_rollout_dev.restype = C.c_int
    lib.fs_get_state.argtypes = [h, C.c_int, C.c_void_p, C.c_size_t]
    lib.fs_get_state.restype = C.c_int
    lib.fs_set_state.argtypes = [h, C.c_int, C.c_void_p, C.c_size_t]
    lib.fs_set_state.restype = C.c_int
    lib.fs_add_vehicle.argtypes = [h, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
    lib.fs_add_vehicle.restype = C.c_int
    lib.fs_dump_trajectory.argtypes = [h, C.c_int, C.c_char_p]
    lib.fs_dump_trajectory.restype = C.c_int
    lib.fs_last_kernel.argtypes = [h]
    lib.fs_last_kernel.restype = C.c_char_p
    lib.fs_policy_act_dev.argtypes = [h, C.POINTER(fs_policy), f32p, f32p, f32p]
    lib.fs_policy_act_dev.restype = C.c_int
    lib.fs_policy_rollout_dev.argtypes = [h, C.POINTER(fs_policy), C.c_int, C.c_int, f32p, f32p, f32p, f32p, u8p]
    lib.fs_policy_rollout_dev.restype = C.c_int
    if lib.fs_abi_version() != FS_ABI_VERSION:
        raise FatalFlowError("libflowsim.so ABI %d != binding ABI %d" % (lib.fs_abi_version(), FS_ABI_VERSION))
    _libs[path] = lib
    return lib


def check(rc, lib=None):
    """Map a C return code to the exception type the reference raises at that call point (``lib``: the library the call
    went to, when it is not the stock one)."""
    if rc == FS_OK:
        return
    msg = (lib or load()).fs_last_error().decode("utf-8", "replace")
    if rc == FS_ERR_INVALID:
        raise ValueError(msg)
    if rc == FS_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise FatalFlowError(msg)          # FS_ERR_HIP, FS_ERR_NOSPACE (network/base.py:603-605)
