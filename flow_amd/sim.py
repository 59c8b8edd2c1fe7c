"""FlowSim: the Python face of one libflowsim handle (one process, one GPU).

Thin by design: it packs a plain ``spec`` dict into ``fs_config``, calls the
C ABI (include/flowsim.h) and hands back numpy arrays (host API) or fills
caller-owned torch-ROCm tensors (device API).  No simulation arithmetic lives
here; without the HIP library or a GPU, construction raises.
"""
import ctypes as C

import numpy as np

from . import _lib as L

PRECISIONS = {"f32": L.FS_F32, "float32": L.FS_F32, "f64": L.FS_F64, "float64": L.FS_F64,
              np.float32: L.FS_F32, np.float64: L.FS_F64, "mixed": L.FS_MIXED, "f16s": L.FS_F16S}
INTEGRATORS = {"euler": L.FS_EULER, "ballistic": L.FS_BALLISTIC}

VEHICLE_DEFAULTS = dict(controller=L.FS_CTRL_SIM, fail_safe=L.FS_FAILSAFE_NONE, speed_mode=0, rl_index=-1,
                        noise=0.0, delay=0.0, max_accel=2.6, max_decel=4.5, length=5.0, sumo_tau=1.0,
                        sumo_min_gap=2.5, sumo_max_speed=30.0, initial_speed=0.0)


def _ptr(a):
    """Raw address of a numpy array / torch tensor / None."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return a.data_ptr()          # torch tensor


class FlowSim:
    """One batched simulator instance: R replicas x N vehicles on one GPU.

    ``spec`` keys (see oracle/refsim.py RingOracle for the same dict):
    num_replicas, num_vehicles, num_rl, vehicles (list of dicts), ring_length
    [R], init_pos [R,N], init_vel [R,N] (optional), sim_step, slowdown_ramp,
    integrator, junction_mode, junction_length, crash_gap, max_speed, env,
    target_velocity, action_low, action_high, clip_actions, evaluate,
    po_max_length, horizon, warmup_steps, sims_per_step, seed, track_aux, num_lanes, init_lane [R,N],
    lane_change_duration, lane_change_mode, last_lc_quirk, segments [(start, internal, flow_start,
    flow_slope)], junction {a_in, a_out, b_in, b_out, lookahead, time_gap, za_lo, za_hi, zb_lo, zb_hi}.
    """

    def __init__(self, spec, precision="f32", device=0):
        # a population with a CompiledController steps on a copy of the library that holds its get_accel (FS_CTRL_USER)
        src = spec.get("user_controller_source")
        if src is not None:
            from flow_amd import build as _build
            self.lib = L.load(_build.build_user(src))
        else:
            self.lib = L.load()
        self.spec = spec
        self.R = int(spec["num_replicas"])
        self.N = int(spec["num_vehicles"])
        self.num_rl = int(spec.get("num_rl", 0))
        self.precision = PRECISIONS[precision]
        self.real = np.float32 if self.precision in (L.FS_F32, L.FS_F16S) else np.float64      # type of the state fields
        self.device = int(device)
        self._h = C.c_void_p()

        veh = (L.fs_vehicle_spec * self.N)()
        if len(spec["vehicles"]) != self.N:
            raise ValueError("spec['vehicles'] must have num_vehicles entries")
        for i, vd in enumerate(spec["vehicles"]):
            d = dict(VEHICLE_DEFAULTS)
            d.update(vd)
            v = veh[i]
            for k in ("controller", "fail_safe", "speed_mode", "rl_index"):
                setattr(v, k, int(d[k]))
            v.type = int(d.get("type", 0))
            v.lane_change_mode = int(d.get("lane_change_mode", 0))
            p = list(d.get("p", [])) + [0.0] * L.FS_MAX_CTRL_PARAMS
            for k in range(L.FS_MAX_CTRL_PARAMS):
                v.p[k] = float(p[k])
            for k in ("noise", "delay", "max_accel", "max_decel", "length", "sumo_tau", "sumo_min_gap",
                      "sumo_max_speed", "initial_speed"):
                setattr(v, k, float(d[k]))

        dt = float(spec["sim_step"])
        self.open_net = spec.get("network") in ("merge", "bottleneck")
        self.bottleneck = spec.get("network") == "bottleneck"
        ring_length = None
        if not self.open_net:
            ring_length = np.ascontiguousarray(
                np.broadcast_to(np.asarray(spec["ring_length"], dtype=np.float64), (self.R,)))
        init_pos = np.ascontiguousarray(np.asarray(spec["init_pos"], dtype=np.float64).reshape(self.R, self.N))
        init_vel = spec.get("init_vel")
        if init_vel is not None:
            init_vel = np.ascontiguousarray(np.asarray(init_vel, dtype=np.float64).reshape(self.R, self.N))
        init_lane = spec.get("init_lane")
        if init_lane is not None:
            init_lane = np.ascontiguousarray(np.asarray(init_lane, dtype=np.int32).reshape(self.R, self.N))
        segs = spec.get("segments")
        if self.open_net:                                  # one table per route, rows tagged with the route
            segs = [row + (r,) for r, rt in enumerate(spec["routes"]) for row in map(tuple, rt["segments"])]
        seg_arr = None
        if segs:
            seg_arr = (L.fs_segment * len(segs))()
            for k, row in enumerate(segs):
                st, inter, fs0, slope = row[:4]
                seg_arr[k].start, seg_arr[k].internal = float(st), int(bool(inter))
                seg_arr[k].flow_start, seg_arr[k].flow_slope = float(fs0), float(slope)
                seg_arr[k].route = int(row[4]) if len(row) > 4 else 0
        junc = L.fs_junction()
        J = spec.get("junction")
        if J and self.open_net:
            junc.enabled = int(J.get("enabled", 1))
            junc.lookahead, junc.time_gap = float(J["lookahead"]), float(J["time_gap"])
        elif J:
            junc.enabled = 1
            for k in ("a_in", "a_out", "b_in", "b_out", "lookahead", "time_gap", "za_lo", "za_hi", "zb_lo", "zb_hi"):
                setattr(junc, k, float(J[k]))
        inflow_arr, init_alive, n_inflows = None, None, 0
        if self.open_net:
            fl = spec.get("inflows", [])
            n_inflows = len(fl)
            if n_inflows > L.FS_MAX_INFLOWS:
                raise NotImplementedError("more than %d inflows is not built" % L.FS_MAX_INFLOWS)
            inflow_arr = (L.fs_inflow * max(n_inflows, 1))()
            for k, f in enumerate(fl):
                a = inflow_arr[k]
                a.type, a.route = int(f["type"]), int(f["route"])
                num = f.get("number", -1)
                a.number = -1 if num is None else int(num)
                a.period, a.begin, a.end = float(f.get("period", 0.0)), float(f.get("begin", 1.0)), float(f.get("end", 86400.0))
                a.depart_speed, a.depart_pos = float(f["depart_speed"]), float(f["depart_pos"])
                prob = f.get("probability")
                a.probability = -1.0 if prob is None else float(prob)
            init_alive = np.ascontiguousarray(np.asarray(spec["init_alive"], dtype=np.uint8).reshape(self.R, self.N))
            if init_vel is None:
                init_vel = np.zeros((self.R, self.N))
            if init_lane is None:
                init_lane = np.ascontiguousarray(np.asarray(spec["init_route"], dtype=np.int32).reshape(self.R, self.N))
        def cells(rows):
            if not rows:
                return None
            arr = (L.fs_cell * len(rows))()
            for k, (start, lo, hi, lane, last) in enumerate(rows):
                arr[k].edge_start, arr[k].lo, arr[k].hi = float(start), float(lo), float(hi)
                arr[k].lane, arr[k].last_segment = int(lane), int(bool(last))
            return arr
        obs_cells, act_cells = cells(spec.get("obs_cells")), cells(spec.get("action_cells"))
        obs_perm = spec.get("obs_perm")
        if obs_perm is not None:
            obs_perm = np.ascontiguousarray(np.asarray(obs_perm, dtype=np.int32).reshape(self.N))
        horizon = spec.get("horizon", float("inf"))
        dp = C.POINTER(C.c_double)
        cfg = L.fs_config(
            struct_size=C.sizeof(L.fs_config), abi_version=L.FS_ABI_VERSION, precision=self.precision,
            network=(L.FS_NET_BOTTLENECK if self.bottleneck else L.FS_NET_MERGE) if self.open_net
            else (L.FS_NET_FIGURE_EIGHT if segs else L.FS_NET_RING),
            env=int(spec.get("env", L.FS_ENV_ACCEL)),
            integrator=INTEGRATORS[spec.get("integrator", "euler")],
            num_replicas=self.R, num_vehicles=self.N, num_rl=self.num_rl,
            horizon=-1 if horizon == float("inf") else int(horizon),
            warmup_steps=int(spec.get("warmup_steps", 0)), sims_per_step=int(spec.get("sims_per_step", 1)),
            junction_mode=int(spec.get("junction_mode", 0)), clip_actions=int(bool(spec.get("clip_actions", True))),
            evaluate=int(bool(spec.get("evaluate", False))), device=self.device,
            track_aux=int(bool(spec.get("track_aux", False))), num_lanes=int(spec.get("num_lanes", 1)),
            lane_change_mode=int(spec.get("lane_change_mode", 512)),
            last_lc_quirk=int(bool(spec.get("last_lc_quirk", True))),
            seed=int(spec.get("seed", 0) or 0) & 0xFFFFFFFFFFFFFFFF,
            sim_step=dt, slowdown_ramp=float(spec.get("slowdown_ramp", dt / (dt + 1e-3))),
            junction_length=float(spec.get("junction_length", 0.1)), crash_gap=float(spec.get("crash_gap", 0.0)),
            max_speed=float(spec["max_speed"]), target_velocity=float(spec.get("target_velocity", 0.0)),
            action_low=float(spec.get("action_low", 0.0)), action_high=float(spec.get("action_high", 0.0)),
            po_max_length=float(spec.get("po_max_length", 1.0)),
            lane_change_duration=float(spec.get("lane_change_duration", 0.0)),
            vehicles=veh, ring_length=ring_length.ctypes.data_as(dp) if ring_length is not None else None,
            init_pos=init_pos.ctypes.data_as(dp),
            init_vel=init_vel.ctypes.data_as(dp) if init_vel is not None else None,
            init_lane=init_lane.ctypes.data_as(C.POINTER(C.c_int32)) if init_lane is not None else None,
            segments=seg_arr, num_segments=len(segs) if segs else 0, num_inflows=n_inflows, junction=junc,
            inflows=inflow_arr, init_alive=init_alive.ctypes.data_as(C.POINTER(C.c_uint8)) if init_alive is not None else None,
            merge_x=float(spec.get("merge_x", 0.0)), box_in=float(spec.get("box_in", 0.0)),
            end_x=float(spec.get("end_x", 0.0)), net_length=float(spec.get("net_length", 0.0)),
            ma_apply_actions=int(bool(spec.get("ma_apply_actions", False))),
            num_obs_cells=len(obs_cells) if obs_cells is not None else 0,
            merge1_x=float(spec.get("merge1_x", spec.get("merge_x", 0.0))),
            merge2_x=float(spec.get("merge2_x", spec.get("merge_x", 0.0))),
            zipper_distance=float(spec.get("zipper_distance", 0.0)), speed_limit=float(spec.get("speed_limit", 0.0)),
            outflow_norm=2000.0 * float(spec.get("scaling", 1)), obs_cells=obs_cells, act_cells=act_cells,
            obs_outflow_window=int(spec.get("obs_outflow_window", 20)),
            reward_outflow_window=int(spec.get("reward_outflow_window", 10)),
            track_followers=int(bool(spec.get("track_followers", True))),
            num_paths=int(spec.get("num_paths", 0)) if spec.get("network") == "bottleneck" else 0,
            lane_change_cooldown_steps=int(spec.get("lane_change_cooldown_steps", 10)), reserved6=0,
            lane_change_min_gain=float(spec.get("lane_change_min_gain", 10.0)),
            sort_vehicles=int(bool(spec.get("sort_vehicles", False))), noise_exact=int(spec.get("noise_math", "hw") == "exact"),
            obs_perm=obs_perm.ctypes.data_as(C.POINTER(C.c_int32)) if obs_perm is not None else None,
            replica_offset=int(spec.get("replica_offset", 0)))
        if self.open_net:
            cfg.route_start[0] = float(spec["routes"][0]["start"])
            cfg.route_start[1] = float(spec["routes"][min(1, len(spec["routes"]) - 1)]["start"])
        L.check(self.lib.fs_create(C.byref(cfg), C.byref(self._h)), self.lib)
        self.obs_dim = self.lib.fs_obs_dim(self._h)
        self.act_dim = self.lib.fs_action_dim(self._h)

    # ------------------------------------------------------------------ life cycle
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.fs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        """Enqueue later launches on ``hip_stream`` (int address of a hipStream_t, e.g.
        ``torch.cuda.current_stream().cuda_stream``; 0 is HIP's default stream)."""
        L.check(self.lib.fs_set_stream(self._h, C.c_void_p(int(hip_stream) if hip_stream else None)), self.lib)

    def use_own_stream(self):
        """Go back to the non-blocking stream the handle created for itself."""
        L.check(self.lib.fs_use_own_stream(self._h), self.lib)

    def sync(self):
        L.check(self.lib.fs_sync(self._h), self.lib)

    # ------------------------------------------------------------------ host API (numpy in / out)
    def reset(self, mask=None):
        """Env.reset for the masked replicas (all if None); returns obs float32[R,obs_dim]."""
        obs = np.empty((self.R, self.obs_dim), dtype=np.float32)
        m = None if mask is None else np.ascontiguousarray(np.asarray(mask, dtype=np.uint8))
        if m is not None and m.shape != (self.R,):
            raise ValueError("mask must have shape [R]")
        L.check(self.lib.fs_reset(self._h, _ptr(m), _ptr(obs)), self.lib)
        return obs

    def step(self, actions=None):
        """Env.step for all replicas; returns (obs [R,obs_dim] f32, reward [R] f32, done [R] bool)."""
        a = None
        if actions is not None and self.act_dim > 0:
            a = np.ascontiguousarray(np.asarray(actions, dtype=np.float32).reshape(self.R, self.act_dim))
        obs = np.empty((self.R, self.obs_dim), dtype=np.float32)
        rew = np.empty(self.R, dtype=np.float32)
        done = np.empty(self.R, dtype=np.uint8)
        L.check(self.lib.fs_step(self._h, _ptr(a), _ptr(obs), _ptr(rew), _ptr(done)), self.lib)
        self.last_done_flags = done          # bit 0: horizon reached, bit 1: collision (include/flowsim.h)
        return obs, rew, done.astype(bool)

    # ------------------------------------------------------------------ device API (torch-ROCm tensors)
    def reset_dev(self, obs, mask=None):
        L.check(self.lib.fs_reset_dev(self._h, _ptr(mask), _ptr(obs)), self.lib)

    def step_dev(self, obs, rew, done, actions=None):
        L.check(self.lib.fs_step_dev(self._h, _ptr(actions), _ptr(obs), _ptr(rew), _ptr(done)), self.lib)

    def rollout_dev(self, num_steps, obs, rew, done, actions=None, action_stride_steps=None,
                    obs_every_step=True):
        """K env steps in one launch.  obs/rew/done are [K,R,...] when obs_every_step else [R,...]."""
        if action_stride_steps is None:
            action_stride_steps = self.R * self.act_dim if actions is not None and actions.dim() == 3 else 0
        L.check(self.lib.fs_rollout_dev(self._h, int(num_steps), _ptr(actions), int(action_stride_steps),
                                        _ptr(obs), _ptr(rew), _ptr(done), int(bool(obs_every_step))), self.lib)

    # ------------------------------------------------------------------ policy in the loop (include/flowsim.h fs_policy)
    def policy_act_dev(self, pol, obs, act, logp):
        L.check(self.lib.fs_policy_act_dev(self._h, C.byref(pol), _ptr(obs), _ptr(act), _ptr(logp)), self.lib)

    def policy_rollout_dev(self, pol, num_steps, obs, act, logp, rew, done, reset_done=False):
        L.check(self.lib.fs_policy_rollout_dev(self._h, C.byref(pol), int(num_steps), int(bool(reset_done)), _ptr(obs),
                                               _ptr(act), _ptr(logp), _ptr(rew), _ptr(done)), self.lib)

    # ------------------------------------------------------------------ state access
    def _field_shape(self, field):
        if field in (L.FS_FIELD_TIME,):
            return (self.R,), np.int32
        if field in (L.FS_FIELD_LANE, L.FS_FIELD_LAST_LC, L.FS_FIELD_LEADER, L.FS_FIELD_INIT_LANE, L.FS_FIELD_ROUTE,
                     L.FS_FIELD_SEQ, L.FS_FIELD_ORIGIN, L.FS_FIELD_FOLLOWER, L.FS_FIELD_CTL_SEQ, L.FS_FIELD_ARRIVED_RL):
            return (self.R, self.N), np.int32
        if field == L.FS_FIELD_COUNTERS:
            return (self.R, 8), np.int32
        if field == L.FS_FIELD_MAX_SPEED:
            return (self.R, self.N), self.real
        if field in (L.FS_FIELD_RING_LENGTH, L.FS_FIELD_INIT_RING_LENGTH):
            return (self.R,), self.real
        return (self.R, self.N), self.real

    def get_state(self, field):
        shape, dt = self._field_shape(field)
        out = np.empty(shape, dtype=dt)
        L.check(self.lib.fs_get_state(self._h, int(field), _ptr(out), out.nbytes), self.lib)
        return out

    def set_state(self, field, value):
        shape, dt = self._field_shape(field)
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(value, dtype=dt), shape))
        L.check(self.lib.fs_set_state(self._h, int(field), _ptr(a), a.nbytes), self.lib)

    @property
    def last_kernel(self):
        """Family of the step kernel the last step / rollout launch chose (fs_last_kernel)."""
        return self.lib.fs_last_kernel(self._h).decode()

    def add_vehicle(self, slot, route, x, speed, replica=0):
        """Put the (absent) vehicle of ``slot`` back into the network (fs_add_vehicle: k.vehicle.add of the reference)."""
        L.check(self.lib.fs_add_vehicle(self._h, int(replica), int(slot), int(route), float(x), float(speed)), self.lib)

    def dump_trajectory(self, replica, csv_path):
        """Append the current state of ``replica`` to ``csv_path`` (fs_dump_trajectory)."""
        L.check(self.lib.fs_dump_trajectory(self._h, int(replica), str(csv_path).encode()), self.lib)

    # convenience
    @property
    def pos(self):
        return self.get_state(L.FS_FIELD_POS)

    @property
    def vel(self):
        return self.get_state(L.FS_FIELD_VEL)

    @property
    def headway(self):
        return self.get_state(L.FS_FIELD_HEADWAY)

    @property
    def time_counter(self):
        return self.get_state(L.FS_FIELD_TIME)
