"""Oracle restatement of Env.step / Env.reset for closed single-lane routes.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, batched over R independent replicas of N vehicles, the sequence the
reference executes in ``flow/envs/base.py:294-412`` (step) and ``:414-560``
(reset) for RingNetwork experiments, with the SUMO side of the step written
out explicitly (S-list in docs/HISTORY.md section 2):

  S1  all controllers read the time-t snapshot        envs/base.py:324-371
  S4  next_vel = max(v + acc*dt, 0)                   vehicle/traci.py:962
  S5  acc is None -> vehicle is not commanded         vehicle/traci.py:960
  S6  slowDown(next_vel, 1e-3) ramp                   vehicle/traci.py:963  (SUMO side, unpinned < 1e-2)
  S9  Euler x' = x + v'*dt  (ballistic optional)      simulation/traci.py:94-96
  S10 headway = gap to the next vehicle along the loop minus its length
                                                      vehicle/traci.py:219-250
  S12 crash <=> some headway < crash_gap after the move (SUMO collision, unpinned)
  S13 reset = placement, no movement, obs, then warm-up steps with no RL action

dtype float64 restates the reference (Python floats); dtype float32 is the
bit-twin of the HIP kernels (same operation order, no contraction).
"""
import numpy as np

from . import controllers as C
from . import rewards as Rw

# controller ids -- must equal include/flowsim.h FS_CTRL_*
CTRL_SIM, CTRL_RL, CTRL_IDM, CTRL_CFM, CTRL_BCM, CTRL_LAC, CTRL_OVM, CTRL_LINEAR_OVM, \
    CTRL_GIPPS, CTRL_FOLLOWER_STOPPER, CTRL_NONLOCAL_FOLLOWER_STOPPER, CTRL_PISATURATION = range(12)
CTRL_USER = 12        # a user-compiled controller (flow_amd.controllers.CompiledController): spec['user_controller_numpy']
FAILSAFE_NONE, FAILSAFE_INSTANTANEOUS, FAILSAFE_SAFE_VELOCITY = range(3)
ENV_ACCEL, ENV_WAVE_ATTENUATION, ENV_WAVE_ATTENUATION_PO, ENV_LANE_CHANGE_ACCEL = range(4)
# multi-agent ring heads (flow/envs/multiagent/ring/*): one observation block per RL vehicle, column = its rl_index
ENV_WAVE_ATTENUATION_PO_MA, ENV_ACCEL_PO_MA = 8, 9
# LaneChangeAccelPOEnv: LaneChangeAccelEnv's actions and reward, per-lane nearest leader / follower of every RL vehicle
ENV_LANE_CHANGE_ACCEL_PO = 10
LC_ENVS = (ENV_LANE_CHANGE_ACCEL, ENV_LANE_CHANGE_ACCEL_PO)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox-4x32-10 (Salmon et al., SC'11), vectorised; all args uint32 arrays."""
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    k0 = np.asarray(k0, dtype=np.uint32)
    k1 = np.asarray(k1, dtype=np.uint32)
    for _ in range(10):
        p0 = M0 * c0.astype(np.uint64)
        p1 = M1 * c2.astype(np.uint64)
        hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
        hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        with np.errstate(over="ignore"):
            k0 = k0 + W0
            k1 = k1 + W1
    return c0, c1, c2, c3


def exact_ln_f32(u):
    """ln(u) for float32 u in (0, 1] as a FIXED sequence of float32 operations (every +, -, *, / is one IEEE rounding; the
    kernels run the same sequence under -ffp-contract=off: flowsim_kernels.h bm_ln_exact).  u = m 2^e with m in
    [sqrt(1/2), sqrt(2)): ln u = e ln 2 + 2 atanh(s), s = (m - 1) / (m + 1), atanh by its odd series up to s^9."""
    f = np.float32
    u = np.asarray(u, dtype=np.float32)
    bits = u.view(np.uint32)
    e = (bits >> np.uint32(23)).astype(np.int32) - np.int32(127)
    m = ((bits & np.uint32(0x7FFFFF)) | np.uint32(0x3F800000)).view(np.float32)
    big = m > f(1.4142135)
    m = np.where(big, m * f(0.5), m)
    e = e + big.astype(np.int32)
    t = m - f(1.0)
    s = t / (f(2.0) + t)
    z = s * s
    p = z * f(0.11111111) + f(0.14285715)
    p = p * z + f(0.2)
    p = p * z + f(0.33333334)
    p = p * z + f(1.0)
    return e.astype(np.float32) * f(0.6931472) + (f(2.0) * s) * p


def exact_cos_turns_f32(t):
    """cos(2 pi t) for float32 t in (-0.25, 1) as a fixed sequence of float32 operations (bm_cos_exact): quarter turns
    a = 4 t, q = floor(a + 1/2), angle (a - q) pi / 2 in [-pi/4, pi/4], even / odd polynomials, selected by q mod 4."""
    f = np.float32
    t = np.asarray(t, dtype=np.float32)
    a = t * f(4.0)
    q = np.floor(a + f(0.5))
    th = (a - q) * f(1.5707964)
    z = th * th
    c = z * f(2.4801587e-05) + f(-1.3888889e-03)
    c = c * z + f(4.1666668e-02)
    c = c * z + f(-0.5)
    c = c * z + f(1.0)
    sn = z * f(2.7557319e-06) + f(-1.9841270e-04)
    sn = sn * z + f(8.3333338e-03)
    sn = sn * z + f(-1.6666667e-01)
    sn = sn * z + f(1.0)
    sn = sn * th
    qi = q.astype(np.int32) & np.int32(3)
    return np.where(qi == 0, c, np.where(qi == 1, -sn, np.where(qi == 2, -c, sn)))


def gaussian_noise(seed, replica, vehicle, step, dtype, exact=False):
    """N(0,1) per (replica, vehicle, step): ONE Philox call, keyed by the 64-bit seed with counter
    (step // 4, vehicle, replica, 0), serves four steps: words (c0, c1) and (c2, c3) feed two Box-Muller
    transforms, each used at the angle 2 pi u2 and a quarter turn back (the sine branch written as a cosine) --
    draw j = step % 4 is r_{j // 2} * cos(2 pi (u2_{j // 2} - (j & 1) / 4)).  Shared definition with the kernels
    (flowsim_kernels.h gauss4 / gauss)."""
    seed = int(seed)
    k0 = np.uint32(seed & 0xFFFFFFFF)
    k1 = np.uint32((seed >> 32) & 0xFFFFFFFF)
    step = np.asarray(step, dtype=np.uint32)
    r0, r1, r2, r3 = philox4x32_10(step >> np.uint32(2), vehicle, replica, np.zeros_like(replica), k0, k1)
    j = step & np.uint32(3)
    first = j < 2
    ra = np.where(first, r0, r2)
    rb = np.where(first, r1, r3)
    # u1 in (0,1], u2 in [0,1)
    u1 = ((ra >> np.uint32(8)).astype(np.float64) + 1.0) * (1.0 / 16777216.0)
    u2 = (rb >> np.uint32(8)).astype(np.float64) * (1.0 / 16777216.0)
    u1 = u1.astype(dtype)
    u2 = u2.astype(dtype)
    u2 = np.where((j & np.uint32(1)) == 1, u2 - np.asarray(0.25, dtype), u2)
    if exact and np.dtype(dtype) == np.float32:               # spec['noise_math'] = 'exact': bit-reproducible on the GPU
        return np.sqrt(np.float32(-2.0) * exact_ln_f32(u1)) * exact_cos_turns_f32(u2)
    two_pi = np.asarray(6.283185307179586, dtype)
    return np.sqrt(np.asarray(-2.0, dtype) * np.log(u1)) * np.cos(two_pi * u2)


def pisaturation_step(hist, n, v_cmd, v, v_lead, h, dt, max_accel, do_update, dtype):
    """PISaturation.get_accel (velocity_controllers.py:208-240) for one slot over R replicas.

    ``hist`` [R,H] ring buffer (slot k holds the k-th appended speed modulo H), ``n`` [R] count of
    appends, ``v_cmd`` [R] previous command.  Returns (accel, new_v_cmd); mutates hist/n where
    ``do_update``.  v_des is the mean of the kept speeds summed OLDEST -> NEWEST in the working
    dtype (the reference's np.mean sums pairwise: equal to rounding)."""
    T = np.dtype(dtype).type
    R, H = hist.shape
    dv = v_lead - v
    dx_s = np.maximum(T(2) * dv, T(4))                                   # :216
    hist_new = hist.copy()
    rows = np.arange(R)
    hist_new[rows, n % H] = v                                            # :219 (append; oldest overwritten = :221-222)
    n_new = n + 1
    cnt = np.minimum(n_new, H)
    start = np.where(n_new > H, n_new % H, 0)
    total = np.zeros(R, dtype=dtype)
    for k in range(H):
        idx = (start + k) % H
        total = np.where(k < cnt, total + hist_new[rows, idx], total)
    v_des = total / cnt.astype(dtype)                                    # :225
    g_l, g_u, gamma, v_catch = T(7), T(30), T(2), T(1)
    v_target = v_des + v_catch * np.minimum(np.maximum((h - g_l) / (g_u - g_l), T(0)), T(1))   # :226-227
    alpha = np.minimum(np.maximum((h - dx_s) / gamma, T(0)), T(1))       # :230
    beta = T(1) - T(0.5) * alpha                                         # :231
    new_cmd = beta * (alpha * v_target + (T(1) - alpha) * v_lead) + (T(1) - beta) * v_cmd    # :234-235
    accel = np.minimum((new_cmd - v) / T(dt), T(max_accel))              # :238-240
    upd = np.asarray(do_update, dtype=bool)
    hist[upd] = hist_new[upd]
    n[upd] = n_new[upd]
    return accel, np.where(upd, new_cmd, v_cmd)

def controller_dispatch(o, v, v_lead, h, has_lead, v_follow, h_follow, rl_value, rl_commanded, on_edge, active,
                        mean_speed, noise_slot=None):
    """BaseController.get_action for every slot (base_controller.py:70-118) plus the RL command
    (envs/base.py:599-615), shared by every oracle class.

    All array arguments are [R,N] except ``active`` [R] and ``mean_speed`` (a callable returning the
    [R] mean speed NonLocalFollowerStopper reads, velocity_controllers.py:127).  ``rl_value`` /
    ``rl_commanded``: the (unclipped) action and whether an action exists for the slot this sub-step;
    ``on_edge`` is False while the vehicle is on an internal edge and junction_mode is on
    (base_controller.py:98-99).  Returns (acc [R,N], commanded [R,N]); advances the LAC / PISaturation
    state of ``o`` for active replicas."""
    T = o.dt_.type
    R, N = v.shape
    acc = np.zeros((R, N), dtype=o.dt_)
    commanded = np.zeros((R, N), dtype=bool)
    ms = None
    for i, vs in enumerate(o.veh):
        ct = vs["controller"]
        p = vs.get("p", [0] * 8)
        sl = (slice(None), i)
        if ct == CTRL_SIM:
            continue
        if ct == CTRL_RL:
            a = rl_value[sl].astype(o.dt_)
            if o.spec.get("clip_actions", True):                  # envs/base.py:584-588
                a = np.clip(a, T(o.spec["action_low"]), T(o.spec["action_high"]))
            acc[sl] = np.where(rl_commanded[sl], a, T(0))
            commanded[sl] = rl_commanded[sl]
            continue
        args = (v[sl], v_lead[sl], h[sl], has_lead[sl])
        cmd = on_edge[sl]
        if ct == CTRL_IDM:
            a = C.idm(*args, v0=p[0], T=p[1], a=p[2], b=p[3], delta=p[4], s0=p[5])
        elif ct == CTRL_CFM:
            a = C.cfm(*args, vs["max_accel"], k_d=p[0], k_v=p[1], k_c=p[2], d_des=p[3], v_des=p[4])
        elif ct == CTRL_BCM:
            a = C.bcm(*args, v_follow[sl], h_follow[sl], vs["max_accel"],
                      k_d=p[0], k_v=p[1], k_c=p[2], d_des=p[3], v_des=p[4])
        elif ct == CTRL_LAC:
            a = C.lac(v[sl], v_lead[sl], h[sl], o.veh_len[i], o.lac_a[sl], o.dt,
                      k_1=p[0], k_2=p[1], h_gap=p[2], tau=p[3])
        elif ct == CTRL_OVM:
            a = C.ovm(*args, vs["max_accel"], alpha=p[0], beta=p[1], h_st=p[2], h_go=p[3], v_max=p[4])
        elif ct == CTRL_LINEAR_OVM:
            a = C.linear_ovm(v[sl], h[sl], v_max=p[0], adaptation=p[1], h_st=p[2])
        elif ct == CTRL_GIPPS:
            a = C.gipps(v[sl], v_lead[sl], h[sl], o.dt, v0=p[0], acc=p[1], b=p[2], b_l=p[3],
                        s0=p[4], tau=p[5])
        elif ct == CTRL_FOLLOWER_STOPPER:
            a = C.follower_stopper(*args, o.dt, v_des=p[0])
        elif ct == CTRL_NONLOCAL_FOLLOWER_STOPPER:
            if ms is None:
                ms = mean_speed()
            a = C.follower_stopper(*args, o.dt, v_des=ms)
        elif ct == CTRL_USER:
            # the user's get_accel restated by the user: a callable (v, v_lead, h, has_lead, v_follow, h_follow, dt,
            # max_accel, p, dtype) -> acceleration [R], every operation in ``dtype`` in the order of the C++ body
            a = o.spec["user_controller_numpy"](v[sl], v_lead[sl], h[sl], has_lead[sl], v_follow[sl], h_follow[sl],
                                                T(o.dt), T(vs["max_accel"]), [T(q) for q in p], o.dt_)
        elif ct == CTRL_PISATURATION:
            a, o.lac_a[sl] = pisaturation_step(o.pis_hist[:, i, :], o.pis_n[:, i], o.lac_a[sl], v[sl],
                                               v_lead[sl], h[sl], o.dt, vs["max_accel"], cmd & active, o.dt_)
        else:
            raise ValueError("unknown controller %r" % ct)
        if ct == CTRL_LAC:                                        # state only advances when get_accel ran
            o.lac_a[sl] = np.where(cmd & active, a, o.lac_a[sl])
        if vs.get("noise", 0) > 0:                                # base_controller.py:109-110
            slot = np.full(R, i, dtype=np.uint32) if noise_slot is None else noise_slot[sl].astype(np.uint32)
            g = gaussian_noise(o.spec.get("seed", 0),
                               replica_ids(o.spec, R), slot,
                               o.step_counter.astype(np.uint32), o.dt_, exact=o.spec.get("noise_math", "hw") == "exact")
            a = a + T(vs["noise"]) * g
        fs = vs.get("fail_safe", FAILSAFE_NONE)
        hl = has_lead[sl]
        nveh = 2 if N > 1 else 1                                  # the fail-safes only ask "is N == 1"
        if fs == FAILSAFE_INSTANTANEOUS:                          # base_controller.py:113-114
            a = np.where(hl, C.failsafe_instantaneous(a, v[sl], h[sl], hl, o.dt, nveh), a)
        elif fs == FAILSAFE_SAFE_VELOCITY:                        # base_controller.py:115-116
            a = np.where(hl, C.failsafe_safe_velocity(a, v[sl], v_lead[sl], h[sl], o.dt, vs.get("delay", 0), nveh), a)
        acc[sl] = a
        commanded[sl] = cmd
    return acc, commanded


def replica_ids(spec, R):
    """Global replica indices of the R rows (they key the Philox streams): a contiguous shard starting at
    `replica_offset` (what fs_config carries), or -- oracle only, to check SAMPLED replicas of a large handle -- the explicit
    list `replica_ids`."""
    ids = spec.get("replica_ids")
    if ids is not None:
        ids = np.asarray(ids, dtype=np.uint32)
        assert ids.shape == (R,)
        return ids
    return (np.arange(R) + int(spec.get("replica_offset", 0))).astype(np.uint32)


class RingOracle:
    """Batched closed-loop (ring) oracle.  ``spec`` is a plain dict:

    num_replicas R, num_vehicles N, sim_step, slowdown_ramp, integrator
    ('euler'|'ballistic'), junction_mode (0|1), junction_length, crash_gap,
    ring_length [R] (sum of the four edges), max_speed, env, target_velocity,
    action_low/high, clip_actions, evaluate, po_max_length, horizon,
    warmup_steps, sims_per_step, seed, init_pos [R,N], init_vel [R,N],
    vehicles: list of N dicts {controller, p[8], fail_safe, noise, delay,
    max_accel, max_decel, length, speed_mode, sumo_tau, sumo_min_gap,
    sumo_max_speed, rl_index}.
    """

    def __init__(self, spec, dtype=np.float64):
        self.spec = spec
        self.dt_ = np.dtype(dtype)
        T = self.dt_.type
        self.R = int(spec["num_replicas"])
        self.N = int(spec["num_vehicles"])
        self.dt = float(spec["sim_step"])
        self.ramp = T(spec.get("slowdown_ramp", self.dt / (self.dt + 1e-3)))
        self.ballistic = spec.get("integrator", "euler") == "ballistic"
        self.junction_mode = int(spec.get("junction_mode", 0))
        self.jlen = float(spec.get("junction_length", 0.1))
        self.crash_gap = T(spec.get("crash_gap", 0.0))
        base = np.broadcast_to(np.asarray(spec["ring_length"], dtype=np.float64), (self.R,))
        self.base_len = base.astype(self.dt_)                        # edges only
        self.L = self.base_len + T(4) * T(self.jlen)                 # network.length(), in T like the kernel
        self.veh = spec["vehicles"]
        self.veh_len = np.array([v.get("length", 5.0) for v in self.veh], dtype=self.dt_)
        # closed loops other than the plain ring (figure eight): an ordered segment table
        # [(phys_start, internal, flow_start, flow_slope)] + the crossing model (docs/HISTORY.md S-J)
        self.segments = spec.get("segments")
        self.junction = spec.get("junction")
        self.rl_slots = [None] * int(spec.get("num_rl", 0))
        for i, v in enumerate(self.veh):
            if v["controller"] == CTRL_RL:
                self.rl_slots[v["rl_index"]] = i
        self.init_pos = np.asarray(spec["init_pos"], dtype=np.float64).astype(self.dt_).reshape(self.R, self.N)
        iv = spec.get("init_vel")
        if iv is None:
            iv = np.broadcast_to(np.array([v.get("initial_speed", 0.0) for v in self.veh]),
                                 (self.R, self.N))
        self.init_vel = np.asarray(iv, dtype=np.float64).astype(self.dt_).reshape(self.R, self.N)
        self.x = self.init_pos.copy()
        self.v = self.init_vel.copy()
        self.prev_v = self.v.copy()
        self.lac_a = np.zeros((self.R, self.N), dtype=self.dt_)     # controller state: LAC a / PISaturation v_cmd
        # PISaturation speed history (velocity_controllers.py:193, 218-222): the last int(38/dt)-1 speeds
        self.pis_H = max(int(38 / self.dt) - 1, 1)
        self.pis_hist = np.zeros((self.R, self.N, self.pis_H), dtype=self.dt_) \
            if any(v["controller"] == CTRL_PISATURATION for v in self.veh) else None
        self.pis_n = np.zeros((self.R, self.N), dtype=np.int64)      # number of speeds ever appended
        self.time_counter = np.zeros(self.R, dtype=np.int64)
        self.step_counter = np.zeros(self.R, dtype=np.int64)   # noise stream position
        self.last_accel = np.zeros((self.R, self.N), dtype=self.dt_)
        self.last_commanded = np.zeros((self.R, self.N), dtype=bool)
        # observation / action order (accel.py:101-123, 150-169; envs/base.py:268-292)
        self.sort_vehicles = bool(spec.get("sort_vehicles", False))
        perm = spec.get("obs_perm")
        self.obs_perm = np.arange(self.N) if perm is None else np.asarray(perm, dtype=np.int64)
        self.x_sort = self.obs_position(self.x).copy()               # AccelEnv.absolute_position

    # ------------------------------------------------------------------ S10
    def headways(self, x=None):
        """h[r,i] = (x_lead - x_i) mod L - len_lead; leader = slot i+1 (cyclic)."""
        x = self.x if x is None else x
        T = self.dt_.type
        if self.N == 1:
            return np.full((self.R, 1), T(1000.0))                   # vehicle/traci.py:237
        d = np.roll(x, -1, axis=1) - x
        d = np.where(d < 0, d + self.L[:, None], d)
        return d - np.roll(self.veh_len, -1)[None, :]

    def in_junction(self, x):
        """True where the front of the vehicle is on an internal edge
        (networks/ring.py:211-214 tables; base_controller.py:98-99)."""
        if self.segments is not None:
            return self._segment_lookup(x)[0]
        base = self.base_len if np.ndim(x) == 1 else self.base_len[:, None]
        quarter = base / self.dt_.type(4)
        q = quarter + self.dt_.type(self.jlen)
        u = x - np.floor(x / q) * q
        return u >= quarter

    def _segment_lookup(self, x):
        """(internal?, table coordinate) of loop coordinate x: the segment is the last one whose start
        is <= x; table coordinate = flow_start + flow_slope * (x - start)  (get_x_by_id,
        vehicle/traci.py:1011-1017 with the network's edge-start table)."""
        T = self.dt_.type
        internal = np.zeros(np.shape(x), dtype=bool)
        start = np.zeros(np.shape(x), dtype=self.dt_)
        fstart = np.zeros(np.shape(x), dtype=self.dt_)
        slope = np.zeros(np.shape(x), dtype=self.dt_)
        for (st, inter, fs, sl) in self.segments:
            hit = x >= T(st)
            internal = np.where(hit, bool(inter), internal)
            start = np.where(hit, T(st), start)
            fstart = np.where(hit, T(fs), fstart)
            slope = np.where(hit, T(sl), slope)
        return internal, fstart + slope * (x - start)

    def obs_position(self, x):
        return x if self.segments is None else self._segment_lookup(x)[1]

    def _yield_speed_cap(self, v):
        """S-J: speed a yielding vehicle may not exceed (SUMO's right-of-way restated): a vehicle on the
        approach of its stream stops at the box entry like behind a standing leader while the other
        stream blocks the box; inf elsewhere.  Returns [R,N]."""
        T = self.dt_.type
        J = self.junction
        x = self.x
        inf = np.full(x.shape, T(3.0e38))
        if J is None:
            return inf
        a_in, a_out, b_in, b_out = T(J["a_in"]), T(J["a_out"]), T(J["b_in"]), T(J["b_out"])
        D, Tg = T(J["lookahead"]), T(J["time_gap"])
        ln = self.veh_len[None, :]
        # the major stream (a) blocks the box while a vehicle is inside it, has not cleared it with its
        # tail, or reaches it within time_gap at its current speed
        major_busy = np.any((x >= a_in - Tg * v) & (x < a_out + ln), axis=1)[:, None]
        minor_in_box = np.any((x >= b_in) & (x < b_out + ln), axis=1)[:, None]
        cap = inf
        for (entry, blocked) in ((b_in, major_busy), (a_in, minor_in_box)):
            approaching = (x >= entry - D) & (x < entry) & blocked
            gap = entry - x
            stop = np.stack([C.sumo_idm_speed(v[:, i], np.zeros(self.R, self.dt_), gap[:, i], np.ones(self.R, bool),
                                              self.dt, accel=vs["max_accel"], decel=vs["max_decel"],
                                              tau=vs.get("sumo_tau", 1.0), min_gap=vs.get("sumo_min_gap", 2.5),
                                              max_speed=vs.get("sumo_max_speed", 30.0))
                             for i, vs in enumerate(self.veh)], axis=1)
            cap = np.where(approaching, np.minimum(cap, stop), cap)
        return cap

    def _crossing_crash(self):
        """S-J: both streams inside the conflict zone of the crossing at once = collision."""
        J = self.junction
        if J is None:
            return np.zeros(self.R, dtype=bool)
        T = self.dt_.type
        x = self.x
        in_a = np.any((x >= T(J["za_lo"])) & (x < T(J["za_hi"])), axis=1)
        in_b = np.any((x >= T(J["zb_lo"])) & (x < T(J["zb_hi"])), axis=1)
        return in_a & in_b

    # ------------------------------------------------------------------ reset
    def reset(self, mask=None):
        m = np.ones(self.R, dtype=bool) if mask is None else np.asarray(mask, dtype=bool)
        self.x[m] = self.init_pos[m]
        self.v[m] = self.init_vel[m]
        self.prev_v[m] = self.init_vel[m]
        self.lac_a[m] = 0
        self.pis_n[m] = 0
        self.time_counter[m] = 0
        self.x_sort[m] = self.obs_position(self.x)[m]
        obs = self.get_state()
        for _ in range(int(self.spec.get("warmup_steps", 0))):       # envs/base.py:554-555
            obs, _, _ = self.step(None, _mask=m)
        return obs

    # ------------------------------------------------------------------ step
    def _order_rank(self, rl_only=False):
        """[R,N] place of every vehicle in sorted(get_ids(), key=absolute_position) -- stable, so ties keep the
        id order (obs_perm); with ``rl_only`` the place among the RL vehicles (accel.py:103-107)."""
        key, perm = self.x_sort, self.obs_perm
        before = (key[:, None, :] < key[:, :, None]) | \
            ((key[:, None, :] == key[:, :, None]) & (perm[None, None, :] < perm[None, :, None]))
        if rl_only:
            is_rl = np.array([v["controller"] == CTRL_RL for v in self.veh])
            before = before & is_rl[None, None, :]
        return before.sum(axis=2)

    def _rl_inputs(self, actions, per_rl=1):
        """(rl_value [R,N], rl_commanded [R,N]): the slot -> action-column mapping (static rl_index, or the place
        among the RL vehicles in sorted order when sort_vehicles)."""
        rl_value = np.zeros((self.R, self.N), dtype=self.dt_)
        rl_cmd = np.zeros((self.R, self.N), dtype=bool)
        if actions is not None:
            acts = np.asarray(actions, dtype=self.dt_)
            col = self._order_rank(rl_only=True) if self.sort_vehicles else None
            for i, vs in enumerate(self.veh):
                if vs["controller"] == CTRL_RL:
                    if col is None:
                        rl_value[:, i] = acts[:, per_rl * vs["rl_index"]]
                    else:
                        rl_value[:, i] = np.take_along_axis(acts, per_rl * col[:, i:i + 1], 1)[:, 0]
                    rl_cmd[:, i] = True
        return rl_value, rl_cmd

    def _on_edge(self):
        if not self.junction_mode:
            return np.ones((self.R, self.N), dtype=bool)
        return ~self.in_junction(self.x)

    def _accelerations(self, actions, active):
        """Returns (acc [R,N], commanded [R,N], h, v_lead, has_lead)."""
        T = self.dt_.type
        R, N = self.R, self.N
        v = self.v
        has_lead = np.full((R, N), N > 1)
        h = self.headways()
        v_lead = np.roll(v, -1, axis=1) if N > 1 else v
        v_follow = np.roll(v, 1, axis=1)
        h_follow = np.roll(h, 1, axis=1)
        rl_value, rl_cmd = self._rl_inputs(actions)
        acc, commanded = controller_dispatch(self, v, v_lead, h, has_lead, v_follow, h_follow, rl_value, rl_cmd,
                                             self._on_edge(), active, lambda: Rw.tree_sum(v) / T(N))
        return acc, commanded, h, v_lead, has_lead

    def _substep(self, actions, active):
        T = self.dt_.type
        dt = T(self.dt)
        acc, commanded, h, v_lead, has_lead = self._accelerations(actions, active)
        if self.sort_vehicles:                                       # accel.py:150-169: additional_command
            self.x_sort = np.where(active[:, None], self.obs_position(self.x), self.x_sort)
        v = self.v
        next_vel = np.maximum(v + acc * dt, T(0))                    # vehicle/traci.py:962
        v_cmd = v + (next_vel - v) * self.ramp                       # S6
        v_new = v.copy()
        for i, vs in enumerate(self.veh):
            sl = (slice(None), i)
            v_sumo = C.sumo_idm_speed(v[sl], v_lead[sl], h[sl], has_lead[sl], self.dt,
                                      accel=vs["max_accel"], decel=vs["max_decel"],
                                      tau=vs.get("sumo_tau", 1.0), min_gap=vs.get("sumo_min_gap", 2.5),
                                      max_speed=vs.get("sumo_max_speed", 30.0))
            vc = v_cmd[sl]
            mode = int(vs.get("speed_mode", 0))
            if mode & 1:                                             # S8 bit0: regard safe speed
                vc = np.minimum(vc, v_sumo)
            if mode & 2:                                             # bit1: regard max accel
                vc = np.minimum(vc, v[sl] + T(vs["max_accel"]) * dt)
            if mode & 4:                                             # bit2: regard max decel
                vc = np.maximum(vc, v[sl] - T(vs["max_decel"]) * dt)
            v_new[sl] = np.where(commanded[sl], vc, v_sumo)
        if self.junction is not None:                                # S-J right of way at the crossing
            cap = self._yield_speed_cap(v)
            for i, vs in enumerate(self.veh):
                obeys = (int(vs.get("speed_mode", 0)) & 1) != 0
                sl = (slice(None), i)
                v_new[sl] = np.where(obeys | ~commanded[sl], np.minimum(v_new[sl], cap[sl]), v_new[sl])
        if self.ballistic:
            x_new = self.x + (v + v_new) / T(2) * dt
        else:
            x_new = self.x + v_new * dt                              # S9
        x_new = np.where(x_new >= self.L[:, None], x_new - self.L[:, None], x_new)
        a2 = active[:, None]
        self.prev_v = np.where(a2, v, self.prev_v)
        self.x = np.where(a2, x_new, self.x)
        self.v = np.where(a2, v_new, self.v)
        self.last_accel = np.where(a2, acc, self.last_accel)
        self.last_commanded = np.where(a2, commanded, self.last_commanded)
        self.time_counter = self.time_counter + active
        self.step_counter = self.step_counter + active
        h_new = self.headways()
        crash = np.any(h_new < self.crash_gap, axis=1) if self.N > 1 else np.zeros(self.R, bool)
        crash = crash | self._crossing_crash()
        return crash & active

    def step(self, actions=None, _mask=None):
        """Env.step for every replica (or those in ``_mask``): returns
        (obs [R,obs_dim], reward [R], done [R])."""
        active = np.ones(self.R, dtype=bool) if _mask is None else _mask.copy()
        crashed = np.zeros(self.R, dtype=bool)
        for _ in range(int(self.spec.get("sims_per_step", 1))):      # envs/base.py:324
            c = self._substep(actions, active & ~crashed)
            crashed |= c                                             # :381-382 break
        obs = self.get_state()
        horizon = self.spec.get("horizon", float("inf"))
        limit = self.spec.get("sims_per_step", 1) * (self.spec.get("warmup_steps", 0) + horizon)
        if self.spec.get("env", ENV_ACCEL) in (ENV_WAVE_ATTENUATION_PO_MA, ENV_ACCEL_PO_MA):
            crashed = np.zeros(self.R, dtype=bool)                   # multiagent/base.py:188-190: crash = 0
        done = (self.time_counter >= limit) | crashed                # :398-400
        reward = self.compute_reward(actions, crashed)
        return obs, reward, done

    # ------------------------------------------------------------------ env heads
    def get_state(self):
        T = self.dt_.type
        env = self.spec.get("env", ENV_ACCEL)
        if env in (ENV_ACCEL, ENV_WAVE_ATTENUATION):
            # accel.py:116-123 / wave_attenuation.py:141-148
            speed = self.v / T(self.spec["max_speed"])
            pos = self.obs_position(self.x) / self.L[:, None]
            if self.sort_vehicles and env == ENV_ACCEL:
                place = self._order_rank()
            else:
                place = np.broadcast_to(self.obs_perm[None, :], speed.shape)
            out_s, out_p = np.empty_like(speed), np.empty_like(pos)
            np.put_along_axis(out_s, place, speed, 1)
            np.put_along_axis(out_p, place, pos, 1)
            return np.concatenate([out_s, out_p], axis=1)
        if env == ENV_WAVE_ATTENUATION_PO:                           # wave_attenuation.py:248-269
            i = self.rl_slots[0]
            j = (i + 1) % self.N                                     # get_leader(rl_id) or rl_id
            max_speed = T(15.)
            max_length = T(self.spec["po_max_length"])
            d = self.x[:, j] - self.x[:, i]
            d = np.where(d < 0, d + self.L, d)                       # % network.length()
            return np.stack([self.v[:, i] / max_speed,
                             (self.v[:, j] - self.v[:, i]) / max_speed,
                             d / max_length], axis=1)
        if env == ENV_WAVE_ATTENUATION_PO_MA:                        # multiagent/ring/wave_attenuation.py:188-208
            out = np.zeros((self.R, 3 * len(self.rl_slots)), self.dt_)
            h = self.headways()
            for i in self.rl_slots:
                c = int(self.spec["vehicles"][i]["rl_index"])
                j = (i + 1) % self.N                                 # get_leader(rl_id) or rl_id
                out[:, 3 * c] = self.v[:, i] / T(15.)
                out[:, 3 * c + 1] = (self.v[:, j] - self.v[:, i]) / T(15.)
                out[:, 3 * c + 2] = h[:, i] / T(self.spec["po_max_length"])      # get_headway: bumper to bumper
            return out
        if env == ENV_ACCEL_PO_MA:                                   # multiagent/ring/accel.py:163-208
            out = np.zeros((self.R, 6 * len(self.rl_slots)), self.dt_)
            h = self.headways()
            xo = self.obs_position(self.x)
            ms, Lr = T(self.spec["max_speed"]), self.L
            lens = self.veh_len
            for i in self.rl_slots:
                c = int(self.spec["vehicles"][i]["rl_index"])
                j, f = (i + 1) % self.N, (i - 1) % self.N
                if self.N > 1:
                    lead_speed, follow_speed = self.v[:, j], self.v[:, f]
                    lead_head = xo[:, j] - xo[:, i] - lens[i]         # (:186-188: no wrap-around, the ego's length)
                    follow_head = h[:, f]                            # get_headway(follower)
                else:
                    lead_speed, follow_speed = np.full(self.R, ms), np.zeros(self.R, self.dt_)
                    lead_head = follow_head = Lr
                out[:, 6 * c + 0] = xo[:, i] / Lr
                out[:, 6 * c + 1] = self.v[:, i] / ms
                out[:, 6 * c + 2] = (lead_speed - self.v[:, i]) / ms
                out[:, 6 * c + 3] = lead_head / Lr
                out[:, 6 * c + 4] = (self.v[:, i] - follow_speed) / ms
                out[:, 6 * c + 5] = follow_head / Lr
            return out
        raise ValueError(env)

    def compute_reward(self, actions, fail):
        env = self.spec.get("env", ENV_ACCEL)
        T = self.dt_.type
        if env == ENV_ACCEL_PO_MA:                                   # multiagent/ring/accel.py:157-161 (fail = crash = 0)
            return Rw.desired_velocity(self.v, self.spec["target_velocity"], np.zeros(self.R, bool))
        if env == ENV_ACCEL:                                         # accel.py:109-114
            if self.spec.get("evaluate", False):
                return Rw.tree_sum(self.v) / T(self.N)
            return Rw.desired_velocity(self.v, self.spec["target_velocity"], fail)
        a = actions
        if a is not None and self.spec.get("clip_actions", True):    # envs/base.py:406-408
            a = np.clip(np.asarray(a, dtype=self.dt_), T(self.spec["action_low"]),
                        T(self.spec["action_high"]))
        return Rw.wave_attenuation_reward(self.v, a, fail)


class MultiLaneRingOracle(RingOracle):
    """Multi-lane ring (RingNetwork with lanes > 1, LaneChangeAccelEnv).

    Restates, on top of RingOracle:

      ML2  own-lane leader/follower search         flow/core/kernel/vehicle/traci.py:219-250 (SUMO getLeader)
      ML3  RL lane-change command                  flow/envs/ring/lane_change_accel.py:132-154,
                                                   flow/core/kernel/vehicle/traci.py:965-997
      ML4  last_lc bookkeeping                     flow/core/kernel/vehicle/traci.py:205-209
      ML6  LaneChangeAccelEnv state / reward       flow/envs/ring/lane_change_accel.py:100-130

    Extra spec keys: num_lanes, init_lane [R,N], lane_change_duration (steps are compared with
    time_counter exactly like the reference does), lane_change_mode (0: every commanded change is
    executed; otherwise it is refused when it would overlap a vehicle of the target lane -- SUMO's
    'no_lat_collide' check restated as a plain overlap test, UNPINNED), last_lc_quirk (True:
    get_last_lc returns the headway as in this fork, vehicle/traci.py:604-614).
    Actions for ENV_LANE_CHANGE_ACCEL are [acc_0, dir_0, acc_1, dir_1, ...] (lane_change_accel.py:134-135).
    """

    def __init__(self, spec, dtype=np.float64):
        super().__init__(spec, dtype)
        self.lanes = int(spec.get("num_lanes", 1))
        il = spec.get("init_lane")
        if il is None:
            il = np.zeros((self.R, self.N), dtype=np.int32)
        self.init_lane = np.asarray(il, dtype=np.int32).reshape(self.R, self.N)
        self.lane = self.init_lane.copy()
        self.last_lc = np.full((self.R, self.N), -(2 ** 30), dtype=np.int64)
        self.lc_duration = self.dt_.type(spec.get("lane_change_duration", 0))
        self.lc_mode = int(spec.get("lane_change_mode", 512))
        self.quirk = bool(spec.get("last_lc_quirk", True))
        # ML7: autonomous lane changing (the simplified model M11 of oracle/opennet.py on a ring) for the non-RL
        # vehicles whose lane_change_mode has a strategic / cooperative / speed-gain / keep-right bit
        self.lc_auto = np.array([(int(v.get("lane_change_mode", 0)) & 0x55) != 0 and v["controller"] != CTRL_RL
                                 for v in self.veh])
        self.lc_enabled = bool(self.lc_auto.any()) and self.lanes > 1
        self.lc_cooldown = max(1, int(spec.get("lane_change_cooldown_steps", 10)))
        self.lc_min_gain = self.dt_.type(spec.get("lane_change_min_gain", 10.0))
        self.num_lane_changes = np.zeros(self.R, dtype=np.int64)

    # ---- ML2
    def neighbours(self, x=None, lane=None):
        """Own-lane leader / follower slot (-1 if none), headway, arc distances matrix d[r,i,j]."""
        x = self.x if x is None else x
        lane = self.lane if lane is None else lane
        T = self.dt_.type
        R, N = self.R, self.N
        d = x[:, None, :] - x[:, :, None]                       # d[r,i,j] = x_j - x_i
        jj = np.arange(N)
        wrap = (d < 0) | ((d == 0) & (jj[None, None, :] < jj[None, :, None]))
        d = np.where(wrap, d + self.L[:, None, None], d)
        same = (lane[:, None, :] == lane[:, :, None]) & (jj[None, None, :] != jj[None, :, None])
        big = T(3.0e38)
        dl = np.where(same, d, big)
        lead = np.argmin(dl, axis=2)
        has = np.take_along_axis(dl, lead[:, :, None], 2)[:, :, 0] < big
        dlead = np.take_along_axis(d, lead[:, :, None], 2)[:, :, 0]
        h = np.where(has, dlead - self.veh_len[lead], T(1000.0))
        # follower: the vehicle whose leader-distance to me is smallest = min over j of d[r,j,i]
        dt_ = np.where(same, np.transpose(d, (0, 2, 1)), big)   # dt_[r,i,j] = d[r,j,i]
        foll = np.argmin(dt_, axis=2)
        lead = np.where(has, lead, -1)
        foll = np.where(has, foll, -1)
        return lead, foll, has, h, d

    def headways(self, x=None):
        return self.neighbours(x)[3]

    def reset(self, mask=None):
        m = np.ones(self.R, dtype=bool) if mask is None else np.asarray(mask, dtype=bool)
        self.lane[m] = self.init_lane[m]
        self.last_lc[m] = -(2 ** 30)
        return super().reset(mask)

    def _accelerations(self, actions, active):
        """As RingOracle._accelerations with the dynamic own-lane neighbours."""
        T = self.dt_.type
        v = self.v
        lead, foll, has_lead, h, _ = self.neighbours()
        li = np.where(lead >= 0, lead, 0)
        fi = np.where(foll >= 0, foll, 0)
        # get_speed(None) is the error value -1001 (vehicle/traci.py get_speed default)
        v_lead = np.where(lead >= 0, np.take_along_axis(v, li, 1), T(-1001))
        v_follow = np.take_along_axis(v, fi, 1)
        h_follow = np.take_along_axis(h, fi, 1)
        per_rl = 2 if self.spec.get("env") in LC_ENVS else 1
        rl_value, rl_cmd = self._rl_inputs(actions, per_rl)
        acc, commanded = controller_dispatch(self, v, v_lead, h, has_lead, v_follow, h_follow, rl_value, rl_cmd,
                                             self._on_edge(), active, lambda: Rw.tree_sum(v) / T(self.N))
        return acc, commanded, h, v_lead, has_lead

    # ---- ML3: the lane the RL vehicles are in after this sub-step's commands
    def _lane_changes(self, actions, active, h):
        if actions is None or self.spec.get("env") not in LC_ENVS:
            return self.lane
        T = self.dt_.type
        acts = np.asarray(actions, dtype=self.dt_)
        new_lane = self.lane.copy()
        _, _, _, _, d = self.neighbours()
        col = self._order_rank(rl_only=True) if self.sort_vehicles else None    # lane_change_accel.py:137-139
        for i, vs in enumerate(self.veh):
            if vs["controller"] != CTRL_RL:
                continue
            dirv = acts[:, 2 * vs["rl_index"] + 1] if col is None else \
                np.take_along_axis(acts, 2 * col[:, i:i + 1] + 1, 1)[:, 0]
            direction = np.where(dirv > T(0.5), 1, np.where(dirv < T(-0.5), -1, 0))
            last = h[:, i] if self.quirk else self.last_lc[:, i].astype(self.dt_)
            blocked = (self.time_counter + 1).astype(self.dt_) <= self.lc_duration + last   # lane_change_accel.py:143-147
            direction = np.where(blocked, 0, direction)
            target = np.clip(self.lane[:, i] + direction, 0, self.lanes - 1)               # traci.py:982-984
            want = (target != self.lane[:, i]) & active
            if self.lc_mode != 0:
                in_t = (self.lane == target[:, None])
                in_t[:, i] = False
                dik = d[:, i, :]                                     # arc from me to k
                dki = np.where(dik == 0, T(0), self.L[:, None] - dik)
                clash = in_t & ((dik < self.veh_len[None, :]) | (dki < self.veh_len[i]))
                want = want & ~clash.any(axis=1)
            new_lane[:, i] = np.where(want, target, self.lane[:, i])
        return new_lane

    def _auto_lane_changes(self, new_lane, active, h):
        """ML7 on the snapshot: every candidate's wish (adjacent lane whose leader gap beats the own headway by
        lc_min_gain with SUMO-IDM desired gaps to the new leader / follower, left wins a tie), then ONE change per
        replica: largest gain, lowest slot."""
        T = self.dt_.type
        R, N = self.R, self.N
        BIG = T(3.0e38)
        _, _, _, _, d = self.neighbours()
        jj = np.arange(N)
        other = jj[None, None, :] != jj[None, :, None]
        v = self.v
        p = {k: np.array([vv.get(k, dflt) for vv in self.veh], dtype=self.dt_)
             for k, dflt in (("sumo_min_gap", 2.5), ("sumo_tau", 1.0), ("max_accel", 2.6), ("max_decel", 4.5))}
        two_sqrt = T(2) * np.sqrt(p["max_accel"] * p["max_decel"])

        def need(a, b):
            return p["sumo_min_gap"][None, :] + np.maximum(T(0), a * p["sumo_tau"][None, :] + a * (a - b) / two_sqrt[None, :])
        ok0 = self.lc_auto[None, :] & ((self.time_counter[:, None] - self.last_lc) >= self.lc_cooldown)
        best_gain = np.full((R, N), -BIG)
        best_lane = np.full((R, N), -1, dtype=np.int64)
        dT = np.transpose(d, (0, 2, 1))                              # dT[r,i,j] = arc from j to i
        for dl in (-1, 1):                                           # right first, so that left wins a tie
            tl = self.lane + dl
            valid = ok0 & (tl >= 0) & (tl < self.lanes)
            in_t = other & (self.lane[:, None, :] == tl[:, :, None])
            dl_ = np.where(in_t, d, BIG)
            li = np.argmin(dl_, axis=2)
            has_l = np.take_along_axis(dl_, li[:, :, None], 2)[:, :, 0] < BIG
            gap_l = np.where(has_l, np.take_along_axis(d, li[:, :, None], 2)[:, :, 0] - self.veh_len[li], T(1000.0))
            v_l = np.where(has_l, np.take_along_axis(v, li, 1), T(0))
            df_ = np.where(in_t, dT, BIG)
            fi = np.argmin(df_, axis=2)
            has_f = np.take_along_axis(df_, fi[:, :, None], 2)[:, :, 0] < BIG
            gap_f = np.where(has_f, np.take_along_axis(dT, fi[:, :, None], 2)[:, :, 0] - self.veh_len[None, :], T(1000.0))
            v_f = np.where(has_f, np.take_along_axis(v, fi, 1), T(0))
            safe = (~has_l | (gap_l >= need(v, v_l))) & (~has_f | (gap_f >= need(v_f, v)))
            gain = gap_l - h
            take = valid & safe & (gain >= self.lc_min_gain) & (gain >= best_gain)
            best_gain = np.where(take, gain, best_gain)
            best_lane = np.where(take, tl, best_lane)
        want = (best_lane >= 0) & active[:, None]
        gsel = np.where(want, best_gain, -BIG)
        win = np.argmax(gsel, axis=1)                                # first maximum = lowest slot
        rows = np.arange(R)
        ok = want[rows, win]
        out = new_lane.copy()
        out[rows[ok], win[ok]] = best_lane[rows[ok], win[ok]]
        self.num_lane_changes = self.num_lane_changes + ok
        return out

    def _substep(self, actions, active):
        T = self.dt_.type
        dt = T(self.dt)
        acc, commanded, h, v_lead, has_lead = self._accelerations(actions, active)
        new_lane = self._lane_changes(actions, active, h)
        if self.lc_enabled:
            new_lane = self._auto_lane_changes(new_lane, active, h)
        if self.sort_vehicles:                                       # accel.py:150-169: additional_command
            self.x_sort = np.where(active[:, None], self.obs_position(self.x), self.x_sort)
        v = self.v
        next_vel = np.maximum(v + acc * dt, T(0))
        v_cmd = v + (next_vel - v) * self.ramp
        v_new = v.copy()
        for i, vs in enumerate(self.veh):
            sl = (slice(None), i)
            v_sumo = C.sumo_idm_speed(v[sl], v_lead[sl], h[sl], has_lead[sl], self.dt,
                                      accel=vs["max_accel"], decel=vs["max_decel"],
                                      tau=vs.get("sumo_tau", 1.0), min_gap=vs.get("sumo_min_gap", 2.5),
                                      max_speed=vs.get("sumo_max_speed", 30.0))
            vc = v_cmd[sl]
            mode = int(vs.get("speed_mode", 0))
            if mode & 1:
                vc = np.minimum(vc, v_sumo)
            if mode & 2:
                vc = np.minimum(vc, v[sl] + T(vs["max_accel"]) * dt)
            if mode & 4:
                vc = np.maximum(vc, v[sl] - T(vs["max_decel"]) * dt)
            v_new[sl] = np.where(commanded[sl], vc, v_sumo)
        x_new = self.x + ((v + v_new) / T(2) * dt if self.ballistic else v_new * dt)
        x_new = np.where(x_new >= self.L[:, None], x_new - self.L[:, None], x_new)
        a2 = active[:, None]
        changed = (new_lane != self.lane) & a2
        self.prev_v = np.where(a2, v, self.prev_v)
        self.x = np.where(a2, x_new, self.x)
        self.v = np.where(a2, v_new, self.v)
        self.lane = np.where(a2, new_lane, self.lane)
        self.last_accel = np.where(a2, acc, self.last_accel)
        self.time_counter = self.time_counter + active
        self.step_counter = self.step_counter + active
        self.last_lc = np.where(changed, self.time_counter[:, None], self.last_lc)          # traci.py:205-209
        _, _, has_new, h_new, _ = self.neighbours()
        crash = np.any(has_new & (h_new < self.crash_gap), axis=1)
        return crash & active

    def lane_neighbours(self):
        """ML8: for every vehicle i and lane q the nearest leader / follower of that lane as the reference finds them
        (flow/core/kernel/vehicle/traci.py:776-867: per lane the vehicles sorted by position, bisect_left at the own
        position, then the edges ahead / behind all the way round the loop).  Stated on ring coordinates: the candidates
        of lane q are ALL its vehicles, the vehicle itself included (the walk round the loop ends on its own edge: a
        vehicle alone in its lane is its own leader and follower, one lap away); arc ahead of i to j = x_j - x_i, plus
        L when negative or j == i (a vehicle of another lane at the SAME position is a leader, gap -length: bisect_left);
        arc behind = x_i - x_j, plus L when <= 0.  Leader = smallest arc ahead (first in id order on a tie, as the stable
        sort + bisect_left pick it), follower = smallest arc behind (last in id order on a tie).
        Returns lead, foll [R, N, lanes] (slot or -1), headway, tailway [R, N, lanes] (1000 for an empty lane)."""
        T = self.dt_.type
        N, x = self.N, self.x
        jj = np.arange(N)
        Lr = self.L[:, None, None]
        ahead = x[:, None, :] - x[:, :, None]
        ahead = np.where((ahead < 0) | (jj[None, None, :] == jj[None, :, None]), ahead + Lr, ahead)
        behind = x[:, :, None] - x[:, None, :]
        behind = np.where(behind <= 0, behind + Lr, behind)
        big = T(3.0e38)
        lead, foll, hw, tw = [], [], [], []
        for q in range(self.lanes):
            cand = np.broadcast_to((self.lane == q)[:, None, :], ahead.shape)
            al, bl = np.where(cand, ahead, big), np.where(cand, behind, big)
            lj = np.argmin(al, axis=2)
            fj = N - 1 - np.argmin(bl[:, :, ::-1], axis=2)
            has = cand.any(axis=2)
            a_min = np.take_along_axis(al, lj[:, :, None], 2)[:, :, 0]
            b_min = np.take_along_axis(bl, fj[:, :, None], 2)[:, :, 0]
            lead.append(np.where(has, lj, -1))
            foll.append(np.where(has, fj, -1))
            hw.append(np.where(has, a_min - self.veh_len[lj], T(1000.0)))
            tw.append(np.where(has, b_min - self.veh_len[None, :], T(1000.0)))
        return tuple(np.stack(a, axis=2) for a in (lead, foll, hw, tw))

    def get_state_po(self):
        """lane_change_accel.py:218-262 with the values it ends up holding: gaps in METRES (its normalisation divides a list
        it has already copied from, :236-247), neighbour speeds over max_speed, the own speed in m/s.  EVERY RL vehicle's block
        is filled here (the reference's `return` sits inside its loop over the RL vehicles: the host env applies that)."""
        T = self.dt_.type
        lead, foll, hw, tw = self.lane_neighbours()
        ms = T(self.spec["max_speed"])
        nrl, lanes = len(self.rl_slots), self.lanes
        out = np.zeros((self.R, 4 * lanes * nrl + nrl), dtype=self.dt_)
        for i, vs in enumerate(self.veh):
            if vs["controller"] != CTRL_RL:
                continue
            c = vs["rl_index"]
            vl = np.where(lead[:, i] >= 0, np.take_along_axis(self.v, np.maximum(lead[:, i], 0), 1) / ms, T(0))
            vf = np.where(foll[:, i] >= 0, np.take_along_axis(self.v, np.maximum(foll[:, i], 0), 1) / ms, T(0))
            out[:, 4 * lanes * c:4 * lanes * (c + 1)] = np.concatenate((hw[:, i], tw[:, i], vl, vf), axis=1)
            out[:, 4 * lanes * nrl + c] = self.v[:, i]
        return out

    def get_state(self):
        if self.spec.get("env") == ENV_LANE_CHANGE_ACCEL_PO:
            return self.get_state_po()
        if self.spec.get("env") != ENV_LANE_CHANGE_ACCEL:
            return super().get_state()
        T = self.dt_.type                                            # lane_change_accel.py:100-117
        cols = [self.v / T(self.spec["max_speed"]), self.x / self.L[:, None], self.lane.astype(self.dt_) / T(self.lanes)]
        # ... for veh_id in self.sorted_ids (get_ids() order = obs_perm unless sort_vehicles)
        place = self._order_rank() if self.sort_vehicles else np.broadcast_to(self.obs_perm[None, :], cols[0].shape)
        out = [np.empty_like(c) for c in cols]
        for o, c in zip(out, cols):
            np.put_along_axis(o, place, c, 1)
        return np.concatenate(out, axis=1)

    def compute_reward(self, actions, fail):
        if self.spec.get("env") not in LC_ENVS:
            return super().compute_reward(actions, fail)
        T = self.dt_.type                                            # lane_change_accel.py:86-98
        reward = Rw.desired_velocity(self.v, self.spec["target_velocity"], fail)
        h = self.headways()
        for i, vs in enumerate(self.veh):
            if vs["controller"] != CTRL_RL:
                continue
            last = h[:, i] if self.quirk else self.last_lc[:, i].astype(self.dt_)
            reward = np.where(last == self.time_counter.astype(self.dt_), reward - T(0.1), reward)
        return reward
