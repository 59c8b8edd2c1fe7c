"""Resolve (env class, EnvParams, SumoParams, Network) into the plain ``spec`` dict that
flow_amd.sim.FlowSim packs into fs_config.  Host-side, runs once per construction."""
import numpy as np

from flow_amd import _lib as L
from flow_amd.controllers import RLController
from flow_amd.networks.figure_eight import FigureEightNetwork
from flow_amd.networks.ring import RingNetwork
from flow_amd.utils.exceptions import FatalFlowError


def handle_seed(sim_params):
    """Philox key of the handle.  ``SumoParams(seed=None)`` means "SUMO --random" in the reference
    (flow/core/kernel/simulation/traci.py:122-127): a seed is drawn here, once per simulator.  Within a handle the
    random streams never replay: the acceleration-noise counter runs on across resets and the entry-lane draws are
    keyed by the replica's episode number (k_reset_open), which is what the reference's re-seeding on
    restart_instance resets (envs/base.py:436-441) amounts to."""
    if sim_params.seed is None:
        import random
        return random.getrandbits(31)
    return int(sim_params.seed)


def vehicle_slots(vehicle_kernel, rl_order, ids=None):
    """One fs_vehicle_spec dict per vehicle, in id (insertion) order (vehicle/traci.py:109-117); ``ids`` restricts
    the table to some of the vehicles."""
    slots = []
    for veh_id in ids or getattr(vehicle_kernel, "_order", None) or vehicle_kernel.get_ids():
        ctrl = vehicle_kernel.get_acc_controller(veh_id)
        if ctrl.FS_ID is None:
            raise NotImplementedError("controller %s is not built" % type(ctrl).__name__)
        cf = ctrl.car_following_params
        d = dict(controller=ctrl.FS_ID, p=[float(x) for x in ctrl.fs_params()],
                 fail_safe={None: 0, 'instantaneous': 1, 'safe_velocity': 2}[ctrl.fail_safe],
                 noise=float(ctrl.accel_noise), delay=float(ctrl.delay), max_accel=float(ctrl.max_accel),
                 max_decel=float(ctrl.max_deaccel), length=float(vehicle_kernel.get_length(veh_id)),
                 speed_mode=int(cf.speed_mode), sumo_tau=float(cf.controller_params["tau"]),
                 sumo_min_gap=float(cf.controller_params["minGap"]),
                 sumo_max_speed=float(cf.controller_params["maxSpeed"]),
                 initial_speed=float(vehicle_kernel.get_initial_speed(veh_id)),
                 lane_change_mode=int(vehicle_kernel.type_parameters[vehicle_kernel.get_type(veh_id)][
                     "lane_change_params"].lane_change_mode),
                 rl_index=rl_order.index(veh_id) if isinstance(ctrl, RLController) else -1,
                 user_source=getattr(ctrl, "source", None))
        slots.append(d)
    return slots


def user_controller_source(slots):
    """The one get_accel body of the population's CompiledControllers (None without any): a handle steps on ONE library."""
    bodies = {str(d["user_source"]).strip() for d in slots if d["controller"] == L.FS_CTRL_USER}
    if len(bodies) > 1:
        raise NotImplementedError("all CompiledControllers of an environment must share one SOURCE (their parameters may "
                                  "differ): a handle steps on one library")
    return bodies.pop() if bodies else None


def initial_positions(network_kernel, initial_config, num_vehicles, num_replicas, rng=None):
    """[R,N] absolute start positions (+ [N] lanes).  The even/random placement of
    network/base.py:221-608 is computed once; a positive ``perturbation`` is drawn per
    replica (base.py:384-389), clamped to the vehicle's edge like the reference."""
    pert = initial_config.perturbation
    cfg = initial_config
    if pert > 0:                                  # draw the perturbation here, vectorised over replicas
        import copy
        cfg = copy.copy(initial_config)
        cfg.perturbation = 0.0
    pos, lanes = network_kernel.generate_starting_positions(cfg, num_vehicles)
    x = np.array([network_kernel.loop_coordinate(e, p) for e, p in pos], dtype=np.float64)
    X = np.tile(x, (num_replicas, 1))
    if pert > 0:
        rng = rng or np.random
        start = np.array([network_kernel.loop_coordinate(e, 0) for e, _ in pos])
        elen = np.array([network_kernel.edge_length(e) for e, _ in pos])
        rel = np.array([p for _, p in pos])
        rel = np.clip(rel[None, :] + rng.normal(0, pert, (num_replicas, num_vehicles)), 0, elen[None, :])
        X = start[None, :] + rel
    return X, np.tile(np.asarray(lanes, dtype=np.int32), (num_replicas, 1))


def check_placement(X, lengths, loop_length, lanes=None):
    """Slot order must be ring order with no overlap (the reference raises 'Not enough vehicles
    have spawned' when SUMO refuses an overlapping insertion, envs/base.py:536-542).  On a
    multi-lane ring only overlaps inside a lane are rejected (the leader is searched per step)."""
    if lanes is not None and lanes.max() > 0:
        Lp = np.broadcast_to(np.asarray(loop_length, dtype=np.float64), (X.shape[0],))[:, None, None]
        d = (X[:, None, :] - X[:, :, None]) % Lp                          # arc i -> j
        same = (lanes[:, None, :] == lanes[:, :, None]) & ~np.eye(X.shape[1], dtype=bool)[None]
        if (same & (d < lengths[None, None, :])).any():
            raise FatalFlowError("initial placement overlaps inside a lane")
    elif X.shape[1] > 1:
        gap = np.roll(X, -1, axis=1) - X
        gap[:, -1] += loop_length if np.ndim(loop_length) == 0 else np.asarray(loop_length)
        gap = gap - np.roll(lengths, -1)[None, :]
        if (gap < 0).any():
            raise FatalFlowError("initial placement overlaps or is out of ring order (perturbation too large?)")
    if (X < 0).any() or (X >= (loop_length if np.ndim(loop_length) == 0 else np.asarray(loop_length)[:, None])).any():
        raise FatalFlowError("initial position outside the network")


def type_slot(veh_k, type_name, type_index, rl_index):
    """fs_vehicle_spec dict of a slot that holds vehicles of ``type_name`` (open networks: slots are
    partitioned by vehicle type, every vehicle of a type has the controller of VehicleParams.add)."""
    tp = veh_k.type_parameters[type_name]
    acc_cls, acc_kw = tp["acceleration_controller"]
    ctrl = acc_cls("slot", car_following_params=tp["car_following_params"], **(acc_kw or {}))
    if ctrl.FS_ID is None:
        raise NotImplementedError("controller %s is not built" % type(ctrl).__name__)
    cf = ctrl.car_following_params
    is_rl = isinstance(ctrl, RLController)
    return dict(controller=ctrl.FS_ID, p=[float(x) for x in ctrl.fs_params()],
                fail_safe={None: 0, 'instantaneous': 1, 'safe_velocity': 2}[ctrl.fail_safe],
                noise=float(ctrl.accel_noise), delay=float(ctrl.delay), max_accel=float(ctrl.max_accel),
                max_decel=float(ctrl.max_deaccel), length=float(tp.get("length", 5.0)),
                speed_mode=int(cf.speed_mode), sumo_tau=float(cf.controller_params["tau"]),
                sumo_min_gap=float(cf.controller_params["minGap"]),
                sumo_max_speed=float(cf.controller_params["maxSpeed"]),
                initial_speed=float(tp.get("initial_speed", 0.0)), type=type_index,
                lane_change_mode=int(tp["lane_change_params"].lane_change_mode),
                rl_index=rl_index if is_rl else -1, user_source=getattr(ctrl, "source", None)), is_rl


def slot_capacities(vehicles, inflows, total, given=None):
    """Slots per vehicle type (M1).  ``given`` ({type: slots}) wins; otherwise every type gets its initial
    vehicles plus a share of the remaining slots proportional to its inflow rate (vehicles / hour)."""
    names = [t["veh_id"] for t in vehicles.initial]
    init = {t["veh_id"]: int(t["num_vehicles"]) for t in vehicles.initial}
    if given:
        caps = [int(given.get(n, init[n])) for n in names]
    else:
        rate = {n: 0.0 for n in names}
        for f in inflows:
            rate[f["vtype"]] += (float(f["vehsPerHour"]) if "vehsPerHour" in f else
                                 3600.0 * float(f["probability"]) if "probability" in f else 3600.0 / float(f["period"]))
        spare = total - sum(init.values())
        if spare < 0:
            raise FatalFlowError("more initial vehicles than vehicle slots (max_vehicles)")
        tot_rate = sum(rate.values())
        caps = [init[n] + (int(spare * rate[n] / tot_rate) if tot_rate > 0 else 0) for n in names]
        left = total - sum(caps)                           # rounding leftovers go to the busiest type
        if tot_rate > 0 and left > 0:
            caps[max(range(len(names)), key=lambda i: rate[names[i]])] += left
    for n, c in zip(names, caps):
        if c < init[n]:
            raise FatalFlowError("slot_capacity[%r] is below its initial vehicles" % n)
    return names, caps


def build_open_spec(env, num_replicas, rng=None):
    """The spec of an open-network env (MergeNetwork): slot pools, route tables, inflow table."""
    network, net_k, veh_k = env.network, env.k.network, env.k.vehicle
    sp, ep = env.sim_params, env.env_params
    ap = network.net_params.additional_params
    lane_drop = network.specify_lane_joins() is not None          # BottleneckNetwork
    if int(ap.get("merge_lanes", 1)) != 1 or int(ap.get("highway_lanes", 1)) != 1:
        raise NotImplementedError("multi-lane merge networks are not built in the HIP step loop yet")
    if lane_drop and int(ap.get("scaling", 1)) not in (1, 2):
        raise NotImplementedError("BottleneckNetwork is built for scaling 1 (4 -> 2 -> 1 lanes) and 2 (8 -> 4 -> 2)")
    tables = net_k.open_tables()
    R = int(num_replicas)
    flows = network.net_params.inflows.get()
    total = int(getattr(sp, "max_vehicles", None) or 64)
    names, caps = slot_capacities(network.vehicles, flows, total, getattr(sp, "slot_capacity", None))
    if env.FS_ENV == L.FS_ENV_MERGE_PO and sum(caps) < int(ep.additional_params["num_rl"]):
        # the observation has num_rl places even if fewer vehicles can ever exist: pad with empty slots
        rl_types = [i for i, n in enumerate(names)
                    if veh_k.type_parameters[n]["acceleration_controller"][0] == RLController]
        caps[rl_types[0] if rl_types else 0] += int(ep.additional_params["num_rl"]) - sum(caps)
    if lane_drop and sum(caps) < total:
        # the bottleneck heads use one lane per lane-segment and need the full wave: spare slots stay empty
        caps[max(range(len(caps)), key=lambda i: caps[i])] += total - sum(caps)
    N = sum(caps)
    n_max = 256 if lane_drop else 64       # FS_MAX_SLOTS_WIDE: k_steps_wide runs a replica on up to four waves
    if N < 1 or N > n_max:
        raise NotImplementedError("this network holds 1..%d vehicle slots per replica (got %d)" % (n_max, N))
    if lane_drop and int(ap.get("scaling", 1)) == 2 and N <= 64:
        raise NotImplementedError("BottleneckNetwork with scaling 2 runs on the workgroup-per-replica kernel: set "
                                  "SumoParams(max_vehicles=...) above 64 (its demand needs that many slots anyway)")
    slots, base, n_rl_slots = [], {}, 0
    for t, (name, cap) in enumerate(zip(names, caps)):
        base[name] = len(slots)
        for _ in range(cap):
            d, is_rl = type_slot(veh_k, name, t, n_rl_slots)
            n_rl_slots += 1 if is_rl else 0
            slots.append(d)
    # initial vehicles: ids in VehicleParams order, the k-th vehicle of a type sits in the k-th slot of its pool
    ids = veh_k.get_ids()
    pos, start_lanes = net_k.generate_starting_positions(network.initial_config, len(ids)) if ids else ([], [])
    alive = np.zeros((R, N), dtype=bool)
    X = np.zeros((R, N))
    V = np.zeros((R, N))
    route = np.zeros((R, N), dtype=np.int32)
    seen, init_slot = {}, {}
    for n_v, (veh_id, (edge, p)) in enumerate(zip(ids, pos)):
        name = veh_k.get_type(veh_id)
        k = seen.get(name, 0)
        seen[name] = k + 1
        i = base[name] + k
        init_slot[veh_id] = i
        # Flow's edge-start table of the bottleneck does not match its edge lengths (bottleneck.py:232-234 vs
        # :116-165), so gen_even_start_pos can return a position past the end of an edge: kept inside the edge
        p = min(p, net_k.edge_length(edge) - 0.01)
        r, x = net_k.open_coordinate(edge, p)
        if lane_drop:       # the path is the entry lane: lane l of an edge after j joins continues entry lane l << j
            joins = sum(1 for m in (tables["merge1_x"], tables["merge2_x"]) if x >= m)
            r = int(start_lanes[n_v]) << joins
        alive[:, i], X[:, i], route[:, i] = True, x, r
        V[:, i] = float(veh_k.get_initial_speed(veh_id))
    pert = network.initial_config.perturbation
    if pert > 0 and ids:                                   # network/base.py:384-389, drawn per replica
        rng = rng or np.random
        for veh_id, (edge, p) in zip(ids, pos):
            i = init_slot[veh_id]
            r, x0 = net_k.open_coordinate(edge, 0.0)
            # base.py:386-389 clamps to [0, edge length]; a vehicle exactly at the end of the last edge would already
            # have arrived, so the upper clamp stays 1 cm inside the edge
            rel = np.clip(p + rng.normal(0, pert, R), 0, net_k.edge_length(edge) - 0.01)
            X[:, i] = x0 + rel
    first_edges = [p[0] for p in network.specify_open_routes()]
    dt_ = sp.sim_step
    inflows = []
    for f in flows:
        if f["edge"] not in first_edges:
            raise NotImplementedError("inflow on edge %r: only the first edge of a route is built" % f["edge"])
        ds = f.get("departSpeed", 0)
        if isinstance(ds, str):
            if ds not in ("speedLimit", "max"):
                raise NotImplementedError("departSpeed=%r is not built" % ds)
            ds = net_k.speed_limit(f["edge"])
        prob = float(f["probability"]) if "probability" in f else None   # per second (params.py:1103-1105)
        period = 0.0 if prob is not None else (3600.0 / float(f["vehsPerHour"]) if "vehsPerHour" in f else float(f["period"]))
        tname = f["vtype"]
        flow_route = first_edges.index(f["edge"])
        if lane_drop:                                      # the "route" of a lane-drop network is the entry lane
            dl = f.get("departLane", "first")
            if dl == "random":
                flow_route = -1
            elif dl == "first":
                flow_route = 0
            elif isinstance(dl, int) and 0 <= dl < tables["num_paths"]:
                flow_route = int(dl)
            else:
                raise NotImplementedError("departLane=%r on a multi-lane edge is not built ('random', 'first' or a "
                                          "lane index are)" % (dl,))
        inflows.append(dict(type=names.index(tname), route=flow_route, period=period, probability=prob,
                            begin=float(f.get("begin", 1)), end=float(f.get("end", 86400)),
                            number=int(f["number"]) if "number" in f else -1, depart_speed=float(ds),
                            depart_pos=float(veh_k.type_parameters[tname].get("length", 5.0)), name=f["name"]))
    dt = sp.sim_step
    ramp = getattr(sp, "slowdown_ramp", None)
    space = env.action_space
    num_rl = int(ep.additional_params["num_rl"]) if env.FS_ENV == L.FS_ENV_MERGE_PO else n_rl_slots
    if env.FS_ENV == L.FS_ENV_MERGE_PO and num_rl > N:
        raise FatalFlowError("num_rl exceeds the vehicle slots of a replica")
    extra = {}
    if lane_drop:
        obs_cells, act_cells = env._fs_cells(tables)
        num_rl = len(act_cells)
        if getattr(env, "PER_VEHICLE_ACTIONS", False):     # BottleneckAccelEnv: one acceleration column per RL slot
            num_rl = n_rl_slots
            if num_rl > 64:
                raise NotImplementedError("at most 64 RL vehicle slots per replica")
        extra = dict(obs_cells=obs_cells, action_cells=act_cells, scaling=int(ap.get("scaling", 1)),
                     track_followers=False,                          # no bottleneck env reads get_follower
                     # M11 (simplified lane changing for types whose lane_change_mode lets SUMO change lanes)
                     lane_change_cooldown_steps=max(1, int(round(float(getattr(sp, "lane_change_cooldown", 5.0)) / dt_))),
                     lane_change_min_gain=float(getattr(sp, "lane_change_min_gain", 10.0)),

                     zipper_distance=float(getattr(sp, "zipper_distance", 50.0)),
                     obs_outflow_window=max(1, min(20, int(20 * dt_ / dt_))),        # get_outflow_rate(20 * sim_step)
                     reward_outflow_window=max(1, min(20, int(10 * dt_ / dt_))))     # get_outflow_rate(10 * sim_step)
    jm = getattr(sp, "junction_mode", None)
    tg = getattr(sp, "crossing_time_gap", None)
    merge_len = float(ap.get("merge_length", 0.0))
    spec = dict(
        network="bottleneck" if lane_drop else "merge", num_replicas=R, num_vehicles=N, num_rl=num_rl, vehicles=slots,
        init_alive=alive, init_pos=X, init_vel=V, init_route=route, inflows=inflows,
        junction=dict(enabled=int(getattr(sp, "merge_right_of_way", True)) if not lane_drop else 0,
                      lookahead=merge_len, time_gap=1.0 if tg is None else float(tg)),
        speed_limit=float(net_k.speed_limit(first_edges[0])),
        sim_step=dt, slowdown_ramp=dt / (dt + 1e-3) if ramp is None else float(ramp),
        integrator="ballistic" if getattr(sp, "use_ballistic", False) else "euler",
        junction_mode=int(1 if jm is None else jm), junction_length=float(net_k.junction_length),
        crash_gap=float(getattr(sp, "crash_gap", 0.0)), max_speed=float(net_k.max_speed()),
        env=env.FS_ENV, target_velocity=float(ep.additional_params.get("target_velocity", 0.0)),
        action_low=float(np.min(space.low)) if space.low.size else 0.0,          # an env without RL vehicles has an
        action_high=float(np.max(space.high)) if space.high.size else 0.0,      # empty action space
        # MultiEnv.clip_actions returns the dict unclipped in this fork (multiagent/base.py:366-391)
        clip_actions=bool(ep.clip_actions) and env.FS_ENV != L.FS_ENV_MERGE_MA,
        evaluate=bool(ep.evaluate) and not lane_drop,      # (BottleneckDesiredVelocityEnv's evaluate reward is the host's)
        horizon=ep.horizon, warmup_steps=int(ep.warmup_steps), sims_per_step=int(ep.sims_per_step),
        seed=handle_seed(sp), noise_math=getattr(sp, "noise_math", "hw"), replica_offset=int(getattr(env, "_replica_offset", 0)), track_aux=bool(getattr(env, "_track_aux", True)),
        ma_apply_actions=not bool(getattr(env, "APPLY_ENUMERATE_QUIRK", True)),
        slot_types=names, slot_base=base, slot_caps=dict(zip(names, caps)), init_slot=init_slot,
        user_controller_source=user_controller_source(slots), **tables)
    spec.update(extra)
    return spec


def build_spec(env, num_replicas, rng=None):
    """The spec of ``env`` (a flow_amd Env under construction) replicated ``num_replicas`` times."""
    network, net_k, veh_k = env.network, env.k.network, env.k.vehicle
    sp, ep = env.sim_params, env.env_params
    if network.specify_open_routes() is not None:
        lane_drop = network.specify_lane_joins() is not None
        ok = (L.FS_ENV_BOTTLENECK_DV, L.FS_ENV_BOTTLENECK) if lane_drop else (L.FS_ENV_MERGE_PO, L.FS_ENV_MERGE_MA)
        if env.FS_ENV not in ok:
            raise NotImplementedError("%s on %s is not built" % (type(env).__name__, type(network).__name__))
        return build_open_spec(env, num_replicas, rng)
    if env.FS_ENV in (L.FS_ENV_MERGE_PO, L.FS_ENV_MERGE_MA, L.FS_ENV_BOTTLENECK_DV, L.FS_ENV_BOTTLENECK):
        raise NotImplementedError("%s needs its open network (MergeNetwork / BottleneckNetwork)" % type(env).__name__)
    if not isinstance(network, (RingNetwork, FigureEightNetwork)):
        raise NotImplementedError("network %s is not built in the HIP step loop yet" % type(network).__name__)
    num_lanes = int(network.net_params.additional_params["lanes"])
    fig8 = isinstance(network, FigureEightNetwork)
    if fig8 and num_lanes != 1:
        raise NotImplementedError("multi-lane figure eight is not built in the HIP step loop yet")
    if len(network.net_params.inflows.get()) > 0:
        raise NotImplementedError("inflows are not built in the HIP step loop yet")
    R, N = int(num_replicas), veh_k.num_vehicles
    obs_perm = None
    if network.initial_config.shuffle:                     # envs/base.py:268-292: positions go to shuffled ids
        veh_k.set_slot_order(env.initial_ids)
        ids = veh_k.get_ids()
        obs_perm = np.array([ids.index(v) for v in env.initial_ids], dtype=np.int32)
    sort_vehicles = bool(ep.additional_params.get("sort_vehicles", False))
    if sort_vehicles and not (env.FS_ENV == L.FS_ENV_ACCEL or
                              (env.FS_ENV in (L.FS_ENV_LANE_CHANGE_ACCEL, L.FS_ENV_LANE_CHANGE_ACCEL_PO) and num_lanes > 1)):
        raise NotImplementedError("sort_vehicles is built for AccelEnv on single-lane closed loops and for "
                                  "LaneChangeAccelEnv on multi-lane rings")
    X, lanes = initial_positions(net_k, network.initial_config, N, R, rng)
    if num_lanes == 1 and obs_perm is None and N > 1 and np.any(np.diff(X[0]) < 0):
        # the placement is not in driving order (edges_distribution as a dict fills edge after edge in the order of
        # the dict, network/base.py:285-308): the simulator keeps its slots in ring order, so the vehicles are handed
        # to the slots by position and the id order travels as the observation permutation (as for shuffle)
        order = np.argsort(X[0], kind="stable")
        ids = veh_k.get_ids()
        veh_k.set_slot_order([ids[j] for j in order])
        obs_perm = order.astype(np.int32)
        X = X[:, order]
    slots = vehicle_slots(veh_k, env._rl_action_order())
    lengths = np.array([s["length"] for s in slots])
    check_placement(X, lengths, net_k.length(), lanes)
    if num_lanes > 1 and env.FS_ENV == L.FS_ENV_WAVE_ATTENUATION_PO:
        raise NotImplementedError("WaveAttenuationPOEnv on a multi-lane ring is not built")
    lc_modes = {int(veh_k.type_parameters[veh_k.get_type(v)]["lane_change_params"].lane_change_mode)
                for v in (veh_k.get_rl_ids() or veh_k.get_ids())}
    dt = sp.sim_step
    ramp = getattr(sp, "slowdown_ramp", None)
    space = env.action_space
    spec = dict(
        num_replicas=R, num_vehicles=N, num_rl=veh_k.num_rl_vehicles, vehicles=slots,
        user_controller_source=user_controller_source(slots),
        ring_length=np.full(R, float(network.net_params.additional_params["length"]) if not fig8
                            else net_k.length() - 4 * float(net_k.junction_length)), init_pos=X,
        segments=net_k.loop_segments(),
        junction=net_k.crossing_model(time_gap=float(getattr(sp, "crossing_time_gap", None) or 3.0)),
        sim_step=dt, slowdown_ramp=dt / (dt + 1e-3) if ramp is None else float(ramp),
        integrator="ballistic" if getattr(sp, "use_ballistic", False) else "euler",
        junction_mode=int(fig8 if getattr(sp, "junction_mode", None) is None else sp.junction_mode),
        junction_length=float(net_k.junction_length),
        crash_gap=float(getattr(sp, "crash_gap", 0.0)), max_speed=float(net_k.max_speed()),
        env=env.FS_ENV, target_velocity=float(ep.additional_params.get("target_velocity", 0.0)),
        action_low=float(space.low[0]) if N and veh_k.num_rl_vehicles else 0.0,      # acceleration bounds
        action_high=float(space.high[0]) if N and veh_k.num_rl_vehicles else 0.0,
        clip_actions=bool(ep.clip_actions) and not getattr(env, "UNCLIPPED_ACTIONS", False), evaluate=bool(ep.evaluate),
        po_max_length=float(env._po_max_length()), horizon=ep.horizon, warmup_steps=int(ep.warmup_steps),
        sims_per_step=int(ep.sims_per_step), seed=handle_seed(sp), noise_math=getattr(sp, "noise_math", "hw"),
        # (FS_MIXED keeps no previous-speed / acceleration fields: a scalar Env with precision='mixed' steps without
        # them -- k.vehicle.get_previous_speed / get_accel then report the values of the last reset)
        track_aux=bool(getattr(env, "_track_aux", True)) and getattr(sp, "precision", "f32") != "mixed",
        replica_offset=int(getattr(env, "_replica_offset", 0)),
        num_lanes=num_lanes, init_lane=lanes,
        lane_change_duration=float(ep.additional_params.get("lane_change_duration", 0)),
        lane_change_mode=max(lc_modes) if lc_modes else 512,
        # ML7: vehicles whose lane_change_mode lets SUMO change lanes do so on their own (simplified model M11)
        lane_change_cooldown_steps=max(1, int(round(float(getattr(sp, "lane_change_cooldown", 5.0)) / dt))),
        lane_change_min_gain=float(getattr(sp, "lane_change_min_gain", 10.0)),
        last_lc_quirk=bool(getattr(env, "LAST_LC_QUIRK", True)), sort_vehicles=sort_vehicles)
    if obs_perm is not None:
        spec["obs_perm"] = obs_perm
    return spec
