"""Resolve (env class, EnvParams, SumoParams, Network) into the plain ``spec`` dict that
flow_amd.sim.FlowSim packs into fs_config.  Host-side, runs once per construction."""
import numpy as np

from flow_amd import _lib as L
from flow_amd.controllers import RLController
from flow_amd.networks.figure_eight import FigureEightNetwork
from flow_amd.networks.ring import RingNetwork
from flow_amd.utils.exceptions import FatalFlowError


def vehicle_slots(vehicle_kernel, rl_order):
    """One fs_vehicle_spec dict per vehicle, in id (insertion) order (vehicle/traci.py:109-117)."""
    slots = []
    for veh_id in vehicle_kernel.get_ids():
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
                 rl_index=rl_order.index(veh_id) if isinstance(ctrl, RLController) else -1)
        slots.append(d)
    return slots


def initial_positions(network_kernel, initial_config, num_vehicles, num_replicas, rng=None):
    """[R,N] absolute start positions (+ [N] lanes).  The even/random placement of
    network/base.py:221-608 is computed once; a positive ``perturbation`` is drawn per
    replica (base.py:384-389), clamped to the vehicle's edge like the reference."""
    if initial_config.shuffle:
        raise NotImplementedError("InitialConfig(shuffle=True) is not built (slot order = ring order)")
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


def build_spec(env, num_replicas, rng=None):
    """The spec of ``env`` (a flow_amd Env under construction) replicated ``num_replicas`` times."""
    network, net_k, veh_k = env.network, env.k.network, env.k.vehicle
    sp, ep = env.sim_params, env.env_params
    if not isinstance(network, (RingNetwork, FigureEightNetwork)):
        raise NotImplementedError("network %s is not built in the HIP step loop yet" % type(network).__name__)
    num_lanes = int(network.net_params.additional_params["lanes"])
    fig8 = isinstance(network, FigureEightNetwork)
    if fig8 and num_lanes != 1:
        raise NotImplementedError("multi-lane figure eight is not built in the HIP step loop yet")
    if len(network.net_params.inflows.get()) > 0:
        raise NotImplementedError("inflows are not built in the HIP step loop yet")
    R, N = int(num_replicas), veh_k.num_vehicles
    slots = vehicle_slots(veh_k, env._rl_action_order())
    X, lanes = initial_positions(net_k, network.initial_config, N, R, rng)
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
        ring_length=np.full(R, float(network.net_params.additional_params["length"]) if not fig8
                            else net_k.length() - 4 * float(net_k.junction_length)), init_pos=X,
        segments=net_k.loop_segments(),
        junction=net_k.crossing_model(time_gap=float(getattr(sp, "crossing_time_gap", 3.0))),
        sim_step=dt, slowdown_ramp=dt / (dt + 1e-3) if ramp is None else float(ramp),
        integrator="ballistic" if getattr(sp, "use_ballistic", False) else "euler",
        junction_mode=int(fig8 if getattr(sp, "junction_mode", None) is None else sp.junction_mode),
        junction_length=float(net_k.junction_length),
        crash_gap=float(getattr(sp, "crash_gap", 0.0)), max_speed=float(net_k.max_speed()),
        env=env.FS_ENV, target_velocity=float(ep.additional_params.get("target_velocity", 0.0)),
        action_low=float(space.low[0]) if N and veh_k.num_rl_vehicles else 0.0,      # acceleration bounds
        action_high=float(space.high[0]) if N and veh_k.num_rl_vehicles else 0.0,
        clip_actions=bool(ep.clip_actions), evaluate=bool(ep.evaluate),
        po_max_length=float(env._po_max_length()), horizon=ep.horizon, warmup_steps=int(ep.warmup_steps),
        sims_per_step=int(ep.sims_per_step), seed=int(sp.seed or 0), track_aux=True,
        num_lanes=num_lanes, init_lane=lanes,
        lane_change_duration=float(ep.additional_params.get("lane_change_duration", 0)),
        lane_change_mode=max(lc_modes) if lc_modes else 512,
        last_lc_quirk=bool(getattr(env, "LAST_LC_QUIRK", True)))
    return spec
