"""Shared spec builders for the oracle and the parity tests (plain dicts; the same
dict feeds oracle.refsim.RingOracle and flow_amd.sim.FlowSim)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import network as Net   # noqa: E402
from oracle import refsim as S      # noqa: E402

IDM_DEFAULT = [30, 1, 1, 1.5, 4, 2, 0, 0]       # car_following_models.py:437-447


def idm_vehicle(**kw):
    d = dict(controller=S.CTRL_IDM, p=IDM_DEFAULT, fail_safe=S.FAILSAFE_NONE, noise=0.0, delay=0.0,
             max_accel=2.6, max_decel=4.5, length=5.0, speed_mode=0, sumo_tau=1.0, sumo_min_gap=2.5,
             sumo_max_speed=30.0, rl_index=-1, initial_speed=0.0)
    d.update(kw)
    return d


def ring_spec(R=1, N=22, length=230.0, bunching=20.0, junction_length=0.0, horizon=1500, **kw):
    net = Net.ring_network(length, junction_length=junction_length)
    pos, _ = net.gen_even_start_pos(N, bunching=bunching)
    x0 = np.array([net.get_x(e, p) for e, p in pos])
    spec = dict(num_replicas=R, num_vehicles=N, num_rl=0, sim_step=0.1, junction_length=junction_length,
                ring_length=np.full(R, length), max_speed=30.0, env=S.ENV_ACCEL, target_velocity=10.0,
                action_low=-3.0, action_high=3.0, horizon=horizon, warmup_steps=0, sims_per_step=1,
                vehicles=[idm_vehicle() for _ in range(N)], init_pos=np.tile(x0, (R, 1)))
    spec.update(kw)
    return spec




def multilane_spec(R=4, N=21, lanes=3, length=230.0, horizon=100, n_rl=0, seed=0, junction_length=0.1, **kw):
    """RingNetwork with ``lanes`` lanes: vehicles placed side by side as gen_even_start_pos does
    (network/base.py:372-378), optionally with ``n_rl`` RL vehicles in the last slots."""
    net = Net.ring_network(length, lanes=lanes, junction_length=junction_length)
    pos, start_lanes = net.gen_even_start_pos(N)
    x0 = np.array([net.get_x(e, p) for e, p in pos])
    rng = np.random.default_rng(seed)
    X = np.tile(x0, (R, 1)) + np.abs(rng.normal(0, 0.2, (R, N)))
    veh = [idm_vehicle() for _ in range(N)]
    for k in range(n_rl):
        veh[N - n_rl + k] = idm_vehicle(controller=S.CTRL_RL, rl_index=k)
    spec = dict(num_replicas=R, num_vehicles=N, num_rl=n_rl, sim_step=0.1, junction_length=junction_length,
                ring_length=np.full(R, length), max_speed=30.0, env=S.ENV_LANE_CHANGE_ACCEL, target_velocity=10.0,
                action_low=-3.0, action_high=3.0, horizon=horizon, warmup_steps=0, sims_per_step=1,
                vehicles=veh, init_pos=X, num_lanes=lanes, init_lane=np.tile(np.array(start_lanes, dtype=np.int32), (R, 1)),
                lane_change_duration=5, lane_change_mode=512, last_lc_quirk=True)
    spec.update(kw)
    return spec


def figure_eight_tables(radius=30.0, lanes=1, junction_length=0.1, center_length=9.4, half_width=0.9,
                        veh_len=5.0, time_gap=3.0):
    """Physical segment table + crossing model of a one-lane figure eight, written out literally here so
    the tests do not depend on the product's network class: (phys_start, internal, flow_start, flow_slope)
    in route order bottom -> top -> upper_ring -> right -> left -> lower_ring (networks/figure_eight.py:189-206),
    flow_* from Flow's edge-start tables (figure_eight.py:225-263; internal edges without a table entry fall
    back to a constant, network/traci.py:280-287)."""
    net = Net.figure_eight_network(radius, lanes, center_length=center_length, junction_length=junction_length)
    tab = net.total_edgestarts_dict
    e = radius * np.pi / 2.0
    order = [("bottom", radius, False, tab["bottom"], 1.0),
             (":center_1", center_length, True, tab[":center_1"], 1.0),
             ("top", radius, False, tab["top"], 1.0),
             (":top_0", junction_length, True, tab[":top"], 0.0),
             ("upper_ring", 3 * e, False, tab["upper_ring"], 1.0),
             (":right_0", junction_length, True, tab[":right"], 0.0),
             ("right", radius, False, tab["right"], 1.0),
             (":center_0", center_length, True, tab[":center_0"], 1.0),
             ("left", radius, False, tab["left"], 1.0),
             (":left_0", junction_length, True, tab[":left"], 0.0),
             ("lower_ring", 3 * e, False, tab["lower_ring"], 1.0),
             (":bottom_0", junction_length, True, tab[":bottom"], 0.0)]
    segs, starts, s0 = [], {}, 0.0
    for name, length, internal, fs, slope in order:
        segs.append((s0, internal, fs, slope))
        starts[name] = s0
        s0 += length
    a_in, b_in = starts[":center_1"], starts[":center_0"]
    junction = dict(a_in=a_in, a_out=a_in + center_length, b_in=b_in, b_out=b_in + center_length,
                    lookahead=radius, time_gap=time_gap,
                    za_lo=a_in + center_length / 2 - half_width, za_hi=a_in + center_length / 2 + veh_len + half_width,
                    zb_lo=b_in + center_length / 2 - half_width, zb_hi=b_in + center_length / 2 + veh_len + half_width)
    return segs, junction, s0, starts, net


def figure_eight_spec(R=4, N=14, radius=30.0, horizon=200, seed=0, junction_length=0.1, center_length=9.4, **kw):
    """FigureEightNetwork, N vehicles placed by gen_even_start_pos (table coordinates -> loop coordinates)."""
    segs, junction, total, starts, net = figure_eight_tables(radius, 1, junction_length, center_length)
    pos, _ = net.gen_even_start_pos(N)
    x0 = np.array([starts[e] + p for e, p in pos])
    order = np.argsort(x0)
    assert (order == np.arange(N)).all()
    rng = np.random.default_rng(seed)
    X = np.tile(x0, (R, 1)) + np.abs(rng.normal(0, 0.2, (R, N)))
    veh = [idm_vehicle(speed_mode=1, max_decel=1.5) for _ in range(N)]
    spec = dict(num_replicas=R, num_vehicles=N, num_rl=0, sim_step=0.1, junction_length=junction_length,
                ring_length=np.full(R, total - 4 * junction_length), max_speed=30.0, env=S.ENV_ACCEL,
                target_velocity=20.0, action_low=-3.0, action_high=3.0, horizon=horizon, warmup_steps=0,
                sims_per_step=1, vehicles=veh, init_pos=X, junction_mode=1, segments=segs, junction=junction)
    spec.update(kw)
    return spec


def merge_tables(pre=200.0, merge=100.0, post=100.0, junction_length=0.1, center_length=22.5, inflow_len=100.0):
    """Route tables of MergeNetwork written out literally (independent of flow_amd.networks.merge so the
    host mirror can be checked against it).  Both routes share one coordinate with the merge point (start
    of edge 'center') at merge_x; Flow's edge-start table is flow/networks/merge.py:198-216; internal edges
    resolve to their table entry without the position (network/traci.py:280-287, slope 0)."""
    j, J = junction_length, center_length
    up0 = inflow_len + j + pre + J                  # length of the highway route upstream of the merge point
    up1 = inflow_len + j + merge + J
    merge_x = max(up0, up1)
    s0, s1 = merge_x - up0, merge_x - up1
    c_start = inflow_len + pre + 22.6               # ("center", INFLOW_EDGE_LEN + premerge + 22.6)
    r0 = [(s0, 0, 0.0, 1.0), (s0 + inflow_len, 1, inflow_len, 0.0), (s0 + inflow_len + j, 0, inflow_len + 0.1, 1.0),
          (s0 + inflow_len + j + pre, 1, inflow_len + pre + 0.1, 0.0), (merge_x, 0, c_start, 1.0)]
    im = inflow_len + pre + post + 22.6             # ("inflow_merge", ...)
    r1 = [(s1, 0, im, 1.0), (s1 + inflow_len, 1, 2 * inflow_len + pre + post + 22.6, 0.0),
          (s1 + inflow_len + j, 0, 2 * inflow_len + pre + post + 22.7, 1.0),
          (s1 + inflow_len + j + merge, 1, inflow_len + pre + 0.1, 0.0), (merge_x, 0, c_start, 1.0)]
    net_length = 2 * inflow_len + pre + merge + post + 2 * j + 2 * J
    return dict(routes=[dict(start=s0, segments=r0), dict(start=s1, segments=r1)], merge_x=merge_x,
                box_in=merge_x - J, end_x=merge_x + post, net_length=net_length)


def merge_spec(R=4, cap_human=12, cap_rl=4, num_rl=2, pre=200.0, merge=100.0, post=100.0, horizon=200, seed=0,
               q_highway=1800.0, q_rl=200.0, q_merge=300.0, n_init=3, env=None, time_gap=1.0, **kw):
    """MergeNetwork + MergePOEnv-like spec: IDM humans (noise 0.2, obey_safe_speed) and RL vehicles entering
    through three inflows as in examples/exp_configs/rl/multiagent/multiagent_merge.py:46-83."""
    from oracle import opennet as O
    tb = merge_tables(pre, merge, post)
    N = cap_human + cap_rl
    veh = [idm_vehicle(noise=0.2, speed_mode=1, type=0) for _ in range(cap_human)] + \
          [idm_vehicle(controller=S.CTRL_RL, rl_index=k, speed_mode=1, type=1) for k in range(cap_rl)]
    rng = np.random.default_rng(seed)
    alive = np.zeros((R, N), dtype=bool)
    alive[:, :n_init] = True
    s0 = tb["routes"][0]["start"]
    # initial humans spread over the highway edges, slot 0 furthest downstream (placement is host work)
    base = s0 + 100.1 + pre - 30.0 - 40.0 * np.arange(n_init)
    X = np.zeros((R, N))
    X[:, :n_init] = base[None, :] + rng.uniform(0, 5.0, (R, n_init))
    spec = dict(num_replicas=R, num_vehicles=N, num_rl=num_rl, sim_step=0.2, max_speed=30.0,
                env=O.ENV_MERGE_PO if env is None else env, target_velocity=20.0, action_low=-1.5, action_high=1.5,
                horizon=horizon, warmup_steps=0, sims_per_step=1, vehicles=veh, seed=seed,
                junction=dict(enabled=1, lookahead=merge, time_gap=time_gap), junction_mode=1,
                inflows=[dict(type=0, route=0, period=3600.0 / q_highway, begin=1.0, end=86400.0, number=-1,
                              depart_speed=10.0, depart_pos=5.0),
                         dict(type=1, route=0, period=3600.0 / q_rl, begin=1.0, end=86400.0, number=-1,
                              depart_speed=10.0, depart_pos=5.0),
                         dict(type=0, route=1, period=3600.0 / q_merge, begin=1.0, end=86400.0, number=-1,
                              depart_speed=7.5, depart_pos=5.0)],
                init_alive=alive, init_pos=X, init_vel=np.zeros((R, N)), init_route=np.zeros((R, N), dtype=np.int32),
                network="merge", **tb)
    spec.update(kw)
    return spec


def bottleneck_tables(junction_length=0.1, zipper_length=20.0, scaling=1):
    """BottleneckNetwork (flow/networks/bottleneck.py:111-165, scaling 1) on one coordinate: edges 1-5 of
    100 / 310 / 140 / 280 / 155 m with 4 / 4 / 4 / 2 / 1 lanes, short internal edges at nodes 2 and 3, zipper
    junctions of ``zipper_length`` at nodes 4 and 5.  Flow's edge-start table is ("1",0),("2",100),("3",405),
    ("4",425),("5",580) (:232-234); internal edges have no table entry (slope 0, value -1001 does not matter: the
    bottleneck envs only use edge-relative positions)."""
    j, z = junction_length, zipper_length
    e = [100.0, 310.0, 140.0, 280.0, 155.0]
    s1 = 0.0
    s2 = s1 + e[0] + j
    s3 = s2 + e[1] + j
    s4 = s3 + e[2] + z
    s5 = s4 + e[3] + z
    end = s5 + e[4]
    segs = [(s1, 0, 0.0, 1.0), (s1 + e[0], 1, -1001.0, 0.0), (s2, 0, 100.0, 1.0), (s2 + e[1], 1, -1001.0, 0.0),
            (s3, 0, 405.0, 1.0), (s3 + e[2], 1, -1001.0, 0.0), (s4, 0, 425.0, 1.0), (s4 + e[3], 1, -1001.0, 0.0),
            (s5, 0, 580.0, 1.0)]
    starts = dict(zip("12345", [s1, s2, s3, s4, s5]))
    lanes = dict(zip("12345", [4 * scaling, 4 * scaling, 4 * scaling, 2 * scaling, scaling]))
    lengths = dict(zip("12345", e))
    return dict(routes=[dict(start=0.0, segments=segs)], num_paths=4 * scaling, merge1_x=s4, merge2_x=s5, merge_x=s5,
                box_in=s5 - z, end_x=end, net_length=sum(e) + 2 * j + 2 * z + 1.0,   # + the 1 m rendering-only fake_edge
                edge_start=starts, edge_lanes=lanes, edge_length=lengths)


def segment_cells(tb, segments):
    """[(edge_start_x, lo, hi, lane, is_last_segment)] in the order the bottleneck envs walk them: edge, segment,
    lane (np.linspace(0, edge_length, n + 1) boundaries, bottleneck.py:796-812)."""
    cells = []
    for edge, n in segments:
        bounds = np.linspace(0, tb["edge_length"][edge], n + 1)
        for k in range(n):
            for lane in range(tb["edge_lanes"][edge]):
                cells.append((tb["edge_start"][edge], float(bounds[k]), float(bounds[k + 1]), lane, k == n - 1))
    return cells


def bottleneck_spec(R=4, cap_human=40, cap_rl=8, horizon=300, seed=0, q=2300.0, av_frac=0.1, env=None,
                    zipper_distance=50.0, warmup_steps=0, scaling=1, **kw):
    """singleagent_bottleneck.py: humans and RL vehicles all driven by the SUMO car-following model, inflow on
    edge 1 with departLane='random', BottleneckDesiredVelocityEnv head (141 observations, 20 actions)."""
    from oracle import opennet as O
    tb = bottleneck_tables(scaling=scaling)
    N = cap_human + cap_rl
    veh = [idm_vehicle(controller=S.CTRL_SIM, speed_mode=31, type=0) for _ in range(cap_human)] + \
          [idm_vehicle(controller=S.CTRL_RL, rl_index=k, speed_mode=9, type=1) for k in range(cap_rl)]
    rng = np.random.default_rng(seed)
    alive = np.zeros((R, N), dtype=bool)
    X = np.zeros((R, N))
    route = np.zeros((R, N), dtype=np.int32)
    # one human and one RL vehicle to start with, on edges 2.. (InitialConfig edges_distribution)
    alive[:, 0], X[:, 0], route[:, 0] = True, 300.0 + rng.uniform(0, 10, R), 1
    alive[:, cap_human], X[:, cap_human], route[:, cap_human] = True, 600.0 + rng.uniform(0, 10, R), 2
    obs_cells = segment_cells(tb, [("1", 1), ("2", 3), ("3", 3), ("4", 3), ("5", 1)])
    act_cells = segment_cells(tb, [("2", 2), ("3", 2), ("4", 2)])
    spec = dict(network="bottleneck", num_replicas=R, num_vehicles=N, num_rl=len(act_cells), sim_step=0.5, max_speed=23.0,
                env=O.ENV_BOTTLENECK_DV if env is None else env, target_velocity=40.0, action_low=-1.5, action_high=1.5,
                horizon=horizon, warmup_steps=warmup_steps, sims_per_step=1, vehicles=veh, seed=seed,
                junction=dict(enabled=0, lookahead=0.0, time_gap=1.0), junction_mode=1, speed_limit=23.0,
                zipper_distance=zipper_distance, scaling=scaling, obs_cells=obs_cells, action_cells=act_cells,
                obs_outflow_window=20, reward_outflow_window=10, track_followers=False,
                inflows=[dict(type=0, route=-1, period=3600.0 / (q * (1 - av_frac)), begin=1.0, end=86400.0, number=-1,
                              depart_speed=10.0, depart_pos=5.0),
                         dict(type=1, route=-1, period=3600.0 / (q * av_frac), begin=1.0, end=86400.0, number=-1,
                              depart_speed=10.0, depart_pos=5.0)],
                init_alive=alive, init_pos=X, init_vel=np.zeros((R, N)), init_route=route,
                **{k: v for k, v in tb.items() if k not in ("edge_start", "edge_lanes", "edge_length")})
    spec.update(kw)
    return spec


GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fig8_fixture_case():
    """tests/golden/fig8_emission.csv (the reference's tests/fast_tests/test_files/fig8_emission.csv, a SUMO
    emission file its visualizer tests read): 14 IDM vehicles on the one-lane figure eight, sim_step 1 s, four
    timestamps.  Returns (spec starting from the fixture's first timestamp, {time: {slot: (loop x, speed)}}, ids in
    slot order).  The experiment of that file is examples/exp_configs/non_rl/figure_eight.py:15-27 (IDMController
    defaults, speed_mode obey_safe_speed, decel 1.5) of a Flow version that still commanded vehicles on
    junction-internal edges (junction_mode 0: idm_8 gains 0.99 m/s inside ':center_0' at t = 4, Flow's IDM, not SUMO's
    2.6 m/s^2)."""
    import csv
    segs, junction, total, starts, _ = figure_eight_tables(30.0, 1, 0.1, 9.4)
    data = {}
    with open(os.path.join(GOLDEN_DIR, "fig8_emission.csv")) as f:
        for r in csv.DictReader(f):
            data.setdefault(int(r["id"].split("_")[1]), {})[float(r["time"])] = (
                starts[r["edge_id"]] + float(r["relative_position"]), float(r["speed"]))
    N = 14
    x0 = np.array([data[i][1.0][0] for i in range(N)])
    assert (np.diff(x0) > 0).all()                       # ids are in driving order already
    veh = [idm_vehicle(speed_mode=1, max_decel=1.5) for _ in range(N)]
    spec = dict(num_replicas=1, num_vehicles=N, num_rl=0, sim_step=1.0, junction_length=0.1,
                ring_length=np.full(1, total - 0.4), max_speed=30.0, env=S.ENV_ACCEL, target_velocity=20.0,
                action_low=-3.0, action_high=3.0, horizon=100, warmup_steps=0, sims_per_step=1, vehicles=veh,
                init_pos=x0[None, :], junction_mode=0, segments=segs, junction=junction)
    expected = {t: {i: data[i][t] for i in range(N)} for t in (2.0, 3.0, 4.0)}
    return spec, expected


# what the crossing model S-J (docs/HISTORY.md section 2) does NOT reproduce of that file: idm_8 starts 0.56 m before the
# crossing on the minor stream while idm_1's tail still covers the crossing point; SUMO lets it creep in behind the
# leaving vehicle (0.84 / 1.76 / 2.75 m/s), S-J holds it until the tail has left the box; idm_7 behind it feels that
# from the third step on
FIG8_FIXTURE_DEVIATIONS = {(8, 2.0), (8, 3.0), (8, 4.0), (7, 4.0)}
