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
