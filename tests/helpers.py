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
