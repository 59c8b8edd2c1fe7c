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


