"""oracle/bottleneck_accel.py (BottleneckAccelEnv with RL vehicles, flow/envs/bottleneck.py:486-757) on the CPU: the sizes
the reference's own test states (tests/fast_tests/test_environments.py:813-878: 2 numbers per edge of get_edge_list()),
hand-computed per-lane leaders / followers across the lane drops (vehicle/traci.py:776-950), the action pairing and a
long run with lane changes, arrivals and re-insertions."""
import numpy as np

from helpers import bottleneck_spec, bottleneck_tables, idm_vehicle
from oracle import opennet as O
from oracle import refsim as S
from oracle.bottleneck_accel import BottleneckAccelOracle


def accel_spec(n_human=6, n_rl=3, R=1, seed=0, lc_mode=512, **kw):
    tb = bottleneck_tables()
    spec = bottleneck_spec(R=R, cap_human=40, cap_rl=max(n_rl, 1), horizon=10 ** 6, seed=seed, env=O.ENV_BOTTLENECK, **kw)
    N = spec["num_vehicles"]
    spec["vehicles"] = [idm_vehicle(controller=S.CTRL_SIM, speed_mode=31, type=0) for _ in range(40)] + \
                       [dict(idm_vehicle(controller=S.CTRL_RL, rl_index=k, speed_mode=25, type=1, max_accel=3.0, max_decel=3.0),
                             lane_change_mode=lc_mode) for k in range(n_rl)] + \
                       [idm_vehicle(controller=S.CTRL_SIM, speed_mode=31, type=0) for _ in range(1 if n_rl == 0 else 0)]
    spec.update(num_rl=n_rl, ma_apply_actions=True, obs_cells=[], action_cells=[], target_velocity=30.0, action_low=-3.0,
                action_high=3.0, clip_actions=True)
    spec["inflows"] = [dict(type=0, route=-1, period=2.0, begin=1.0, end=86400.0, number=-1, depart_speed=10.0, depart_pos=5.0)]
    alive = np.zeros((R, N), dtype=bool)
    X, route = np.zeros((R, N)), np.zeros((R, N), dtype=np.int32)
    spec.update(init_alive=alive, init_pos=X, init_route=route, init_vel=np.zeros((R, N)))
    j, z = 0.1, 20.0
    path = [("1", 100.0, 4), (":2_0", j, 4), ("2", 310.0, 4), (":3_0", j, 4), ("3", 140.0, 4), (":4_0", z, 4),
            ("4", 280.0, 2), (":5_0", z, 2), ("5", 155.0, 1)]
    spec["accel_env"] = dict(path=path, connections={"4": {i: i // 2 for i in range(4)}, "5": {i: i // 2 for i in range(2)}},
                             edge_list=["1", "2", "3", "4", "5", "fake_edge"],
                             edge_length={"1": 100.0, "2": 310.0, "3": 140.0, "4": 280.0, "5": 155.0, "fake_edge": 1.0},
                             rl_names={40 + k: "rl_%d" % k for k in range(n_rl)}, lane_change_duration=5, scaling=1,
                             add_rl_if_exit=True, max_speed=23.0, max_accel=3.0, max_decel=3.0,
                             lane_change_mode={40 + k: lc_mode for k in range(n_rl)})
    return spec, tb


def place(spec, slot, x, path_, v=0.0):
    spec["init_alive"][:, slot], spec["init_pos"][:, slot], spec["init_route"][:, slot] = True, x, path_
    spec["init_vel"][:, slot] = v


def test_sizes_of_the_reference_test_and_the_padding_of_missing_rl_vehicles():
    spec, tb = accel_spec(n_rl=0)
    place(spec, 0, 50.0, 1, 7.0)
    ora = BottleneckAccelOracle(spec, np.float64)
    ora.reset()
    obs = ora.accel_state(0)
    assert obs.shape == (12,)                                   # test_environments.py:866-873
    np.testing.assert_allclose(obs[:2], [7.0 / 23.0, 1 / 100.0])
    assert (obs[2:] == 0).all()
    spec, tb = accel_spec(n_rl=3)
    place(spec, 41, 150.0, 2, 5.0)                              # only rl_1 is in the network: rl_0 and rl_2 are padded
    ora = BottleneckAccelOracle(spec, np.float64)
    ora.reset()
    obs = ora.accel_state(0)
    assert obs.shape == (4 * 3 + 16 * 3 + 12,)
    assert (obs[0:4] == 0).all() and (obs[8:12] == 0).all()
    np.testing.assert_allclose(obs[4:8], [(100.0 + 150.0 - 100.1) / 1000, 5.0 / 23.0, 2 / 4, 2 / 6])
    assert (obs[12:28] == 0).all() and (obs[44:60] == 0).all()
    # alone in the network: every lane empty -> headway = tailway = 1, no leader speed, get_speed('') behind
    np.testing.assert_allclose(obs[28:44], [1.0] * 8 + [0.0] * 4 + [-1001 / 23.0] * 4)


def test_lane_leaders_and_followers_across_the_lane_drops_by_hand():
    spec, tb = accel_spec(n_rl=1)
    s3, s4, s5 = tb["edge_start"]["3"], tb["edge_start"]["4"], tb["edge_start"]["5"]
    place(spec, 40, s3 + 100.0, 1, 10.0)                         # the RL vehicle: edge 3, lane 1
    place(spec, 0, s3 + 120.0, 1, 11.0)                          # same lane, 20 m ahead
    place(spec, 1, s3 + 100.0, 0, 12.0)                          # lane 0, SAME position: bisect_left makes it a leader
    place(spec, 2, s4 + 30.0, 3, 13.0)                           # entry lane 3 on edge 4 = lane 1 there: what lane 2 AND 3 lead to
    place(spec, 3, s3 - 60.0, 2, 14.0)                           # edge 2 (through :3_0), lane 2: follower of lane 2
    place(spec, 4, s3 + 40.0, 3, 15.0)                           # edge 3 lane 3, behind
    place(spec, 5, s3 + 10.0, 1, 16.0)                           # own lane, behind
    ora = BottleneckAccelOracle(spec, np.float64)
    ora.reset()
    obs = ora.accel_state(0)
    rel = obs[4:20]
    L = 5.0
    want_head = [0.0 - L, 20.0 - L, (140.0 - 100.0) + 20.0 + 30.0 - L, (140.0 - 100.0) + 20.0 + 30.0 - L]
    want_tail = [1000.0, 90.0 - L, 100.0 + 0.1 + 60.0 - 0.1 - L, 60.0 - L]
    # lane 2's follower sits 60 m before the start of edge 3 = on edge 2 at 310 + 0.1 - 60 (the internal :3_0 in between)
    want_tail[2] = 100.0 - (310.0 + 0.1 - 60.0) + (0.1 + 310.0) - L
    np.testing.assert_allclose(rel[0:4], np.array(want_head) / 1000, atol=1e-12)
    np.testing.assert_allclose(rel[4:8], np.array(want_tail) / 1000, atol=1e-12)
    np.testing.assert_allclose(rel[8:12], np.array([12.0, 11.0, 13.0, 13.0]) / 23.0)
    np.testing.assert_allclose(rel[12:16], np.array([-1001.0, 16.0, 14.0, 15.0]) / 23.0)
    # on edge 4 the vehicle of slot 2 looks back: lane 1's followers come through internal lane 2 only (prev_edge(...)[0])
    spec["accel_env"]["rl_names"] = {2: "rl_0"}
    spec["vehicles"][2], spec["vehicles"][40] = spec["vehicles"][40], spec["vehicles"][2]
    ora = BottleneckAccelOracle(spec, np.float64)
    ora.reset()
    hw, tw, ld, fl = ora._multi_lane(0, 2, _edge_dict(ora))
    assert ld == ["", ""] and fl[0] == 1 and fl[1] == 3           # lane 0 <- :4_0 lane 0 <- edge 3 lane 0; lane 1 <- lane 2
    np.testing.assert_allclose(tw, [30.0 + 20.0 + 40.0 - L, 30.0 + 20.0 + 140.0 + 0.1 + 60.0 - 0.1 - L], atol=1e-9)


def _edge_dict(ora, r=0):
    d = {}
    for i in ora._ids(r):
        edge, pos = ora._edge_pos(r, i)
        d.setdefault(edge, [[] for _ in range(4)])[ora._lane(r, i)].append((i, pos))
    for e in d:
        for lane in d[e]:
            lane.sort(key=lambda t: t[1])
    return d


def test_actions_pair_up_with_the_rl_vehicles_sorted_by_position_and_a_long_run_holds_together():
    spec, tb = accel_spec(n_rl=3)
    place(spec, 40, 300.0, 0, 5.0)
    place(spec, 41, 120.0, 1, 5.0)
    place(spec, 42, 210.0, 2, 5.0)
    ora = BottleneckAccelOracle(spec, np.float32)
    ora.reset()
    a = np.array([[1.0, 0.0, 2.0, 0.0, -1.0, 0.0]])               # sorted by x: rl_1 (120), rl_2 (210), rl_0 (300)
    v0 = ora.v[0, 40:43].copy()
    ora.step(a)
    dv = ora.v[0, 40:43] - v0
    assert dv[1] > 0.4 and dv[2] > 0.9 and dv[0] < -0.4          # rl_1 <- 1.0, rl_2 <- 2.0, rl_0 <- -1.0 (slowDown ramp)
    rng = np.random.default_rng(0)
    readded = changes = 0
    for k in range(700):
        act = rng.uniform(-1, 1, (1, 6)) * np.tile([3.0, 1.4], 3)
        if k % 3:
            act[:, 1::2] = 0.0
        before = {i: ora._lane(0, i) for i in (40, 41, 42) if ora.route[0, i] >= 0}
        obs, rew, done = ora.step(act)
        assert obs[0].shape == (72,) and np.isfinite(obs[0]).all() and np.isfinite(rew[0])
        after = {i: ora._lane(0, i) for i in (40, 41, 42) if ora.route[0, i] >= 0}
        readded += len(set(after) - set(before))
        changes += sum(1 for i in after if i in before and after[i] != before[i] and float(ora.x[0, i]) < tb["edge_start"]["4"] - 25)
    assert readded >= 3 and changes >= 5 and int(ora.total_arrived[0]) > 50
