"""Pin the oracle's reward / geometry / step restatement.

Sources: reference tests/fast_tests/test_rewards.py, test_environments.py,
test_scenario_base_class.py; the SUMO emission fixture
tests/fast_tests/test_files/ring_230_emission.csv (column subset committed as
tests/golden/ring_230_emission.csv); golden vectors from the imported reference."""
import csv
import json
import os
from collections import defaultdict

import numpy as np
from scipy.optimize import fsolve

from oracle import network as Net  # noqa: F401
from oracle import refsim as S
from oracle import rewards as Rw

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
RW = json.load(open(os.path.join(GOLDEN, "rewards.json")))

from helpers import idm_vehicle, ring_spec  # noqa: E402


# ----------------------------------------------------------------- rewards
def test_desired_velocity_vs_imported_reference():
    for c in RW["desired_velocity"]:
        got = Rw.desired_velocity(np.array(c["vel"]), c["target_velocity"], c["fail"])
        np.testing.assert_allclose(got, c["out"], rtol=1e-13, atol=1e-15)
        got = Rw.average_velocity(np.array(c["vel"]), c["fail"])
        np.testing.assert_allclose(got, c["average_velocity"], rtol=1e-13, atol=1e-15)


def test_desired_velocity_known_answers():
    # reference tests/fast_tests/test_rewards.py:44-45: 1 - sqrt(90)/10
    vel = np.zeros(10)
    vel[0] = 10.0
    np.testing.assert_allclose(Rw.desired_velocity(vel, 10), 1 - np.sqrt(90) / 10, rtol=1e-7)
    np.testing.assert_allclose(Rw.desired_velocity(vel, 10), RW["known"]["dv_one_at_target"], rtol=1e-13)
    np.testing.assert_allclose(Rw.desired_velocity(np.zeros(10), 10), RW["known"]["dv_all_stopped"], atol=1e-15)


def test_energy_consumption_known_answer():
    k = RW["known"]["energy"]
    np.testing.assert_allclose(Rw.energy_consumption(k["speed"], k["prev"], 0.1), k["out"], rtol=1e-13)


def test_wave_attenuation_reward_table():
    # reference tests/fast_tests/test_environments.py:552-613: 22 stopped vehicles
    v = np.zeros((1, 22))
    assert Rw.wave_attenuation_reward(v, None)[0] == 0
    assert Rw.wave_attenuation_reward(v, np.array([[0.0]]))[0] == 0
    np.testing.assert_allclose(Rw.wave_attenuation_reward(v, np.array([[1.0]]))[0], -4.0)
    assert Rw.wave_attenuation_reward(v, np.array([[1.0]]), fail=True)[0] == 0
    v = np.full((1, 22), 1.0)
    np.testing.assert_allclose(Rw.wave_attenuation_reward(v, np.array([[0.0]]))[0], 0.2)
    np.testing.assert_allclose(Rw.wave_attenuation_reward(v, np.array([[1.0]]))[0], -3.8)


def test_v_eq_max_function_roots():
    # reference tests/fast_tests/test_environments.py:435-447
    v = fsolve(Rw.v_eq_max_function, np.array(4), args=(22, 230))[0]
    np.testing.assert_allclose(v, 3.7136148111012934, rtol=1e-9)
    v = fsolve(Rw.v_eq_max_function, np.array(4), args=(22, 270))[0]
    np.testing.assert_allclose(v, 5.6143732387852054, rtol=1e-9)


# ----------------------------------------------------------------- geometry
def test_ring_geometry_tables():
    net = Net.ring_network(1000, junction_length=0.1)
    assert net.edge_length("bottom") == 250                      # test_scenario_base_class.py:656-660
    np.testing.assert_allclose(net.length(), 1000.4)
    np.testing.assert_allclose(net.non_internal_length(), 1000)
    assert net.get_edge(0) == ("bottom", 0)
    e, p = net.get_edge(250.05)
    assert e == ":right_0" and abs(p - 0.05) < 1e-12
    np.testing.assert_allclose(net.get_x("right", 3.0), 253.1)
    assert net.get_x("", 0) == -1001


def test_even_start_positions_invariants():
    # reference tests/fast_tests/test_scenario_base_class.py:143-238: equal spacing, x0, bunching
    net = Net.ring_network(230)
    pos, lanes = net.gen_even_start_pos(15, x0=5)
    x = np.array([net.get_x(e, p) for e, p in pos])
    assert abs(x[0] - 5) < 1e-12
    nominal = np.diff(x)
    # spacing is uniform up to the 0.1 m junction offsets of the edge-start table
    assert np.ptp(nominal) <= 0.1 + 1e-9
    pos_b, _ = net.gen_even_start_pos(15, bunching=10)
    xb = np.array([net.get_x(e, p) for e, p in pos_b])
    assert xb[-1] + 5 < 230 - 10 + 0.4 + 1e-9
    assert lanes == [0] * 15


def test_c1_start_positions():
    # x_i = i*(100/22 + 5) before junction offsets
    net = Net.ring_network(230, junction_length=0.0)
    pos, _ = net.gen_even_start_pos(22, bunching=20)
    nominal = np.arange(22) * (100 / 22 + 5)
    # the walk of base.py:380 runs in the edge-start table's own coordinate, so x is nominal
    x = np.array([net.get_x(e, p) for e, p in pos])
    np.testing.assert_allclose(x, nominal, atol=1e-9)
    assert pos[7][0] == "right" and abs(pos[7][1] - (nominal[7] - 57.6)) < 1e-9


# ----------------------------------------------------------------- the SUMO fixture (S6/S9)
def load_emission():
    rows = defaultdict(dict)
    with open(os.path.join(GOLDEN, "ring_230_emission.csv")) as f:
        for r in csv.DictReader(f):
            rows[round(float(r["time"]), 1)][int(r["id"].split("_")[1])] = (
                float(r["speed"]), r["edge_id"], float(r["relative_position"]))
    return rows


def simulate_fixture(ramp):
    # the fixture predates the 0.1 m junction offsets: seamless ring of 230 m
    N = 22
    x0 = np.arange(N) * (100 / N + 5)
    spec = ring_spec(R=1, N=N, junction_length=0.0, slowdown_ramp=ramp)
    spec["init_pos"] = x0[None, :]
    sim = S.RingOracle(spec)
    sim.reset()
    out = {0.1: (sim.v[0].copy(), sim.x[0].copy())}
    for k in range(1, 5):
        sim.step(None)
        out[round(0.1 * (k + 1), 1)] = (sim.v[0].copy(), sim.x[0].copy())
    return out


def test_emission_fixture_speeds_with_slowdown_ramp():
    em = load_emission()
    sim = simulate_fixture(0.1 / (0.1 + 1e-3))
    mism = 0
    for t in (0.1, 0.2, 0.3, 0.4, 0.5):
        assert len(em[t]) == 22
        for i in range(22):
            mism += round(float(sim[t][0][i]), 2) != em[t][i][0]
    assert mism == 0, "S6: dt/(dt+1e-3) ramp must reproduce all 110 rounded speeds"


def test_emission_fixture_rejects_unit_ramp():
    em = load_emission()
    sim = simulate_fixture(1.0)
    mism = sum(round(float(sim[t][0][i]), 2) != em[t][i][0]
               for t in (0.1, 0.2, 0.3, 0.4, 0.5) for i in range(22))
    assert mism == 23


def test_emission_fixture_positions_euler():
    em = load_emission()
    sim = simulate_fixture(0.1 / (0.1 + 1e-3))
    edge_start = {"bottom": 0.0, "right": 57.5, "top": 115.0, "left": 172.5}
    worst = 0.0
    for t in (0.1, 0.2, 0.3, 0.4, 0.5):
        for i in range(22):
            _, e, rel = em[t][i]
            worst = max(worst, abs(edge_start[e] + rel - float(sim[t][1][i])))
    assert worst <= 0.0051 + 1e-9    # 2-decimal rounding of the fixture


# ----------------------------------------------------------------- the SUMO figure-eight fixture (dt = 1 s)
def test_fig8_emission_fixture_positions_and_speeds():
    """13 of the 14 SUMO trajectories of tests/golden/fig8_emission.csv (positions incl. the edge changes top ->
    upper_ring across the 0.1 m internal edge, speeds) to the fixture's two decimals over three 1-second steps: S4-S9
    (Euler update with the new speed, slowDown ramp, safe-speed cap not binding) and the netconvert lengths hold at
    a sim_step ten times the ring fixture's.  The 14th (idm_8, entering the crossing behind a leaving vehicle) is
    the recorded deviation of the crossing model."""
    from helpers import FIG8_FIXTURE_DEVIATIONS, fig8_fixture_case
    spec, expected = fig8_fixture_case()
    sim = S.RingOracle(spec, np.float64)
    sim.reset()
    checked, worst_x, worst_v = 0, 0.0, 0.0
    for t in (2.0, 3.0, 4.0):
        sim.step(None)
        for i, (x, v) in expected[t].items():
            if (i, t) in FIG8_FIXTURE_DEVIATIONS:
                continue
            worst_x = max(worst_x, abs(float(sim.x[0, i]) - x))
            worst_v = max(worst_v, abs(float(sim.v[0, i]) - v))
            checked += 1
    assert checked == 38
    assert worst_v <= 0.0075 and worst_x <= 0.0105, (worst_x, worst_v)      # two decimals per column (x = edge start + position)
    # the recorded deviation, so that a change of the crossing model shows up here
    assert float(sim.v[0, 8]) == 0.0 and abs(expected[4.0][8][1] - 2.75) < 1e-9


def test_merge_emission_fixture_uncommanded_vehicles_accelerate_at_the_sumo_models_rate():
    """tests/golden/merge_emission.csv (reference fixture, SUMO output, sim_step 0.2): five SimCarFollowingController
    vehicles starting at rest gain accel * dt = 0.2 m/s per step (SumoCarFollowingParams(accel=1.0) of merge.json),
    minus a random shortfall of at most 0.07 m/s in some steps (SUMO-internal, not modelled).  sumo_idm_speed -- what
    uncommanded vehicles run here -- gives the 0.2."""
    from oracle import controllers as Ctl
    rows = defaultdict(dict)
    with open(os.path.join(GOLDEN, "merge_emission.csv")) as f:
        for r in csv.DictReader(f):
            rows[r["id"]][round(float(r["time"]), 1)] = float(r["speed"])
    n = 0
    for vid, tr in rows.items():
        times = sorted(tr)
        assert tr[times[0]] == 0.0
        for a, b in zip(times[:-1], times[1:]):
            v = np.array([tr[a]])
            nxt = float(Ctl.sumo_idm_speed(v, np.array([0.0]), np.array([1000.0]), np.array([False]), 0.2,
                                           accel=1.0, decel=1.5, tau=1.0, min_gap=2.5, max_speed=30.0)[0])
            gain, ref_gain = tr[b] - tr[a], nxt - tr[a]
            assert abs(ref_gain - 0.2) < 1e-3
            assert ref_gain - 0.075 <= gain <= ref_gain + 0.0051, (vid, a, gain)
            n += 1
    assert n == 22


# ----------------------------------------------------------------- step bookkeeping
def test_horizon_done_and_obs_layout():
    spec = ring_spec(R=3, N=5, bunching=0, horizon=4)
    sim = S.RingOracle(spec)
    obs = sim.reset()
    assert obs.shape == (3, 10)
    np.testing.assert_allclose(obs[:, :5], 0)
    np.testing.assert_allclose(obs[0, 5:], sim.x[0] / 230.0)
    for k in range(4):
        obs, rew, done = sim.step(None)
        assert done.all() == (k == 3)
    assert np.all((obs >= 0) & (obs <= 1))


def test_warmup_steps_are_taken():
    # reference tests/fast_tests/test_environment_base_class.py:260-277
    spec = ring_spec(R=1, N=5, bunching=0, warmup_steps=7, horizon=3)
    sim = S.RingOracle(spec)
    sim.reset()
    assert sim.time_counter[0] == 7
    done = False
    for _ in range(3):
        _, _, d = sim.step(None)
        done = bool(d[0])
    assert done


def test_sims_per_step():
    # reference tests/fast_tests/test_environment_base_class.py:284-303
    spec = ring_spec(R=1, N=5, bunching=0, sims_per_step=3)
    sim = S.RingOracle(spec)
    sim.reset()
    sim.step(None)
    assert sim.time_counter[0] == 3


def test_apply_acceleration_semantics():
    # reference tests/fast_tests/test_environment_base_class.py:146-188:
    # v1 ~= max(v0 + a*dt, 0) to 1 decimal for an RL vehicle under aggressive speed mode
    N = 5
    spec = ring_spec(R=1, N=N, bunching=0, num_rl=N)
    spec["vehicles"] = [idm_vehicle(controller=S.CTRL_RL, rl_index=i) for i in range(N)]
    spec["init_vel"] = np.array([[1.0, 2.0, 0.05, 3.0, 0.0]])
    sim = S.RingOracle(spec)
    sim.reset()
    acc = np.array([[1.0, -1.0, -3.0, 0.5, 2.0]])
    v0 = sim.v.copy()
    sim.step(acc)
    np.testing.assert_array_almost_equal(sim.v, np.maximum(v0 + acc * 0.1, 0), decimal=1)


def test_float32_trajectory_tracks_float64():
    spec = ring_spec(R=2, N=22)
    a = S.RingOracle(spec, np.float64)
    b = S.RingOracle(spec, np.float32)
    a.reset(), b.reset()
    for _ in range(300):
        a.step(None), b.step(None)
    assert b.x.dtype == np.float32
    np.testing.assert_allclose(b.x, a.x, atol=2e-3)
    np.testing.assert_allclose(b.v, a.v, atol=2e-3)


def test_per_lane_neighbours_known_answers_of_reference_test_vehicles():
    """reference tests/fast_tests/test_vehicles.py:205-253 (TestMultiLaneData.test_no_junctions_ring): 21 vehicles side by
    side in threes on a 3-lane ring of 230 m; lane leaders / headways / followers / tailways of test_0 -- the oracle's
    restatement of _multi_lane_headways (MultiLaneRingOracle.lane_neighbours) and the LaneChangeAccelPOEnv row built on it.
    The tailway 28.577143 crosses the ring's four junctions: 230 + 4 * 0.18 - 6 * 230 / 7 - 5."""
    from helpers import idm_vehicle, multilane_spec
    spec = multilane_spec(R=1, N=21, lanes=3, length=230.0, junction_length=0.18, env=S.ENV_LANE_CHANGE_ACCEL_PO)
    spec["init_pos"] = np.repeat(np.arange(7) * 230.0 / 7.0, 3)[None, :]
    spec["vehicles"] = [idm_vehicle(controller=S.CTRL_RL, rl_index=i) for i in range(21)]
    spec["num_rl"] = 21
    ora = S.MultiLaneRingOracle(spec, np.float64)
    obs = ora.reset()
    lead, foll, hw, tw = ora.lane_neighbours()
    assert list(lead[0, 0]) == [3, 1, 2] and list(foll[0, 0]) == [18, 19, 20]
    np.testing.assert_allclose(hw[0, 0], [27.85714285714286, -5, -5], atol=1e-9)
    np.testing.assert_allclose(tw[0, 0], [28.577143] * 3, atol=1e-6)
    np.testing.assert_allclose(obs[0, :12], [27.85714285714286, -5, -5] + [28.577143] * 3 + [0.0] * 6, atol=1e-6)
    assert obs.shape == (1, 4 * 3 * 21 + 21)
    # a vehicle alone in its lane is its own leader and follower one lap away (the reference's walk over the next / previous
    # edges ends on its own edge, vehicle/traci.py:883-905); an empty lane reads 1000 / ''
    spec = multilane_spec(R=1, N=3, lanes=3, length=100.0, junction_length=0.0, env=S.ENV_LANE_CHANGE_ACCEL_PO, n_rl=1)
    spec["init_pos"] = np.array([[20.0, 30.0, 10.0]])
    spec["init_lane"] = np.array([[1, 1, 0]], dtype=np.int32)
    ora = S.MultiLaneRingOracle(spec, np.float64)
    ora.reset()
    lead, foll, hw, tw = ora.lane_neighbours()
    assert list(lead[0, 2]) == [2, 0, -1] and list(foll[0, 2]) == [2, 1, -1]
    np.testing.assert_allclose(hw[0, 2], [95.0, 5.0, 1000.0])
    np.testing.assert_allclose(tw[0, 2], [95.0, 75.0, 1000.0])
