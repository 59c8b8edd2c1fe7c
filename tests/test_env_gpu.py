"""GPU tests of the drop-in API: flow_amd.envs.* / make_create_env / VecFlowEnv driven the way
the reference's own tests drive flow.envs (tests/fast_tests/test_environment_base_class.py,
test_environments.py, test_experiment_base_class.py), with the oracle as the checker."""
import os
import random

import numpy as np
import pytest

from helpers import ring_spec
from oracle import refsim as S

pytestmark = pytest.mark.gpu


def ring_flow_params(n=22, length=230, horizon=1500, precision="f32", bunching=20, env_name=None, add_env=None,
                     rl=0, warmup=0, speed_mode="aggressive", **sim_kw):
    from flow_amd.controllers import ContinuousRouter, IDMController, RLController
    from flow_amd.core.params import (EnvParams, InitialConfig, NetParams, SumoCarFollowingParams, SumoParams,
                                      VehicleParams)
    from flow_amd.envs import AccelEnv
    from flow_amd.envs.ring.accel import ADDITIONAL_ENV_PARAMS
    from flow_amd.networks import RingNetwork
    vehicles = VehicleParams()
    vehicles.add(veh_id="idm", acceleration_controller=(IDMController, {}),
                 routing_controller=(ContinuousRouter, {}),
                 car_following_params=SumoCarFollowingParams(speed_mode=speed_mode), num_vehicles=n - rl)
    if rl:
        vehicles.add(veh_id="rl", acceleration_controller=(RLController, {}),
                     routing_controller=(ContinuousRouter, {}),
                     car_following_params=SumoCarFollowingParams(speed_mode="aggressive"), num_vehicles=rl)
    return dict(
        exp_tag="ring", env_name=env_name or AccelEnv, network=RingNetwork, simulator="traci",
        sim=SumoParams(sim_step=0.1, render=False, precision=precision, **sim_kw),
        env=EnvParams(horizon=horizon, warmup_steps=warmup, additional_params=add_env or dict(ADDITIONAL_ENV_PARAMS)),
        net=NetParams(additional_params={"length": length, "lanes": 1, "speed_limit": 30, "resolution": 40}),
        veh=vehicles, initial=InitialConfig(bunching=bunching))


def test_c1_accel_env_trajectory_f64_within_1e4_of_reference_arithmetic():
    """BASELINE configs[0]: 22-vehicle sugiyama ring, 1 env, 1500 steps; north_star bar 1e-4."""
    from flow_amd.utils.registry import make_create_env
    create_env, _ = make_create_env(ring_flow_params(precision="f64"))
    env = create_env()
    ora = S.RingOracle(ring_spec(R=1, N=22, junction_length=0.1, horizon=1500), np.float64)
    obs = env.reset()
    oref = ora.reset()
    np.testing.assert_allclose(obs, oref[0], atol=1e-7)
    ids = env.k.vehicle.get_ids()
    for k in range(1500):
        obs, rew, done, info = env.step(None)
        oref, rref, dref = ora.step(None)
        assert done == bool(dref[0]) == (k == 1499)
        if k % 250 == 0 or k == 1499:
            x = np.array(env.k.vehicle.get_x_by_id(ids))
            v = np.array(env.k.vehicle.get_speed(ids))
            assert np.abs(x - ora.x[0]).max() < 1e-4 and np.abs(v - ora.v[0]).max() < 1e-4
            np.testing.assert_allclose(rew, rref[0], atol=1e-6)
            np.testing.assert_allclose(np.array(env.k.vehicle.get_headway(ids)), ora.headways()[0], atol=1e-6)
    assert obs.shape == (44,) and info == {}
    assert np.mean(env.k.vehicle.get_speed(ids)) > 1.0
    assert env.k.vehicle.get_edge(ids[0]) in ("bottom", "right", "top", "left", ":right_0", ":top_0", ":left_0",
                                              ":bottom_0")
    env.terminate()


def test_apply_acceleration_semantics_one_decimal():
    # reference tests/fast_tests/test_environment_base_class.py:146-188
    from flow_amd.utils.registry import make_create_env
    env = make_create_env(ring_flow_params(n=5, rl=5, bunching=0))[0]()
    env.reset()
    ids = env.k.vehicle.get_ids()
    for vid, s in zip(ids, [1.0, 2.0, 0.05, 3.0, 0.0]):
        env.k.vehicle.test_set_speed(vid, s)
    v0 = np.array(env.k.vehicle.get_speed(ids))
    acc = np.array([1.0, -1.0, -3.0, 0.5, 2.0])
    env.step(acc)
    v1 = np.array(env.k.vehicle.get_speed(ids))
    np.testing.assert_array_almost_equal(v1, np.maximum(v0 + acc * 0.1, 0), decimal=1)
    env.terminate()


def test_rl_actions_speed_after_ten_steps():
    # reference tests/fast_tests/test_experiment_base_class.py:84-111: a=1 for 10 steps -> speed ~ 1
    from flow_amd.utils.registry import make_create_env
    env = make_create_env(ring_flow_params(n=1, rl=1, bunching=0))[0]()
    env.reset()
    for _ in range(10):
        obs, rew, done, _ = env.step([1.0])
    np.testing.assert_almost_equal(env.k.vehicle.get_speed("rl_0"), 1.0, decimal=1)
    env.terminate()


def test_action_clipping_box():
    # reference tests/fast_tests/test_environment_base_class.py:436-499
    from flow_amd.utils.registry import make_create_env
    env = make_create_env(ring_flow_params(n=4, rl=2, bunching=0))[0]()
    np.testing.assert_array_equal(env.clip_actions(np.array([10.0, -7.0])), [3.0, -3.0])
    assert env.clip_actions(None) is None
    env.reset()
    env.step(np.array([10.0, -7.0]))          # clipped to +-3 inside apply_rl_actions
    np.testing.assert_allclose(env.k.vehicle.get_accel("rl_0"), 3.0)
    env.terminate()


def test_warmup_and_sims_per_step_and_horizon():
    from flow_amd.utils.registry import make_create_env
    fp = ring_flow_params(n=5, bunching=0, horizon=4, warmup=6)
    fp["env"].sims_per_step = 3
    env = make_create_env(fp)[0]()
    env.reset()
    assert env.time_counter == 18            # warm-up steps were taken (reference :260-277)
    d = False
    for k in range(4):
        _, _, d, _ = env.step(None)
        assert d == (k == 3)
    assert env.time_counter == 30
    env.terminate()


def test_wave_attenuation_ring_length_sequence_and_po_obs():
    # reset draws random.randint(lo, hi) from Python's global RNG exactly like the reference
    # (wave_attenuation.py:172-174).  The reference's own test (test_environments.py:412-433) pins
    # 239, 256 after random.seed(9001) with restart_instance=True; those values also depend on
    # draws made inside sumolib/traci between resets (not in the repo), so what is pinned here is
    # the bare randint sequence of the same seed: 222, 239, 236.
    from flow_amd.envs import WaveAttenuationPOEnv
    from flow_amd.utils.registry import make_create_env
    fp = ring_flow_params(n=22, rl=1, length=260, bunching=0, horizon=50, warmup=5, env_name=WaveAttenuationPOEnv,
                          add_env={"max_accel": 1, "max_decel": 1, "ring_length": [220, 270]})
    env = make_create_env(fp)[0]()
    random.seed(9001)
    seen = []
    for _ in range(3):
        obs = env.reset()
        seen.append(env.net_params.additional_params["length"])
        assert obs.shape == (3,)
    assert seen == [222, 239, 236]
    np.testing.assert_allclose(env.k.network.length(), 236.4)
    obs, rew, done, _ = env.step([0.5])
    rl, lead = "rl_0", env.k.vehicle.get_leader("rl_0")
    exp = [env.k.vehicle.get_speed(rl) / 15., (env.k.vehicle.get_speed(lead) - env.k.vehicle.get_speed(rl)) / 15.,
           ((env.k.vehicle.get_x_by_id(lead) - env.k.vehicle.get_x_by_id(rl)) % env.k.network.length()) / 270.]
    np.testing.assert_allclose(obs, exp, atol=2e-6)
    vel = np.array(env.k.vehicle.get_speed(env.k.vehicle.get_ids()))
    np.testing.assert_allclose(rew, 4 * vel.mean() / 20 - 4 * 0.5, atol=1e-5)   # wave_attenuation.py:128-137
    assert env.compute_reward(None) == 0
    env.terminate()


def test_vec_env_step_rollout_and_reset_done():
    import torch
    from flow_amd.envs import VecFlowEnv
    R = 64
    fp = ring_flow_params(n=22, horizon=20)
    fp["initial"].perturbation = 0.3
    np.random.seed(3)
    vec = VecFlowEnv(fp, num_replicas=R)
    spec = dict(vec.env._spec)
    ora = S.RingOracle(spec, np.float32)
    obs = vec.reset()
    np.testing.assert_array_equal(obs.cpu().numpy(), ora.reset().astype(np.float32))
    for _ in range(5):
        o, r, d = vec.step()
        oo, rr, dd = ora.step(None)
    np.testing.assert_array_equal(o.cpu().numpy(), oo.astype(np.float32))
    np.testing.assert_array_equal(r.cpu().numpy(), rr.astype(np.float32))
    obs_k, rew_k, done_k = vec.rollout(15)
    for k in range(15):
        oo, rr, dd = ora.step(None)
    np.testing.assert_array_equal(obs_k[-1].cpu().numpy(), oo.astype(np.float32))
    assert bool(done_k[-1].all()) and not bool(done_k[-2].any())
    # device-side masked reset: mark half the replicas done, reset exactly those
    vec.step()
    vec._done.zero_()
    vec._done[::2] = 1
    before = vec.positions.copy()
    vec.reset_done()
    torch.cuda.synchronize()
    after = vec.positions
    np.testing.assert_array_equal(after[1::2], before[1::2])
    np.testing.assert_array_equal(after[::2], np.asarray(spec["init_pos"], dtype=np.float32)[::2])
    tc = vec.get_state(5)
    assert (tc[::2] == 0).all() and (tc[1::2] == 21).all()
    view = vec.vehicle_view(3)
    assert len(view.get_speed(view.get_ids())) == 22
    vec.close()


def test_vec_env_reaches_the_rollout_kernels_for_the_reference_ring_experiment():
    """The headline kernels must be reachable through the drop-in API, not only through the C ABI: the reference's
    ring experiment (examples/exp_configs/non_rl/ring.py:13-61) with the speed mode it ships ("right_of_way", the
    SumoCarFollowingParams default) and with "aggressive", in float32 and FS_MIXED, lands on k_rollout_pair; the same
    env with track_aux=True (scalar-Env accessors kept current) steps on the generic kernel -- bit-identically."""
    from flow_amd.envs import VecFlowEnv
    R, K = 40, 120
    for mode, want in (("right_of_way", "k_rollout_pair+speed_mode"), ("aggressive", "k_rollout_pair")):
        outs = {}
        for precision in ("f32", "mixed"):
            fp = ring_flow_params(n=22, horizon=100, precision=precision, speed_mode=mode)
            fp["initial"].perturbation = 0.3
            np.random.seed(4)
            vec = VecFlowEnv(fp, num_replicas=R)
            vec.reset()
            o, r, d = vec.rollout(K)
            assert vec.sim.last_kernel == want, (mode, precision, vec.sim.last_kernel)
            outs[precision] = (o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy(), vec.positions.copy())
            assert bool(d[99].all()) and not bool(d[98].any())
            vec.close()
        fp = ring_flow_params(n=22, horizon=100, speed_mode=mode)
        fp["initial"].perturbation = 0.3
        np.random.seed(4)
        gen = VecFlowEnv(fp, num_replicas=R, track_aux=True)
        gen.reset()
        o, r, d = gen.rollout(K)
        assert gen.sim.last_kernel.startswith("k_steps")
        np.testing.assert_array_equal(outs["f32"][0], o.cpu().numpy())
        np.testing.assert_array_equal(outs["f32"][1], r.cpu().numpy())
        np.testing.assert_array_equal(outs["f32"][3], gen.positions)
        assert len(gen.env.k.vehicle.get_previous_speed(gen.env.k.vehicle.get_ids())) == 22
        gen.close()
        # FS_MIXED stays within float32 rounding of the float32 run over 120 steps, and is not the same numbers
        assert np.abs(outs["mixed"][0] - outs["f32"][0]).max() < 1e-4


def test_custom_python_env_hooks_still_work():
    """A user subclass with Python get_state / compute_reward (the reference's extension point)."""
    from flow_amd.core import rewards
    from flow_amd.envs import Env
    from flow_amd.utils.registry import make_create_env
    from flow_amd.utils.spaces import Box

    class MyEnv(Env):
        @property
        def action_space(self):
            return Box(low=-1, high=1, shape=(self.initial_vehicles.num_rl_vehicles,), dtype=np.float32)

        @property
        def observation_space(self):
            return Box(low=0, high=100, shape=(self.initial_vehicles.num_vehicles,), dtype=np.float32)

        def _apply_rl_actions(self, rl_actions):
            self.k.vehicle.apply_acceleration(self.k.vehicle.get_rl_ids(), rl_actions)

        def get_state(self):
            return np.array(self.k.vehicle.get_headway(self.k.vehicle.get_ids()))

        def compute_reward(self, rl_actions, **kwargs):
            return rewards.average_velocity(self, fail=kwargs["fail"])

    fp = ring_flow_params(n=6, rl=1, bunching=0, horizon=30, env_name=MyEnv)
    env = make_create_env(fp)[0]()
    obs = env.reset()
    assert obs.shape == (6,) and np.all(obs > 0)
    for _ in range(10):
        obs, rew, done, _ = env.step([0.7])
    np.testing.assert_allclose(rew, np.mean(env.k.vehicle.get_speed(env.k.vehicle.get_ids())))
    np.testing.assert_allclose(obs.sum(), env.k.network.length() - 6 * 5, atol=1e-3)
    env.terminate()


def test_experiment_run_and_emission_csv(tmp_path):
    """Experiment.run contract (reference tests/fast_tests/test_experiment_base_class.py:40-77, 120-185):
    horizon honoured, runs deterministic, emission CSV with the reference's columns; and the first rows
    reproduce the reference's SUMO fixture to its 2-decimal precision."""
    import csv
    import os
    from flow_amd.core.experiment import Experiment
    fp = ring_flow_params(n=22, horizon=12, emission_path=str(tmp_path), junction_length=0.0)
    exp = Experiment(fp)
    info = exp.run(2, convert_to_csv=True)
    assert len(info["returns"]) == 2 and info["returns"][0] == info["returns"][1]
    assert exp.env.time_counter == 12
    path = info["emission_csv"]
    assert os.path.basename(path).startswith("ring_") and path.endswith("-emission.csv")
    rows = list(csv.DictReader(open(path)))
    for col in ("time", "id", "edge_id", "relative_position", "speed", "lane_number", "x", "y"):
        assert col in rows[0]
    assert len(rows) == 2 * 13 * 22
    golden = os.path.join(os.path.dirname(__file__), "golden", "ring_230_emission.csv")
    ref = {(r["id"], round(float(r["time"]), 1)): r for r in csv.DictReader(open(golden))}
    mine = {}
    for r in rows:                       # first run only
        key = (r["id"], round(float(r["time"]), 1))
        mine.setdefault(key, r)
    checked = 0
    for (vid, t), r in ref.items():
        if t > 0.5:
            continue
        m = mine[(vid, t)]
        assert round(float(m["speed"]), 2) == float(r["speed"])
        assert m["edge_id"] == r["edge_id"]
        # the fixture predates the 0.1 m junction offsets of the edge-start table: compare loop coordinates
        k = ["bottom", "right", "top", "left"].index(r["edge_id"])
        mine_s = k * 57.6 + float(m["relative_position"])
        ref_s = k * 57.5 + float(r["relative_position"])
        assert abs(mine_s - ref_s) <= 0.0051
        checked += 1
    assert checked == 110


def lane_change_flow_params(n=21, rl=1, lanes=3, horizon=50, lc_mode="aggressive"):
    from flow_amd.controllers import ContinuousRouter, IDMController, RLController
    from flow_amd.core.params import (EnvParams, InitialConfig, NetParams, SumoCarFollowingParams,
                                      SumoLaneChangeParams, SumoParams, VehicleParams)
    from flow_amd.envs import LaneChangeAccelEnv
    from flow_amd.envs.ring.lane_change_accel import ADDITIONAL_ENV_PARAMS
    from flow_amd.networks import RingNetwork
    vehicles = VehicleParams()
    vehicles.add(veh_id="test", acceleration_controller=(IDMController, {}), routing_controller=(ContinuousRouter, {}),
                 car_following_params=SumoCarFollowingParams(speed_mode="aggressive"), num_vehicles=n - rl)
    if rl:
        vehicles.add(veh_id="rl", acceleration_controller=(RLController, {}), routing_controller=(ContinuousRouter, {}),
                     car_following_params=SumoCarFollowingParams(speed_mode="aggressive"),
                     lane_change_params=SumoLaneChangeParams(lane_change_mode=lc_mode), num_vehicles=rl)
    return dict(exp_tag="ring3", env_name=LaneChangeAccelEnv, network=RingNetwork, simulator="traci",
                sim=SumoParams(sim_step=0.1, render=False),
                env=EnvParams(horizon=horizon, additional_params=dict(ADDITIONAL_ENV_PARAMS)),
                net=NetParams(additional_params={"length": 230, "lanes": lanes, "speed_limit": 30, "resolution": 40}),
                veh=vehicles, initial=InitialConfig(lanes_distribution=float("inf")))


def test_multilane_views_known_answers_of_reference_test_vehicles():
    """reference tests/fast_tests/test_vehicles.py:199-253: 21 vehicles on a 3-lane ring; lane leaders and
    lane headways of test_0 (the forward values do not cross a junction, so they do not depend on netconvert)."""
    from flow_amd.utils.registry import make_create_env
    env = make_create_env(lane_change_flow_params(n=21, rl=0))[0]()
    env.reset()
    k = env.k.vehicle
    assert sorted(k.get_lane_leaders("test_0")) == ["test_1", "test_2", "test_3"]
    np.testing.assert_allclose(sorted(k.get_lane_headways("test_0")), sorted([27.85714285714286, -5, -5]), atol=1e-5)
    assert sorted(k.get_lane_followers("test_0")) == ["test_18", "test_19", "test_20"]
    assert k.get_lane("test_4") == 1 and k.get_leader("test_0") == "test_3" and k.get_follower("test_3") == "test_0"
    assert env.observation_space.shape == (63,)
    env.terminate()


def test_lane_change_accel_env_step_through_the_env_api():
    from flow_amd.utils.registry import make_create_env
    env = make_create_env(lane_change_flow_params(n=9, rl=1, lanes=3, horizon=200))[0]()
    obs = env.reset()
    assert obs.shape == (27,)
    rl = "rl_0"
    lane0 = env.k.vehicle.get_lane(rl)
    assert lane0 == 2                                        # slot 8 of a side-by-side fill over 3 lanes
    # the fork's rate limit reads the headway: with headway h, changes are refused until time_counter > 5 + h
    h = env.k.vehicle.get_headway(rl)
    for _ in range(3):
        obs, rew, done, _ = env.step([0.5, -1])
    assert env.k.vehicle.get_lane(rl) == lane0 and env.time_counter == 3 <= 5 + h
    with pytest.raises(ValueError):
        env.step([0.0, 0.4])                                 # vehicle/traci.py:973-975
    env.terminate()

    # upstream meaning of get_last_lc (time of the last lane change): change at once, then rate-limited
    from flow_amd.envs import LaneChangeAccelEnv

    class UpstreamLC(LaneChangeAccelEnv):
        LAST_LC_QUIRK = False

    fp = lane_change_flow_params(n=9, rl=1, lanes=3, horizon=200)
    fp["env_name"] = UpstreamLC
    env = make_create_env(fp)[0]()
    env.reset()
    assert env.k.vehicle.get_last_lc(rl) == -float("inf")
    obs, rew, done, _ = env.step([0.2, -1])
    assert env.k.vehicle.get_lane(rl) == 1 and env.k.vehicle.get_last_lc(rl) == 1
    for _ in range(5):                                      # time_counter <= lane_change_duration + last_lc: refused
        obs, rew, done, _ = env.step([0.2, -1])
    assert env.k.vehicle.get_lane(rl) == 1 and env.time_counter == 6
    obs, rew, done, _ = env.step([0.2, -1])
    assert env.k.vehicle.get_lane(rl) == 0 and env.k.vehicle.get_last_lc(rl) == 7
    obs, rew, done, _ = env.step([0.2, -1])                 # already in lane 0: clipped (traci.py:982-984)
    assert env.k.vehicle.get_lane(rl) == 0
    np.testing.assert_allclose(obs[18:], np.array(env.k.vehicle.get_lane(env.k.vehicle.get_ids())) / 3.0)
    assert env.k.vehicle.get_leader(rl) in ("test_0", "test_3", "test_6")
    env.terminate()


def test_figure_eight_accel_env_runs_a_full_episode_and_matches_the_oracle():
    """examples/exp_configs/non_rl/figure_eight.py: 14 IDM vehicles, obey_safe_speed, AccelEnv, horizon 1500.
    The episode must complete without a crossing collision (vehicles queue at the crossing), observations stay in
    the Box, and the first 400 steps agree with the oracle on the same spec (f64, 1e-9)."""
    from flow_amd.controllers import ContinuousRouter, IDMController, StaticLaneChanger
    from flow_amd.core.params import EnvParams, NetParams, SumoCarFollowingParams, SumoParams, VehicleParams
    from flow_amd.envs import AccelEnv
    from flow_amd.envs.ring.accel import ADDITIONAL_ENV_PARAMS
    from flow_amd.networks import FigureEightNetwork
    from flow_amd.networks.figure_eight import ADDITIONAL_NET_PARAMS
    from flow_amd.utils.registry import make_create_env
    vehicles = VehicleParams()
    vehicles.add(veh_id="idm", acceleration_controller=(IDMController, {}),
                 lane_change_controller=(StaticLaneChanger, {}), routing_controller=(ContinuousRouter, {}),
                 car_following_params=SumoCarFollowingParams(speed_mode="obey_safe_speed", decel=1.5),
                 initial_speed=0, num_vehicles=14)
    fp = dict(exp_tag="figure8", env_name=AccelEnv, network=FigureEightNetwork, simulator="traci",
              sim=SumoParams(render=False, precision="f64"),
              env=EnvParams(horizon=1500, additional_params=dict(ADDITIONAL_ENV_PARAMS)),
              net=NetParams(additional_params=dict(ADDITIONAL_NET_PARAMS)), veh=vehicles)
    env = make_create_env(fp)[0]()
    ora = S.RingOracle(dict(env._spec), np.float64)
    obs = env.reset()
    ora.reset()
    ids = env.k.vehicle.get_ids()
    edges_seen = set()
    for k in range(1500):
        obs, rew, done, _ = env.step(None)
        assert done == (k == 1499), "no crossing collision before the horizon"
        if k < 400:
            o_ref, r_ref, _ = ora.step(None)
            if k % 50 == 0:
                np.testing.assert_allclose(obs, o_ref[0], atol=1e-6)
                np.testing.assert_allclose(rew, r_ref[0], atol=1e-6)
        if k % 25 == 0:
            edges_seen.update(env.k.vehicle.get_edge(ids))
    assert obs.shape == (28,) and (obs >= 0).all() and (obs <= 1).all()
    assert {"bottom", "top", "upper_ring", "right", "left", "lower_ring"} <= edges_seen
    assert np.mean(env.k.vehicle.get_speed(ids)) > 2.0
    x = env.k.vehicle.get_x_by_id(ids[0])
    e, p = env.k.vehicle.get_edge(ids[0]), env.k.vehicle.get_position(ids[0])
    assert abs(env.k.network.get_x(e, p) - x) < 1e-9
    env.terminate()


def test_vec_wave_attenuation_po_redraws_ring_length_per_replica():
    """examples/exp_configs/rl/singleagent/singleagent_ring.py: 21 IDM + 1 RL on a ring whose length is redrawn in
    [220, 270] at every reset, 750... (here 20) warm-up steps with the RL vehicle SUMO-driven.  Vectorised: every
    replica draws its own length; the rollout must match an oracle built from the same per-replica lengths."""
    import torch
    from flow_amd import _lib as L
    from flow_amd.envs import VecFlowEnv, WaveAttenuationPOEnv
    R = 48
    fp = ring_flow_params(n=22, rl=1, length=260, bunching=0, horizon=40, warmup=20, env_name=WaveAttenuationPOEnv,
                          add_env={"max_accel": 1, "max_decel": 1, "ring_length": [220, 270]})
    fp["env"].clip_actions = False
    vec = VecFlowEnv(fp, num_replicas=R, seed=5)
    obs = vec.reset()
    lengths = vec.get_state(L.FS_FIELD_RING_LENGTH)
    assert lengths.min() >= 220 and lengths.max() <= 270 and len(np.unique(lengths)) > 10
    init = vec.get_state(L.FS_FIELD_INIT_POS)
    assert (init < (lengths + 0.4)[:, None]).all() and (np.diff(init, axis=1) >= 5).all()
    spec = dict(vec.env._spec)
    spec["ring_length"], spec["init_pos"] = lengths.astype(np.float64), init.astype(np.float64)
    ora = S.RingOracle(spec, np.float32)
    np.testing.assert_array_equal(obs.cpu().numpy(), ora.reset().astype(np.float32))
    assert obs.shape == (R, 3)
    acts = torch.rand((30, R, 1), device=obs.device) * 2 - 1
    o_k, r_k, d_k = vec.rollout(30, acts)
    for k in range(30):
        o_ref, r_ref, d_ref = ora.step(acts[k].cpu().numpy())
    np.testing.assert_array_equal(o_k[-1].cpu().numpy(), o_ref.astype(np.float32))
    np.testing.assert_array_equal(r_k[-1].cpu().numpy(), r_ref.astype(np.float32))
    # episode end for a third of the replicas: they (and only they) draw new lengths
    vec._done.zero_()
    vec._done[::3] = 1
    vec.reset_done()
    new_lengths = vec.get_state(L.FS_FIELD_RING_LENGTH)
    np.testing.assert_array_equal(new_lengths[1::3], lengths[1::3])
    assert (new_lengths[::3] != lengths[::3]).any()
    tc = vec.get_state(L.FS_FIELD_TIME)
    assert (tc[::3] == 20).all() and (tc[1::3] == 50).all()
    vec.close()


def test_figure_eight_wave_attenuation_po_env_vectorised():
    """BASELINE configs[2] as written: FigureEightNetwork, 13 IDM + 1 RL, WaveAttenuationPOEnv (3 observations)."""
    import torch
    from flow_amd.controllers import ContinuousRouter, IDMController, RLController
    from flow_amd.core.params import EnvParams, NetParams, SumoCarFollowingParams, SumoParams, VehicleParams
    from flow_amd.envs import VecFlowEnv, WaveAttenuationPOEnv
    from flow_amd.networks import FigureEightNetwork
    from flow_amd.networks.figure_eight import ADDITIONAL_NET_PARAMS
    veh = VehicleParams()
    veh.add(veh_id="human", acceleration_controller=(IDMController, {"noise": 0.2}),
            routing_controller=(ContinuousRouter, {}),
            car_following_params=SumoCarFollowingParams(speed_mode="obey_safe_speed", decel=1.5), num_vehicles=13)
    veh.add(veh_id="rl", acceleration_controller=(RLController, {}), routing_controller=(ContinuousRouter, {}),
            car_following_params=SumoCarFollowingParams(speed_mode="obey_safe_speed", decel=1.5), num_vehicles=1)
    fp = dict(exp_tag="figure_eight", env_name=WaveAttenuationPOEnv, network=FigureEightNetwork, simulator="traci",
              sim=SumoParams(sim_step=0.1, render=False, seed=3, precision="f64"),
              env=EnvParams(horizon=100, warmup_steps=10,
                            additional_params={"max_accel": 3, "max_decel": 3, "ring_length": None}),
              net=NetParams(additional_params=dict(ADDITIONAL_NET_PARAMS)), veh=veh)
    R = 32
    vec = VecFlowEnv(fp, num_replicas=R)
    ora = S.RingOracle(dict(vec.env._spec), np.float64)
    obs = vec.reset()
    np.testing.assert_allclose(obs.cpu().numpy(), ora.reset().astype(np.float32), atol=1e-6)
    assert obs.shape == (R, 3)
    acts = torch.rand((60, R, 1), device=obs.device) * 6 - 3
    o_k, r_k, d_k = vec.rollout(60, acts)
    for k in range(60):
        o_ref, r_ref, d_ref = ora.step(acts[k].cpu().numpy())
    np.testing.assert_allclose(o_k[-1].cpu().numpy(), o_ref.astype(np.float32), atol=1e-5)
    np.testing.assert_allclose(r_k[-1].cpu().numpy(), r_ref.astype(np.float32), atol=1e-5)
    np.testing.assert_array_equal(d_k[-1].cpu().numpy().astype(bool), d_ref)
    vec.close()


def test_examples_simulate_script_runs_reference_style_configs(tmp_path):
    """examples/simulate.py ring / figure_eight: experiment files written with ``from flow...`` imports run through
    flow_amd.install_as_flow() and Experiment.run, and the ring run leaves an emission CSV behind."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "simulate.py"), "ring", "--gen_emission"],
                         cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "steps/second" in out.stdout and "Round 0, return" in out.stdout
    files = os.listdir(os.path.join(str(tmp_path), "data"))
    assert len(files) == 1 and files[0].endswith("-emission.csv")
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "simulate.py"), "figure_eight"],
                         cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]


def test_reference_sorting_tests_with_shuffled_start_positions():
    """tests/fast_tests/test_environments.py:281-330 (test_sorting / test_no_sorting): with InitialConfig.shuffle the
    ids are not in ring order; sort_vehicles=True makes sorted_ids ascending in absolute_position, False leaves
    get_ids() untouched.  Plus: the sorted observation equals the oracle's, RL actions reach the vehicles in sorted
    order, and a shuffle reset draws a new placement."""
    import random
    from flow_amd.controllers import ContinuousRouter, IDMController, RLController
    from flow_amd.core.params import EnvParams, InitialConfig, NetParams, SumoParams, VehicleParams
    from flow_amd.envs import AccelEnv
    from flow_amd.networks.ring import ADDITIONAL_NET_PARAMS, RingNetwork
    from oracle import refsim as S

    def network():
        v = VehicleParams()
        v.add("rl", acceleration_controller=(RLController, {}), routing_controller=(ContinuousRouter, {}),
              num_vehicles=2)
        v.add("human", acceleration_controller=(IDMController, {}), routing_controller=(ContinuousRouter, {}),
              num_vehicles=8)
        return RingNetwork("ring", v, NetParams(additional_params=dict(ADDITIONAL_NET_PARAMS)),
                           InitialConfig(shuffle=True))

    add = {"max_accel": 3, "max_decel": 3, "target_velocity": 10, "sort_vehicles": True}
    random.seed(11)
    env = AccelEnv(sim_params=SumoParams(sim_step=0.1), network=network(),
                   env_params=EnvParams(horizon=200, additional_params=add))
    env.reset()
    env.additional_command()
    sorted_ids = env.sorted_ids
    positions = [env.absolute_position[veh_id] for veh_id in sorted_ids]
    assert all(positions[i] <= positions[i + 1] for i in range(len(positions) - 1))
    assert sorted_ids != env.k.vehicle.get_ids()                       # the shuffle really moved somebody
    first = dict(env.initial_state)
    ora = S.RingOracle(env._spec, np.float32)
    ora.reset()
    rng = np.random.default_rng(0)
    for k in range(200):
        a = rng.uniform(-1, 1, 2).astype(np.float32)
        obs, rew, done, _ = env.step(a)
        o_ref, r_ref, d_ref = ora.step(a[None, :])
        np.testing.assert_array_equal(obs.astype(np.float32), o_ref[0].astype(np.float32))
        # the host's sorted_ids (running absolute_position) is the order the kernel used for this observation
        x_by_id = {v: env.k.vehicle.get_x_by_id(v) / env.k.network.length() for v in env.sorted_ids}
        np.testing.assert_allclose(obs[10:], [x_by_id[v] for v in env.sorted_ids], atol=1e-6)
    env.reset()                                                        # shuffle: a new placement every reset
    assert dict(env.initial_state) != first
    env.terminate()

    add["sort_vehicles"] = False
    env = AccelEnv(sim_params=SumoParams(sim_step=0.1), network=network(),
                   env_params=EnvParams(horizon=50, additional_params=add))
    env.reset()
    env.additional_command()
    assert list(env.sorted_ids) == env.k.vehicle.get_ids()
    obs, _, _, _ = env.step(np.array([0.5, -0.5]))
    # observation in get_ids() order although the ring order is shuffled
    np.testing.assert_allclose(obs[10:], [env.k.vehicle.get_x_by_id(v) / env.k.network.length()
                                          for v in env.k.vehicle.get_ids()], atol=1e-6)
    assert env.k.vehicle.get_accel("rl_0") == pytest.approx(0.5) and env.k.vehicle.get_accel("rl_1") == pytest.approx(-0.5)
    env.terminate()


def test_lane_change_accel_po_env_reference_tests_and_observation():
    """tests/fast_tests/test_environments.py:117-198 (TestLaneChangeAccelPOEnv: 1 RL + 1 human on a one-lane ring:
    observation size 5, action size 2, observed ['human_0']) and the per-lane lists on a 3-lane ring."""
    from flow_amd.controllers import ContinuousRouter, IDMController, RLController
    from flow_amd.core.params import EnvParams, InitialConfig, NetParams, SumoParams, VehicleParams
    from flow_amd.envs import LaneChangeAccelPOEnv
    from flow_amd.networks.ring import ADDITIONAL_NET_PARAMS, RingNetwork
    add = {"max_accel": 3, "max_decel": 3, "target_velocity": 10, "lane_change_duration": 5, "sort_vehicles": False}
    v = VehicleParams()
    v.add("rl", acceleration_controller=(RLController, {}), num_vehicles=1)
    v.add("human", acceleration_controller=(IDMController, {}), num_vehicles=1)
    net = RingNetwork("test_merge", v, NetParams(additional_params=dict(ADDITIONAL_NET_PARAMS)))
    env = LaneChangeAccelPOEnv(sim_params=SumoParams(), network=net, env_params=EnvParams(additional_params=add))
    assert env.observation_space.shape == (5,) and env.action_space.shape == (2,)
    assert list(env.action_space.low) == [-3, -1] and list(env.action_space.high) == [3, 1]
    env.reset()
    obs, _, _, _ = env.step(None)
    env.additional_command()
    assert env.k.vehicle.get_observed_ids() == ["human_0"]
    x_rl, x_h = env.k.vehicle.get_x_by_id("rl_0"), env.k.vehicle.get_x_by_id("human_0")
    L_ = env.k.network.length()
    # [headway, tailway, leader speed / v_max, follower speed / v_max, ego speed]; head/tailway in METRES (see class doc)
    np.testing.assert_allclose(obs, [(x_h - x_rl) % L_ - 5, (x_rl - x_h) % L_ - 5, env.k.vehicle.get_speed("human_0") / 30,
                                     env.k.vehicle.get_speed("human_0") / 30, env.k.vehicle.get_speed("rl_0")], atol=1e-5)
    env.terminate()

    lanes3 = dict(ADDITIONAL_NET_PARAMS)
    lanes3["lanes"] = 3
    v = VehicleParams()
    v.add("human", acceleration_controller=(IDMController, {}), routing_controller=(ContinuousRouter, {}),
          num_vehicles=19)
    v.add("rl", acceleration_controller=(RLController, {}), routing_controller=(ContinuousRouter, {}), num_vehicles=2)
    net = RingNetwork("ring3", v, NetParams(additional_params=lanes3), InitialConfig(lanes_distribution=float("inf")))
    env = LaneChangeAccelPOEnv(sim_params=SumoParams(sim_step=0.1), network=net,
                               env_params=EnvParams(horizon=50, additional_params=add))
    assert env.observation_space.shape == (4 * 2 * 3 + 2,)
    env.reset()
    for k in range(30):
        obs, rew, done, _ = env.step(np.array([0.5, 1 if k == 5 else 0, -0.5, 0]))
    assert obs.shape == (4 * 2 * 3 + 1,)                 # the reference returns inside its loop over the RL vehicles
    veh = env.k.vehicle
    rl = veh.get_rl_ids()[0]
    # the kernel's row (float32 arithmetic on the device) against the host accessors (float64 on the same state)
    np.testing.assert_allclose(obs[0:3], veh.get_lane_headways(rl), atol=1e-4)
    np.testing.assert_allclose(obs[3:6], veh.get_lane_tailways(rl), atol=1e-4)
    np.testing.assert_allclose(obs[6:9], [s / 30 for s in veh.get_lane_leaders_speed(rl)], atol=1e-6)
    np.testing.assert_allclose(obs[9:12], [s / 30 for s in veh.get_lane_followers_speed(rl)], atol=1e-6)
    assert (obs[12:24] == 0).all() and obs[24] == veh.get_speed(rl)

    class AllRl(LaneChangeAccelPOEnv):
        RETURN_IN_LOOP_QUIRK = False
    env.terminate()
    env = AllRl(sim_params=SumoParams(sim_step=0.1), network=net, env_params=EnvParams(horizon=50, additional_params=add))
    env.reset()
    obs, _, _, _ = env.step(None)
    assert obs.shape == (26,) and obs[24] == env.k.vehicle.get_speed("rl_0") and obs[25] == env.k.vehicle.get_speed("rl_1")
    env.terminate()


def test_lane_change_accel_env_sort_vehicles_through_the_env_api():
    """LaneChangeAccelEnv(sort_vehicles=True): the observation lists the vehicles in the order of the host-side
    ``sorted_ids`` (AccelEnv.absolute_position, accel.py:134-169) and action pair k reaches the k-th RL vehicle of
    that order -- the kernel's ranking and the reference's Python bookkeeping must agree step by step."""
    from flow_amd.utils.registry import make_create_env
    fp = lane_change_flow_params(n=12, rl=3, lanes=2, horizon=150)
    fp["env"].additional_params["sort_vehicles"] = True
    fp["env"].additional_params["lane_change_duration"] = 0
    env = make_create_env(fp)[0]()
    obs = env.reset()
    veh = env.k.vehicle
    rng = np.random.default_rng(4)
    changed = False
    for k in range(150):
        order = env.sorted_ids                                   # before the step: who takes which action pair
        rl_order = [v for v in order if v in veh.get_rl_ids()]
        a = np.zeros(6)
        a[0::2] = rng.uniform(-1, 2, 3)
        a[1::2] = rng.integers(-1, 2, 3)
        lanes_before = {v: veh.get_lane(v) for v in rl_order}
        obs, rew, done, _ = env.step(a)
        order = env.sorted_ids                                   # after additional_command: the observation order
        n = len(order)
        np.testing.assert_allclose(obs[:n], np.array(veh.get_speed(order)) / 30.0, atol=1e-6)
        np.testing.assert_allclose(obs[2 * n:], np.array(veh.get_lane(order)) / 2.0, atol=1e-6)
        for j, v in enumerate(rl_order):                         # a lane change can only go where pair j pointed
            moved = veh.get_lane(v) - lanes_before[v]
            assert moved == 0 or moved == int(a[2 * j + 1]), (k, v, moved, a)
            changed = changed or moved != 0
        if order != veh.get_ids():
            reordered = True
    assert changed and reordered
    env.terminate()


def test_edges_distribution_as_a_dict_reference_test():
    """tests/fast_tests/test_scenario_base_class.py:389-409: {edge: number of vehicles}; a count that does not match
    the vehicles raises AssertionError; the edges are filled in the order of the dict (here not the driving order:
    the simulator's slots are handed out by position and the id order travels as the observation permutation)."""
    from flow_amd.controllers import ContinuousRouter, IDMController
    from flow_amd.core.params import EnvParams, InitialConfig, NetParams, SumoParams, VehicleParams
    from flow_amd.envs import AccelEnv
    from flow_amd.envs.ring.accel import ADDITIONAL_ENV_PARAMS
    from flow_amd.networks import RingNetwork

    def make(edges, lanes=1):
        vehicles = VehicleParams()
        vehicles.add(veh_id="test", acceleration_controller=(IDMController, {}),
                     routing_controller=(ContinuousRouter, {}), num_vehicles=15)
        net = RingNetwork("ring", vehicles, NetParams(additional_params={"length": 230, "lanes": lanes,
                                                                         "speed_limit": 30, "resolution": 40}),
                          InitialConfig(edges_distribution=edges))
        return AccelEnv(EnvParams(additional_params=dict(ADDITIONAL_ENV_PARAMS)), SumoParams(sim_step=0.1), net)

    with pytest.raises(AssertionError):
        make({"top": 2, "bottom": 1})
    edges = {"top": 5, "bottom": 6, "left": 4}
    for lanes in (1, 4):
        env = make(edges, lanes)
        env.reset()
        veh = env.k.vehicle
        for edge in edges:
            assert len(veh.get_ids_by_edge(edge)) == edges[edge]
        assert [veh.get_edge(v) for v in veh.get_ids()] == ["top"] * 5 + ["bottom"] * 6 + ["left"] * 4
        if lanes == 1:
            assert veh.get_leader("test_4") == "test_11" and veh.get_leader("test_14") == "test_5"
            obs = env.reset()                                   # observation in id order: speeds, then positions
            x = np.array([veh.get_x_by_id(v) for v in veh.get_ids()])
            np.testing.assert_allclose(obs[15:], x / env.k.network.length(), atol=1e-6)
        for _ in range(20):
            obs, rew, done, _ = env.step(None)
        assert np.isfinite(obs).all() and min(veh.get_headway(veh.get_ids())) > 0
        env.terminate()


def test_shuffle_on_a_multi_lane_ring():
    """InitialConfig(shuffle=True) with lanes > 1 (envs/base.py:268-292): ids keep their order in get_ids() and in the
    observation, the places (position, lane) are handed out in shuffled order, and a reset re-shuffles."""
    import random
    from flow_amd.utils.registry import make_create_env
    fp = lane_change_flow_params(n=12, rl=2, lanes=3, horizon=60)
    fp["initial"].shuffle = True
    random.seed(5)
    env = make_create_env(fp)[0]()
    obs = env.reset()
    veh = env.k.vehicle
    ids = veh.get_ids()
    assert ids == ["test_%d" % i for i in range(10)] + ["rl_0", "rl_1"]
    x = np.array([veh.get_x_by_id(v) for v in ids])
    assert (np.diff(x) < 0).any()                              # not in driving order any more
    np.testing.assert_allclose(obs[12:24], x / env.k.network.length(), atol=1e-6)
    np.testing.assert_allclose(obs[24:], np.array(veh.get_lane(ids)) / 3.0, atol=1e-6)
    first = x.copy()
    for _ in range(10):
        obs, rew, done, _ = env.step([0.3, 0, -0.2, 0])
    np.testing.assert_allclose(obs[:12], np.array(veh.get_speed(ids)) / 30.0, atol=1e-6)
    env.reset()
    again = np.array([veh.get_x_by_id(v) for v in veh.get_ids()])
    assert sorted(np.round(again, 6)) == sorted(np.round(first, 6)) and not np.allclose(again, first)
    env.terminate()


def test_sim_lane_change_controller_humans_change_lanes_on_a_multilane_ring():
    """a19: SimLaneChangeController vehicles (lane_change_controllers.py:7-16: "SUMO decides") whose
    SumoLaneChangeParams.lane_change_mode allows it ("strategic", core/params.py:20-25) change lanes on their own on a
    multi-lane ring (simplified model ML7); with the default mode "no_lat_collide" nobody moves."""
    from flow_amd.controllers import ContinuousRouter, IDMController, SimLaneChangeController
    from flow_amd.core.params import (EnvParams, InitialConfig, NetParams, SumoCarFollowingParams,
                                      SumoLaneChangeParams, SumoParams, VehicleParams)
    from flow_amd.envs import AccelEnv
    from flow_amd.networks import RingNetwork
    from flow_amd.utils.registry import make_create_env

    def params(mode):
        veh = VehicleParams()
        for name, v0 in (("slow", 6), ("fast", 25)):
            veh.add(veh_id=name, acceleration_controller=(IDMController, {"v0": v0}),
                    lane_change_controller=(SimLaneChangeController, {}), routing_controller=(ContinuousRouter, {}),
                    car_following_params=SumoCarFollowingParams(speed_mode="aggressive"),
                    lane_change_params=SumoLaneChangeParams(lane_change_mode=mode), num_vehicles=6)
        return dict(exp_tag="ring2", env_name=AccelEnv, network=RingNetwork, simulator="traci",
                    sim=SumoParams(sim_step=0.1, render=False),
                    env=EnvParams(horizon=400, additional_params={"max_accel": 3, "max_decel": 3, "target_velocity": 10,
                                                                  "sort_vehicles": False}),
                    net=NetParams(additional_params={"length": 230, "lanes": 2, "speed_limit": 30, "resolution": 40}),
                    veh=veh, initial=InitialConfig(lanes_distribution=1))          # everybody starts in lane 0
    moved = {}
    for mode in ("strategic", "no_lat_collide"):
        env = make_create_env(params(mode))[0]()
        env.reset()
        ids = env.k.vehicle.get_ids()
        assert all(env.k.vehicle.get_lane(v) == 0 for v in ids)
        for _ in range(300):
            _, _, done, _ = env.step(None)
            assert not done
        moved[mode] = [v for v in ids if env.k.vehicle.get_lane(v) != 0]
        env.terminate()
    assert len(moved["strategic"]) > 0 and moved["no_lat_collide"] == []
