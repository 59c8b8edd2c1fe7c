"""The reference's benchmark configurations that fall on the built path (flow/benchmarks/{figureeight0-2, merge0-2,
bottleneck0-2}.py; their flow_params restated here from those files): each is constructed through make_create_env,
stepped with random actions and compared with the oracle run on the env's own spec.  bottleneck2 (scaling = 2:
8 -> 4 -> 2 lanes) needs more than 64 vehicle slots per replica (SumoParams(max_vehicles=...))."""
import numpy as np
import pytest

from oracle import opennet as O
from oracle import refsim as S

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def slot_order_kernels(monkeypatch):
    """This module holds the SLOT-order open-network kernels (k_steps_open / k_steps_wide) to the oracle; the queue-order
    kernels that take the same configurations by default have tests of their own (test_queue_gpu.py, test_dropq_gpu.py)."""
    monkeypatch.setenv("FLOWSIM_NO_QUEUE", "1")


def make_env(flow_params):
    from flow_amd.utils.registry import make_create_env
    return make_create_env(flow_params)[0]()


def figure_eight_benchmark(k, noise):
    """figureeight0: 13 humans + 1 RL; figureeight1: 7 x (1 human, 1 RL); figureeight2: 14 RL (:21-75)."""
    from flow_amd.controllers import ContinuousRouter, IDMController, RLController
    from flow_amd.core.params import EnvParams, InitialConfig, NetParams, SumoCarFollowingParams, SumoParams, VehicleParams
    from flow_amd.envs import AccelEnv
    from flow_amd.networks import FigureEightNetwork
    from flow_amd.networks.figure_eight import ADDITIONAL_NET_PARAMS
    human = dict(acceleration_controller=(IDMController, {"noise": noise}), routing_controller=(ContinuousRouter, {}),
                 car_following_params=SumoCarFollowingParams(speed_mode="obey_safe_speed", decel=1.5))
    rl = dict(acceleration_controller=(RLController, {}), routing_controller=(ContinuousRouter, {}),
              car_following_params=SumoCarFollowingParams(speed_mode="obey_safe_speed"))
    vehicles = VehicleParams()
    if k == 0:
        vehicles.add(veh_id="human", num_vehicles=13, **human)
        vehicles.add(veh_id="rl", num_vehicles=1, **rl)
    elif k == 1:
        for i in range(7):
            vehicles.add(veh_id="human{}".format(i), num_vehicles=1, **human)
            vehicles.add(veh_id="rl{}".format(i), num_vehicles=1, **rl)
    else:
        vehicles.add(veh_id="rl", num_vehicles=14, **rl)
    return dict(exp_tag="figure_eight_{}".format(k), env_name=AccelEnv, network=FigureEightNetwork, simulator='traci',
                sim=SumoParams(sim_step=0.1, render=False),
                env=EnvParams(horizon=1500, additional_params={"target_velocity": 20, "max_accel": 3, "max_decel": 3,
                                                               "sort_vehicles": False}),
                net=NetParams(additional_params=dict(ADDITIONAL_NET_PARAMS)), veh=vehicles, initial=InitialConfig())


@pytest.mark.parametrize("k", [0, 1, 2])
def test_figure_eight_benchmarks(k):
    env = make_env(figure_eight_benchmark(k, noise=0.0))
    n_rl = (1, 7, 14)[k]
    assert env.action_space.shape == (n_rl,) and env.observation_space.shape == (28,)
    ora = S.RingOracle(env._spec, np.float32)
    np.testing.assert_array_equal(env.reset(), ora.reset()[0].astype(np.float32))
    rng = np.random.default_rng(k)
    for _ in range(200):
        a = rng.uniform(-3, 3, n_rl).astype(np.float32)
        obs, rew, done, _ = env.step(a)
        o_ref, r_ref, d_ref = ora.step(a[None, :])
        np.testing.assert_array_equal(obs, o_ref[0].astype(np.float32))
        assert rew == np.float32(r_ref[0]) and done == bool(d_ref[0])
    env.terminate()
    noisy = make_env(figure_eight_benchmark(k, noise=0.2))             # as shipped: runs, finite, stays on the loop
    noisy.reset()
    for _ in range(50):
        obs, rew, done, _ = noisy.step(rng.uniform(-1, 1, n_rl))
    assert np.isfinite(obs).all() and 0 <= obs.min() and obs.max() <= 1.0 + 1e-6
    noisy.terminate()


def merge_benchmark(k):
    """merge0 / merge1 / merge2: 10 % / 25 % / 33.3 % of the 2000 veh/h highway inflow are RL vehicles, 5 / 13 / 17
    controlled places; humans and RL vehicles on the SUMO car-following model with speed mode 9 (:21-120)."""
    from flow_amd.controllers import RLController, SimCarFollowingController
    from flow_amd.core.params import (EnvParams, InFlows, InitialConfig, NetParams, SumoCarFollowingParams, SumoParams,
                                      VehicleParams)
    from flow_amd.envs import MergePOEnv
    from flow_amd.networks import MergeNetwork
    from flow_amd.networks.merge import ADDITIONAL_NET_PARAMS
    penetration, num_rl = ((0.1, 5), (0.25, 13), (0.333, 17))[k]
    add = dict(ADDITIONAL_NET_PARAMS)
    add.update(merge_lanes=1, highway_lanes=1, pre_merge_length=500)
    vehicles = VehicleParams()
    vehicles.add(veh_id="human", acceleration_controller=(SimCarFollowingController, {}),
                 car_following_params=SumoCarFollowingParams(speed_mode=9), num_vehicles=5)
    vehicles.add(veh_id="rl", acceleration_controller=(RLController, {}),
                 car_following_params=SumoCarFollowingParams(speed_mode=9), num_vehicles=0)
    inflow = InFlows()
    inflow.add(veh_type="human", edge="inflow_highway", vehs_per_hour=(1 - penetration) * 2000, departLane="free",
               departSpeed=10)
    inflow.add(veh_type="rl", edge="inflow_highway", vehs_per_hour=penetration * 2000, departLane="free", departSpeed=10)
    inflow.add(veh_type="human", edge="inflow_merge", vehs_per_hour=100, departLane="free", departSpeed=7.5)
    return dict(exp_tag="merge_{}".format(k), env_name=MergePOEnv, network=MergeNetwork, simulator='traci',
                sim=SumoParams(restart_instance=True, sim_step=0.5, render=False),
                env=EnvParams(horizon=750, sims_per_step=2, warmup_steps=0,
                              additional_params={"max_accel": 1.5, "max_decel": 1.5, "target_velocity": 20,
                                                 "num_rl": num_rl}),
                net=NetParams(inflows=inflow, additional_params=add), veh=vehicles, initial=InitialConfig())


@pytest.mark.parametrize("k", [0, 1, 2])
def test_merge_benchmarks(k):
    env = make_env(merge_benchmark(k))
    num_rl = (5, 13, 17)[k]
    assert env.observation_space.shape == (5 * num_rl,) and env.action_space.shape == (num_rl,)
    ora = O.MergeOracle(env._spec, np.float32)
    np.testing.assert_array_equal(env.reset(), ora.reset()[0].astype(np.float32))
    rng = np.random.default_rng(10 + k)
    for _ in range(300):
        a = rng.uniform(-1.5, 1.5, num_rl).astype(np.float32)
        obs, rew, done, _ = env.step(a)
        o_ref, r_ref, d_ref = ora.step(a[None, :])
        np.testing.assert_array_equal(obs, o_ref[0].astype(np.float32))
        assert rew == np.float32(r_ref[0]) and done == bool(d_ref[0])
    assert len(env.rl_veh) >= 1 and len(env.k.vehicle.get_ids()) > 15
    env.terminate()


def bottleneck_benchmark(k, **sim_kw):
    """bottleneck0: scaling 1, 10 % AVs, no lane changes; bottleneck1: scaling 1, 25 % AVs, humans change lanes
    (lane_change_mode 1621); bottleneck2: scaling 2, 10 % AVs (:20-150)."""
    from flow_amd.controllers import ContinuousRouter, RLController
    from flow_amd.core.params import (EnvParams, InFlows, InitialConfig, NetParams, SumoCarFollowingParams,
                                      SumoLaneChangeParams, SumoParams, VehicleParams)
    from flow_amd.envs import BottleneckDesiredVelocityEnv
    from flow_amd.networks import BottleneckNetwork
    scaling, av_frac, human_lc = ((1, 0.10, 0), (1, 0.25, 1621), (2, 0.10, 0))[k]
    vehicles = VehicleParams()
    vehicles.add(veh_id="human", routing_controller=(ContinuousRouter, {}),
                 car_following_params=SumoCarFollowingParams(speed_mode=9),
                 lane_change_params=SumoLaneChangeParams(lane_change_mode=human_lc), num_vehicles=1 * scaling)
    vehicles.add(veh_id="rl", acceleration_controller=(RLController, {}), routing_controller=(ContinuousRouter, {}),
                 car_following_params=SumoCarFollowingParams(speed_mode=9),
                 lane_change_params=SumoLaneChangeParams(lane_change_mode=0), num_vehicles=1 * scaling)
    add = {"target_velocity": 40, "disable_tb": True, "disable_ramp_metering": True,
           "controlled_segments": [("1", 1, False), ("2", 2, True), ("3", 2, True), ("4", 2, True), ("5", 1, False)],
           "symmetric": False, "observed_segments": [("1", 1), ("2", 3), ("3", 3), ("4", 3), ("5", 1)],
           "reset_inflow": False, "lane_change_duration": 5, "max_accel": 3, "max_decel": 3,
           "inflow_range": [1200 * scaling, 2500 * scaling]}
    flow_rate = 2000 * scaling
    inflow = InFlows()
    inflow.add(veh_type="human", edge="1", vehs_per_hour=flow_rate * (1 - av_frac), departLane="random", departSpeed=10)
    inflow.add(veh_type="rl", edge="1", vehs_per_hour=flow_rate * av_frac, departLane="random", departSpeed=10)
    return dict(exp_tag="bottleneck_{}".format(k), env_name=BottleneckDesiredVelocityEnv, network=BottleneckNetwork,
                simulator='traci', sim=SumoParams(sim_step=0.5, render=False, print_warnings=False,
                                                  restart_instance=True, **sim_kw),
                env=EnvParams(warmup_steps=40, sims_per_step=1, horizon=1500, additional_params=add),
                net=NetParams(inflows=inflow, additional_params={"scaling": scaling, "speed_limit": 23}), veh=vehicles,
                initial=InitialConfig(spacing="uniform", min_gap=5, lanes_distribution=float("inf"),
                                      edges_distribution=["2", "3", "4", "5"]))


@pytest.mark.parametrize("k,slots", [(0, 64), pytest.param(1, 64, marks=pytest.mark.slow), pytest.param(0, 160, marks=pytest.mark.slow),
                                     pytest.param(1, 160, marks=pytest.mark.slow)])
def test_bottleneck_benchmarks(k, slots):
    env = make_env(bottleneck_benchmark(k, max_vehicles=slots))
    assert env.observation_space.shape == (141,) and env.action_space.shape == (20,)
    assert env._spec["num_vehicles"] == slots
    ora = O.MergeOracle(env._spec, np.float32)
    np.testing.assert_array_equal(env.reset(), ora.reset()[0].astype(np.float32))
    rng = np.random.default_rng(20 + k)
    for _ in range(400):
        a = rng.uniform(-1.5, 1.5, 20).astype(np.float32)
        obs, rew, done, _ = env.step(a)
        o_ref, r_ref, d_ref = ora.step(a[None, :])
        np.testing.assert_array_equal(obs, o_ref[0].astype(np.float32))
        assert rew == np.float32(r_ref[0]) and done == bool(d_ref[0])
    if k == 1:
        assert int(ora.num_lane_changes[0]) > 10                  # the simplified lane-change model was at work
    assert len(env.k.vehicle.get_ids()) == int(ora.alive[0].sum()) > 30
    env.terminate()


@pytest.mark.slow            # (scaling 2 stays in the fast set through tests/test_wide_gpu.py::test_wide_scaling_two_eight_entry_lanes)
def test_bottleneck2_benchmark_scaling_two():
    with pytest.raises(NotImplementedError, match="above 64"):
        make_env(bottleneck_benchmark(2))                                  # the default pool of 64 slots
    env = make_env(bottleneck_benchmark(2, max_vehicles=256))
    assert env.observation_space.shape == (4 * 70 + 1,) and env.action_space.shape == (40,)
    assert env._spec["num_paths"] == 8 and env._spec["scaling"] == 2
    ora = O.MergeOracle(env._spec, np.float32)
    np.testing.assert_array_equal(env.reset(), ora.reset()[0].astype(np.float32))
    rng = np.random.default_rng(22)
    for _ in range(220):
        a = rng.uniform(-1.5, 1.5, 40).astype(np.float32)
        obs, rew, done, _ = env.step(a)
        o_ref, r_ref, d_ref = ora.step(a[None, :])
        np.testing.assert_array_equal(obs, o_ref[0].astype(np.float32))
        assert rew == np.float32(r_ref[0]) and done == bool(d_ref[0])
    veh = env.k.vehicle
    ids = veh.get_ids()
    assert len(ids) == int(ora.alive[0].sum()) > 60
    lanes = {e: set() for e in "12345"}
    for v in ids:
        if veh.get_edge(v) in lanes:
            lanes[veh.get_edge(v)].add(veh.get_lane(v))
    assert max(lanes["2"]) >= 6 and max(lanes["4"]) <= 3 and max(lanes["5"] | {0}) <= 1
    env.terminate()
