"""Multi-agent environments on closed loops (flow/envs/multiagent/ring/*): dict-in / dict-out wrappers whose
observations are assembled on the host from the device state; checked against the oracle run with the same actions."""
import numpy as np
import pytest

from oracle import refsim as S

pytestmark = pytest.mark.gpu


def ring_network(n_human=10, n_rl=2, length=260):
    from flow_amd.controllers import ContinuousRouter, IDMController, RLController
    from flow_amd.core.params import InitialConfig, NetParams, SumoCarFollowingParams, VehicleParams
    from flow_amd.networks.ring import ADDITIONAL_NET_PARAMS, RingNetwork
    v = VehicleParams()
    v.add("human", acceleration_controller=(IDMController, {}), routing_controller=(ContinuousRouter, {}),
          car_following_params=SumoCarFollowingParams(speed_mode="aggressive"), num_vehicles=n_human)
    v.add("rl", acceleration_controller=(RLController, {}), routing_controller=(ContinuousRouter, {}),
          car_following_params=SumoCarFollowingParams(speed_mode="aggressive"), num_vehicles=n_rl)
    add = dict(ADDITIONAL_NET_PARAMS)
    add["length"] = length
    return RingNetwork("ring", v, NetParams(additional_params=add), InitialConfig(bunching=20))


def test_multiagent_accel_po_env():
    """multiagent/ring/accel.py:100-229 (observation space, required parameters: test_environments.py:950-1025)."""
    from flow_amd.core.params import EnvParams, SumoParams
    from flow_amd.envs.multiagent import MultiAgentAccelPOEnv
    net = ring_network()
    full = {"max_accel": 1, "max_decel": 1, "target_velocity": 20}
    for key in full:
        with pytest.raises(KeyError):
            MultiAgentAccelPOEnv(EnvParams(additional_params={k: v for k, v in full.items() if k != key}), SumoParams(),
                                 net)
    env = MultiAgentAccelPOEnv(EnvParams(horizon=100, additional_params=full), SumoParams(sim_step=0.1), net)
    assert env.observation_space.shape == (6,) and env.observation_space.low[0] == -5
    assert env.action_space.shape == (1,) and env.action_space.high[0] == 1
    ora = S.RingOracle(env._spec, np.float64)
    obs = env.reset()
    ora.reset()
    assert set(obs) == {"rl_0", "rl_1"}
    L_ = env.k.network.length()
    rng = np.random.default_rng(0)
    for k in range(100):
        acts = {rl: np.array([rng.uniform(-1, 1)]) for rl in obs}
        obs, rew, done, _ = env.step(acts)
        ora.step(np.array([[float(acts["rl_0"][0]), float(acts["rl_1"][0])]]))
        x, v = ora.x[0], ora.v[0]
        h = ora.headways()[0]
        for a, rl in enumerate(("rl_0", "rl_1")):
            i, n = 10 + a, 12
            lead, foll = (i + 1) % n, (i - 1) % n
            want = [x[i] / L_, v[i] / 30, (v[lead] - v[i]) / 30, (x[lead] - x[i] - 5) / L_, (v[i] - v[foll]) / 30,
                    h[foll] / L_]
            np.testing.assert_allclose(obs[rl], want, rtol=0, atol=2e-5)
        assert set(rew) == {"rl_0", "rl_1"} and abs(rew["rl_0"] - rew["rl_1"]) == 0
        assert done["__all__"] == (k == 99)
    env.additional_command()
    assert set(env.k.vehicle.get_observed_ids()) >= {"human_9", "rl_1", "rl_0", "human_0"}
    env.terminate()


def test_multiagent_wave_attenuation_po_env():
    from flow_amd.core.params import EnvParams, SumoParams
    from flow_amd.envs.multiagent import MultiAgentWaveAttenuationPOEnv
    net = ring_network(n_human=19, n_rl=2)
    with pytest.raises(KeyError):
        MultiAgentWaveAttenuationPOEnv(EnvParams(additional_params={"max_accel": 1, "max_decel": 1}), SumoParams(), net)
    add = {"max_accel": 1, "max_decel": 1, "ring_length": [230, 230]}
    env = MultiAgentWaveAttenuationPOEnv(EnvParams(horizon=60, warmup_steps=10, additional_params=add),
                                         SumoParams(sim_step=0.1), net)
    obs = env.reset()
    assert set(obs) == {"rl_0", "rl_1"} and env.k.network.length() == pytest.approx(230.4)
    total = None
    for k in range(60):
        acts = {rl: 0.3 for rl in obs}
        obs, rew, done, _ = env.step(acts)
        v = np.array(env.k.vehicle.get_speed(env.k.vehicle.get_ids()))
        want = 4.0 * np.mean(v) / 20 + 4 * (0 - 0.3)
        assert rew == {"rl_0": pytest.approx(want), "rl_1": pytest.approx(want)}
        for rl in obs:
            lead = env.k.vehicle.get_leader(rl)
            np.testing.assert_allclose(obs[rl], [env.k.vehicle.get_speed(rl) / 15,
                                                 (env.k.vehicle.get_speed(lead) - env.k.vehicle.get_speed(rl)) / 15,
                                                 env.k.vehicle.get_headway(rl) / 230])
    assert done["__all__"] and env.compute_reward(None, fail=False) == 0
    env.terminate()


def test_adversarial_accel_env():
    from flow_amd.core.params import EnvParams, SumoParams
    from flow_amd.envs.multiagent import AdversarialAccelEnv
    net = ring_network(n_human=12, n_rl=2)
    add = {"max_accel": 3, "max_decel": 3, "target_velocity": 10, "sort_vehicles": False, "perturb_weight": 0.5}
    env = AdversarialAccelEnv(EnvParams(horizon=50, additional_params=add), SumoParams(sim_step=0.1), net)
    ora = S.RingOracle(env._spec, np.float32)
    obs = env.reset()
    ora.reset()
    assert set(obs) == {"av", "adversary"} and obs["av"].shape == (28,)
    rng = np.random.default_rng(1)
    for k in range(50):
        av, adv = rng.uniform(-1, 1, 2), rng.uniform(-1, 1, 2)
        obs, rew, done, _ = env.step({"av": av, "adversary": adv})
        ora.step((av + 0.5 * adv)[None, :].astype(np.float32))
        assert rew["av"] == -rew["adversary"] and rew["av"] >= 0
        np.testing.assert_array_equal(obs["av"], obs["adversary"])
    # interleaved [v_i / v_max, x_i / L] per vehicle (np.ndarray.flatten of the N x 2 array)
    np.testing.assert_allclose(obs["av"][0::2], ora.v[0] / 30, rtol=0, atol=1e-6)
    np.testing.assert_allclose(obs["av"][1::2], ora.x[0] / env.k.network.length(), rtol=0, atol=1e-6)
    env.terminate()


def lord_of_the_rings(num_rings=3, noise=0.0, horizon=120, warmup=20):
    """examples/exp_configs/rl/multiagent/lord_of_the_rings.py:30-104 (NUM_RINGS rings of 21 IDM + 1 RL vehicle)."""
    from flow_amd.controllers import ContinuousRouter, IDMController, RLController
    from flow_amd.core.params import EnvParams, InitialConfig, NetParams, SumoParams, VehicleParams
    from flow_amd.envs.multiagent import MultiWaveAttenuationPOEnv
    from flow_amd.networks import MultiRingNetwork
    vehicles = VehicleParams()
    for i in range(num_rings):
        vehicles.add(veh_id='human_{}'.format(i), acceleration_controller=(IDMController, {'noise': noise}),
                     routing_controller=(ContinuousRouter, {}), num_vehicles=21)
        vehicles.add(veh_id='rl_{}'.format(i), acceleration_controller=(RLController, {}),
                     routing_controller=(ContinuousRouter, {}), num_vehicles=1)
    return dict(exp_tag='lord_of_numrings{}'.format(num_rings), env_name=MultiWaveAttenuationPOEnv,
                network=MultiRingNetwork, simulator='traci', sim=SumoParams(sim_step=0.1, render=False),
                env=EnvParams(horizon=horizon, warmup_steps=warmup,
                              additional_params={'max_accel': 1, 'max_decel': 1, 'ring_length': [230, 230],
                                                 'target_velocity': 4}),
                net=NetParams(additional_params={'length': 230, 'lanes': 1, 'speed_limit': 30, 'resolution': 40,
                                                 'num_rings': num_rings}),
                veh=vehicles, initial=InitialConfig(bunching=20.0, spacing='custom'))


def test_multi_wave_attenuation_po_env_on_the_lord_of_the_rings_network():
    """multiagent/ring/wave_attenuation.py:34-127 on MultiRingNetwork: ring r runs as replica r of the ring kernel; the
    per-agent observation and reward are the reference's formulas over that state, checked against the oracle."""
    from flow_amd.utils.registry import make_create_env
    K = 3
    env = make_create_env(lord_of_the_rings(K))[0]()
    assert env.observation_space.shape == (3,) and env.action_space.shape == (1,)
    spec = env._spec
    assert spec["num_replicas"] == K and spec["num_vehicles"] == 22 and spec["num_rl"] == 1
    # custom placement (multi_ring.py:98-150): every ring starts at its own x = 0; the spacing comes from the total length
    inc = (K * 230 - 20 - 5 * 22 * K) / (22 * K) + 5
    np.testing.assert_allclose(spec["init_pos"][1][:3], [0, inc, 2 * inc], atol=1e-9)
    net = env.k.network
    assert net.get_edge_list()[:5] == ["bottom_0", "right_0", "top_0", "left_0", "bottom_1"]
    assert net.get_x("right_2", 1.5) == 2 * 230 + 57.5 + 1.5
    ora = S.RingOracle(spec, np.float64)
    obs = env.reset()
    ora.reset()
    rl_ids = ["rl_{}_0".format(r) for r in range(K)]
    assert sorted(obs) == rl_ids and env.k.vehicle.get_rl_ids() == rl_ids
    veh = env.k.vehicle
    assert veh.get_leader("rl_1_0") == "human_1_0" and veh.get_follower("human_2_0") == "rl_2_0"
    assert veh.get_edge("human_1_0") == "bottom_1" and veh.get_edge("rl_2_0").endswith("_2")
    rng = np.random.default_rng(0)
    for k in range(120):
        acts = {rl: np.array([rng.uniform(-1, 1)]) for rl in rl_ids}
        obs, rew, done, _ = env.step(acts)
        ora.step(np.array([[float(acts[rl][0])] for rl in rl_ids]))
        h = ora.headways()
        for r, rl in enumerate(rl_ids):
            x, v = ora.x[r], ora.v[r]
            want = [v[21] / 15, (v[0] - v[21]) / 15, h[r][21] / 230]
            np.testing.assert_allclose(obs[rl], want, rtol=0, atol=2e-5)
            # reward: the vehicles of ring r that are on one of its four edges (not inside a 0.1 m junction)
            quarter = 57.5 + 0.1
            on_edge = (x % quarter) < 57.5
            vel = v[on_edge]
            mx = np.linalg.norm(np.full(len(vel), 4.0))
            np.testing.assert_allclose(rew[rl], max(mx - np.linalg.norm(vel - 4.0), 0) / mx, rtol=0, atol=2e-5)
        assert done["__all__"] == (k == 119) and not any(done[rl] for rl in rl_ids)
    env.additional_command()
    assert set(veh.get_observed_ids()) >= {"human_0_0", "human_1_0", "human_2_0"}
    assert len(veh.get_ids_by_edge(env.gen_edges("1"))) in (21, 22)
    env.terminate()


def test_multi_ring_requires_equal_rings_and_the_net_params():
    from flow_amd.controllers import IDMController, RLController
    from flow_amd.core.params import EnvParams, NetParams, SumoParams, VehicleParams
    from flow_amd.envs.multiagent import MultiWaveAttenuationPOEnv
    from flow_amd.networks import MultiRingNetwork
    from flow_amd.networks.multi_ring import ADDITIONAL_NET_PARAMS
    v = VehicleParams()
    v.add("human", acceleration_controller=(IDMController, {}), num_vehicles=5)
    for key in ADDITIONAL_NET_PARAMS:
        with pytest.raises(KeyError):
            MultiRingNetwork("m", v, NetParams(additional_params={k: x for k, x in ADDITIONAL_NET_PARAMS.items() if k != key}))
    uneven = VehicleParams()
    uneven.add("human_0", acceleration_controller=(IDMController, {}), num_vehicles=4)
    uneven.add("rl_0", acceleration_controller=(RLController, {}), num_vehicles=1)
    uneven.add("rl_1", acceleration_controller=(RLController, {}), num_vehicles=1)
    uneven.add("human_1", acceleration_controller=(IDMController, {}), num_vehicles=4)
    add = dict(ADDITIONAL_NET_PARAMS, num_rings=2)
    env_params = EnvParams(additional_params={'max_accel': 1, 'max_decel': 1, 'ring_length': [230, 230],
                                              'target_velocity': 4})
    with pytest.raises(NotImplementedError, match="same vehicle types in the same order"):
        MultiWaveAttenuationPOEnv(env_params, SumoParams(), MultiRingNetwork("m", uneven, NetParams(additional_params=add)))


def test_lord_of_the_rings_example_file_runs_with_seven_rings():
    import importlib
    import os
    import sys
    import flow_amd
    flow_amd.install_as_flow()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "examples"))
    try:
        mod = importlib.import_module("exp_configs.rl.lord_of_the_rings")
    finally:
        sys.path.pop(0)
    from flow_amd.utils.registry import make_create_env
    params = dict(mod.flow_params)
    params["env"].warmup_steps = 30
    env = make_create_env(params)[0]()
    obs = env.reset()
    assert len(obs) == 7 and env._spec["num_replicas"] == 7
    for _ in range(20):
        obs, rew, done, _ = env.step({rl: np.array([0.3]) for rl in obs})
    assert len(rew) == 7 and all(0 <= r <= 1 for r in rew.values()) and not done["__all__"]
    speeds = env.k.vehicle.get_speed(env.k.vehicle.get_ids())
    assert len(speeds) == 154 and min(speeds) >= 0
    env.terminate()


def fig8_vehicles(groups, noise=0.0):
    """[(n_human, n_rl)] -> VehicleParams in the layout of examples/exp_configs/rl/multiagent/*figure_eight.py."""
    from flow_amd.controllers import ContinuousRouter, IDMController, RLController
    from flow_amd.core.params import SumoCarFollowingParams, VehicleParams
    v = VehicleParams()
    for i, (n_h, n_rl) in enumerate(groups):
        tag = "_{}".format(i) if len(groups) > 1 else ""
        v.add(veh_id="human" + tag, acceleration_controller=(IDMController, {"noise": noise}),
              routing_controller=(ContinuousRouter, {}),
              car_following_params=SumoCarFollowingParams(speed_mode="obey_safe_speed", decel=1.5), num_vehicles=n_h)
        v.add(veh_id="rl" + tag, acceleration_controller=(RLController, {}), routing_controller=(ContinuousRouter, {}),
              car_following_params=SumoCarFollowingParams(speed_mode="obey_safe_speed", accel=3, decel=3),
              num_vehicles=n_rl)
    return v


def test_multiagent_envs_on_the_figure_eight_examples():
    """multiagent_figure_eight.py (MultiAgentAccelPOEnv, 2 x (6 humans + 1 RL)) and adversarial_figure_eight.py
    (AdversarialAccelEnv, 13 humans + 1 RL) on FigureEightNetwork: trajectories against the oracle, observation
    formulas over that state."""
    from flow_amd.core.params import EnvParams, InitialConfig, NetParams, SumoParams
    from flow_amd.envs.multiagent import AdversarialAccelEnv, MultiAgentAccelPOEnv
    from flow_amd.networks import FigureEightNetwork
    from flow_amd.networks.figure_eight import ADDITIONAL_NET_PARAMS
    add = {"target_velocity": 20, "max_accel": 3, "max_decel": 3, "sort_vehicles": False}
    net = FigureEightNetwork("fig8", fig8_vehicles([(6, 1), (6, 1)]), NetParams(additional_params=dict(ADDITIONAL_NET_PARAMS)),
                             InitialConfig())
    env = MultiAgentAccelPOEnv(EnvParams(horizon=150, additional_params=add), SumoParams(sim_step=0.1), net)
    ora = S.RingOracle(env._spec, np.float64)
    obs = env.reset()
    ora.reset()
    assert sorted(obs) == ["rl_0_0", "rl_1_0"] and env.observation_space.shape == (6,)
    rng = np.random.default_rng(3)
    for k in range(150):
        acts = {rl: np.array([rng.uniform(-3, 3)]) for rl in obs}
        obs, rew, done, _ = env.step(acts)
        ora.step(np.array([[float(acts["rl_0_0"][0]), float(acts["rl_1_0"][0])]]))
    ids = env.k.vehicle.get_ids()
    np.testing.assert_allclose(env.k.vehicle.get_speed(ids), ora.v[0], rtol=0, atol=2e-4)
    for rl, i in (("rl_0_0", 6), ("rl_1_0", 13)):
        assert abs(obs[rl][1] - ora.v[0][i] / env.k.network.max_speed()) < 2e-5
    assert set(rew) == {"rl_0_0", "rl_1_0"} and done["__all__"]
    env.terminate()

    net = FigureEightNetwork("fig8", fig8_vehicles([(13, 1)]), NetParams(additional_params=dict(ADDITIONAL_NET_PARAMS)),
                             InitialConfig())
    env = AdversarialAccelEnv(EnvParams(horizon=100, additional_params=dict(add, perturb_weight=0.03)),
                              SumoParams(sim_step=0.1), net)
    ora = S.RingOracle(env._spec, np.float64)
    obs = env.reset()
    ora.reset()
    assert sorted(obs) == ["adversary", "av"] and obs["av"].shape == (28,)
    for k in range(100):
        av, adv = rng.uniform(-3, 3, 1), rng.uniform(-3, 3, 1)
        obs, rew, done, _ = env.step({"av": av, "adversary": adv})
        ora.step(np.array([np.float32(av + 0.03 * adv)]))
    np.testing.assert_allclose(obs["av"][0::2], ora.v[0] / env.k.network.max_speed(), rtol=0, atol=2e-5)
    assert rew["av"] == -rew["adversary"] and 0 < rew["av"] < 1
    env.terminate()
