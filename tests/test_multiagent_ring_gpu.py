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
