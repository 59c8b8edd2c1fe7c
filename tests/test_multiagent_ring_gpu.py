"""The multi-agent ring environments (flow/envs/multiagent/ring/): per-agent observation blocks written by the step
kernel (heads FS_ENV_ACCEL_PO_MA / FS_ENV_WAVE_ATTENUATION_PO_MA), bit-exact against the oracle's restatement of the
reference's get_state / compute_reward; and the environment classes with the reference's dict interface."""
import numpy as np
import pytest

from helpers import idm_vehicle, ring_spec
from oracle import refsim as S

pytestmark = pytest.mark.gpu


def make(spec, precision="f32"):
    from flow_amd.sim import FlowSim
    return FlowSim(spec, precision=precision)


def ma_spec(env, R=7, N=13, rl_slots=(2, 7, 12), seed=0, **kw):
    rng = np.random.default_rng(seed)
    spec = ring_spec(R=R, N=N, length=200.0, bunching=10, junction_length=0.1, horizon=80, env=env, num_rl=len(rl_slots),
                     action_low=-1.0, action_high=1.0, po_max_length=270.0, target_velocity=20.0, clip_actions=False, **kw)
    spec["init_pos"] = np.asarray(spec["init_pos"]) + np.abs(rng.normal(0, 0.3, (R, N)))
    veh = [idm_vehicle(speed_mode=25, length=5.0 + 0.5 * (i % 3)) for i in range(N)]
    # columns deliberately not in slot order (rl ids are sorted lexicographically in the reference: "rl_10" < "rl_2")
    for col, i in zip(np.random.default_rng(1).permutation(len(rl_slots)), rl_slots):
        veh[i] = idm_vehicle(controller=S.CTRL_RL, rl_index=int(col), speed_mode=25, length=veh[i]["length"])
    spec["vehicles"] = veh
    return spec


@pytest.mark.parametrize("env", [S.ENV_ACCEL_PO_MA, S.ENV_WAVE_ATTENUATION_PO_MA])
@pytest.mark.parametrize("N,rl_slots", [(13, (2, 7, 12)), (22, (0, 21)), (1, (0,))])
def test_multi_agent_ring_heads_bit_exact(env, N, rl_slots):
    K, R = 60, 7
    spec = ma_spec(env, R=R, N=N, rl_slots=rl_slots, seed=N, warmup_steps=5 if N > 1 else 0)
    acts = np.random.default_rng(N).uniform(-1.4, 1.4, (K, R, len(rl_slots))).astype(np.float32)
    sim, ora = make(spec), S.RingOracle(spec, np.float32)
    np.testing.assert_array_equal(sim.reset(), ora.reset().astype(np.float32))
    assert sim.obs_dim == (6 if env == S.ENV_ACCEL_PO_MA else 3) * len(rl_slots)
    for k in range(K):
        o, r, d = sim.step(acts[k])
        o_ref, r_ref, d_ref = ora.step(acts[k])
        np.testing.assert_array_equal(o, o_ref.astype(np.float32), err_msg="obs, step %d" % k)
        np.testing.assert_array_equal(r, r_ref.astype(np.float32), err_msg="reward, step %d" % k)
        np.testing.assert_array_equal(d, d_ref)
    np.testing.assert_array_equal(sim.pos, ora.x)
    sim.close()


def test_multi_agent_heads_ignore_crashes_like_the_reference():
    """multiagent/base.py:188-190 sets crash = 0: a collision neither ends the episode nor zeroes the reward."""
    spec = ma_spec(S.ENV_ACCEL_PO_MA, R=3, N=6, rl_slots=(1,), seed=3)
    spec["ring_length"] = np.full(3, 60.0)
    spec["init_pos"] = np.tile(np.arange(6) * 9.0, (3, 1))
    sim, ora = make(spec), S.RingOracle(spec, np.float32)
    sim.reset(), ora.reset()
    act = np.full((3, 1), 1.0, np.float32)                     # "aggressive" would crash; speed mode 25 keeps it safe:
    spec2 = dict(spec, vehicles=[dict(v, speed_mode=0) for v in spec["vehicles"]])
    sim.close()
    sim, ora = make(spec2), S.RingOracle(spec2, np.float32)
    sim.reset(), ora.reset()
    crashed_once = False
    for k in range(150):
        o, r, d = sim.step(act)
        o_ref, r_ref, d_ref = ora.step(act)
        np.testing.assert_array_equal(r, r_ref.astype(np.float32))
        np.testing.assert_array_equal(d, d_ref)
        crashed_once = crashed_once or bool((ora.headways() < 0).any())
        assert not d.any() or k >= 79                          # done only through the horizon
    assert crashed_once
    sim.close()


def flow_params(env_cls, add_env, n_rl=2, n_human=11, ring_length=None):
    from flow_amd.controllers import ContinuousRouter, IDMController, RLController
    from flow_amd.core.params import EnvParams, InitialConfig, NetParams, SumoParams, VehicleParams
    from flow_amd.networks import RingNetwork
    veh = VehicleParams()
    veh.add(veh_id="human", acceleration_controller=(IDMController, {}), routing_controller=(ContinuousRouter, {}),
            num_vehicles=n_human)
    veh.add(veh_id="rl", acceleration_controller=(RLController, {}), routing_controller=(ContinuousRouter, {}),
            num_vehicles=n_rl)
    return dict(exp_tag="ma_ring", env_name=env_cls, network=RingNetwork, simulator="traci",
                sim=SumoParams(sim_step=0.1, render=False), env=EnvParams(horizon=50, additional_params=add_env),
                net=NetParams(additional_params={"length": 230, "lanes": 1, "speed_limit": 30, "resolution": 40}),
                veh=veh, initial=InitialConfig(bunching=20))


def test_environment_classes_speak_the_reference_dict_interface():
    from flow_amd.envs.multiagent import AdversarialAccelEnv, MultiAgentAccelPOEnv, MultiAgentWaveAttenuationPOEnv
    from flow_amd.utils.registry import make_create_env
    # MultiAgentAccelPOEnv: one 6-vector, one reward per RL vehicle, '__all__' in done
    fp = flow_params(MultiAgentAccelPOEnv, {"max_accel": 1, "max_decel": 1, "target_velocity": 20})
    env = make_create_env(fp)[0]()
    obs = env.reset()
    assert sorted(obs) == ["rl_0", "rl_1"] and all(v.shape == (6,) for v in obs.values())
    assert env.observation_space.shape == (6,) and env.action_space.shape == (1,)
    for _ in range(5):
        obs, rew, done, info = env.step({"rl_0": [0.5], "rl_1": [-0.5]})
    assert set(rew) == {"rl_0", "rl_1"} and rew["rl_0"] == rew["rl_1"] and done["__all__"] is False
    v = env.k.vehicle
    ms, Lr = env.k.network.max_speed(), env.k.network.length()
    lead = v.get_leader("rl_0")
    want = [v.get_x_by_id("rl_0") / Lr, v.get_speed("rl_0") / ms, (v.get_speed(lead) - v.get_speed("rl_0")) / ms,
            (v.get_x_by_id(lead) - v.get_x_by_id("rl_0") - v.get_length("rl_0")) / Lr,
            (v.get_speed("rl_0") - v.get_speed(v.get_follower("rl_0"))) / ms, v.get_headway(v.get_follower("rl_0")) / Lr]
    np.testing.assert_allclose(obs["rl_0"], want, atol=2e-6)
    assert v.get_speed("rl_0") > v.get_speed("rl_1")               # the actions reached their vehicles
    env.terminate()
    # MultiAgentWaveAttenuationPOEnv: three values per agent, a ring length per episode
    fp = flow_params(MultiAgentWaveAttenuationPOEnv, {"max_accel": 1, "max_decel": 1, "ring_length": [220, 270]})
    env = make_create_env(fp)[0]()
    lengths = set()
    for _ in range(4):
        obs = env.reset()
        lengths.add(round(env.k.network.length() if False else float(env.sim.get_state(6)[0])))
    assert len(lengths) > 1 and all(220 <= l <= 270 for l in lengths)
    obs, rew, done, info = env.step({"rl_0": [0.3], "rl_1": [0.1]})
    assert all(o.shape == (3,) for o in obs.values()) and set(rew) == {"rl_0", "rl_1"}
    v = env.k.vehicle
    np.testing.assert_allclose(obs["rl_1"], [v.get_speed("rl_1") / 15, (v.get_speed(v.get_leader("rl_1")) - v.get_speed("rl_1")) / 15,
                                             v.get_headway("rl_1") / 270], atol=2e-6)
    assert env.compute_reward(None) == 0
    env.terminate()
    # AdversarialAccelEnv: AccelEnv's state for both agents, opposite rewards, the adversary perturbs the AV's action
    fp = flow_params(AdversarialAccelEnv, {"max_accel": 3, "max_decel": 3, "target_velocity": 10, "sort_vehicles": False,
                                           "perturb_weight": 0.5}, n_rl=1)
    env = make_create_env(fp)[0]()
    obs = env.reset()
    assert set(obs) == {"av", "adversary"} and obs["av"].shape == (24,)
    v0 = env.k.vehicle.get_speed("rl_0")
    obs, rew, done, info = env.step({"av": np.array([1.0]), "adversary": np.array([1.0])})
    assert rew["av"] == -rew["adversary"] and rew["av"] > 0
    np.testing.assert_allclose(env.k.vehicle.get_speed("rl_0") - v0, 1.5 * 0.1 * (0.1 / 0.101), atol=1e-6)
    np.testing.assert_allclose(obs["av"][0::2], [env.k.vehicle.get_speed(i) / 30 for i in env.k.vehicle.get_ids()], atol=2e-6)
    env.terminate()


@pytest.mark.parametrize("name", ["adversarial_figure_eight", "multiagent_figure_eight"])
def test_reference_multi_agent_figure_eight_experiments_construct_and_step(name):
    """examples/exp_configs/rl/multiagent/{adversarial,multiagent}_figure_eight.py: the flow_params build through
    make_create_env, step with the agents' dicts on the figure eight (segment table + crossing) and, batched, through
    VecFlowEnv."""
    import importlib
    import os
    import sys
    import flow_amd
    flow_amd.install_as_flow()                     # the experiment files import `flow.*` as the reference's do
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from flow_amd.envs import VecFlowEnv
    from flow_amd.utils.registry import make_create_env
    fp = importlib.import_module("exp_configs.rl.multiagent." + name).flow_params
    env = make_create_env(fp)[0]()
    obs = env.reset()
    agents = sorted(obs)
    assert agents == (["adversary", "av"] if name.startswith("adversarial") else ["rl_0_0", "rl_1_0"])
    total = 0.0
    for _ in range(20):
        obs, rew, done, _ = env.step({a: np.array([0.4]) for a in agents})
        total += sum(rew.values())
    assert all(np.isfinite(o).all() for o in obs.values()) and done["__all__"] is False
    if name.startswith("adversarial"):
        assert abs(total) < 1e-6 and obs["av"].shape == (28,)          # the adversary's reward is the negative of the AV's
    else:
        assert all(o.shape == (6,) for o in obs.values()) and rew["rl_0_0"] == rew["rl_1_0"] > 0
    env.terminate()
    vec = VecFlowEnv(fp, num_replicas=16, device=0)
    o = vec.reset()
    assert tuple(o.shape) == (16, vec.obs_dim) and vec.obs_dim == (28 if name.startswith("adversarial") else 12)
    vec.close()


# ---- the multi-agent experiments on the rollout kernels -----------------------------------------
def _ma_rollout(spec, K, acts, env=None):
    from test_parity_gpu import _rollout
    return _rollout(spec, K, acts, env=env)


def ma_ring_experiment_spec(env, R, rl_slots, noise, seed=0, N=22):
    """multiagent_ring.py's population: 22 vehicles, IDM(noise = 0.2, min_gap = 0) humans, RL vehicles spread over the
    ring, clip_actions = False, a ring length per replica."""
    spec = ma_spec(env, R=R, N=N, rl_slots=rl_slots, seed=seed, warmup_steps=0)
    rng = np.random.default_rng(seed + 1)
    spec["ring_length"] = rng.uniform(220.0, 270.0, R)
    spec["init_pos"] = np.sort(rng.uniform(0, 1, (R, N)), axis=1) * 30.0 + np.arange(N) * (spec["ring_length"][:, None] - 31.0) / N
    spec["horizon"] = 60
    for i, v in enumerate(spec["vehicles"]):
        if v["controller"] != S.CTRL_RL:
            v["noise"] = noise
            v["sumo_min_gap"] = 0.0
    spec["seed"] = 11
    return spec


@pytest.mark.parametrize("env,name", [(S.ENV_WAVE_ATTENUATION_PO_MA, "k_ring_pair<POMA>"), (S.ENV_ACCEL_PO_MA, "k_ring_pair<AccelMA>")])
@pytest.mark.parametrize("rl_slots", [(0, 11), (4, 5, 13, 21), (3,)])
def test_multi_agent_ring_rollouts_run_on_the_pair_kernel_bit_exact(env, name, rl_slots):
    """The multi-agent ring heads in k_ring_pair's 16-step group form with one action column per agent: equal to the
    generic k_steps bit for bit (noise included), equal to the oracle without noise; RL vehicles in both halves of a
    lane, next to each other ((4, 5): one lane; (13, ...): B of lane 6; (21, 0)-style wrap through slot 21), columns not
    in slot order; 70 steps = four groups + a remainder, across the horizon; then a second launch mid-block."""
    K, R = 70, 9
    acts = np.random.default_rng(5).uniform(-1.5, 1.5, (K, R, len(rl_slots))).astype(np.float32)
    noisy = ma_ring_experiment_spec(env, R, rl_slots, noise=0.2)
    a, oa, ra, da = _ma_rollout(noisy, K, acts)
    b, ob, rb, db = _ma_rollout(noisy, K, acts, env={"FLOWSIM_FORCE_GENERIC": "1"})
    assert a.last_kernel == name and b.last_kernel.startswith("k_steps")
    np.testing.assert_array_equal(oa, ob)
    np.testing.assert_array_equal(ra, rb)
    np.testing.assert_array_equal(da, db)
    assert not np.isnan(oa).any() and da[59].all() and not da[:59].any()
    from test_parity_gpu import _continue
    o2, r2, _ = _continue(a, 23, acts[:23])
    o3, r3, _ = _continue(b, 23, acts[:23])
    np.testing.assert_array_equal(o2, o3)
    np.testing.assert_array_equal(r2, r3)
    np.testing.assert_array_equal(a.pos, b.pos)
    np.testing.assert_array_equal(a.vel, b.vel)
    a.close(), b.close()
    quiet = ma_ring_experiment_spec(env, R, rl_slots, noise=0.0)
    c, oc, rc, dc = _ma_rollout(quiet, K, acts)
    assert c.last_kernel == name
    ora = S.RingOracle(quiet, np.float32)
    ora.reset()
    for k in range(K):
        o_ref, r_ref, d_ref = ora.step(acts[k])
        np.testing.assert_array_equal(oc[k], o_ref.astype(np.float32), err_msg="obs, step %d" % k)
        np.testing.assert_array_equal(rc[k], r_ref.astype(np.float32), err_msg="reward, step %d" % k)
        np.testing.assert_array_equal(dc[k].astype(bool), d_ref)
    np.testing.assert_array_equal(c.pos, ora.x)
    c.close()


@pytest.mark.parametrize("head", [S.ENV_ACCEL, S.ENV_WAVE_ATTENUATION_PO])
def test_several_action_columns_stay_in_the_group_form_of_the_single_agent_heads(head):
    """AccelEnv / WaveAttenuationPOEnv with three RL vehicles: the rollout keeps k_ring_pair's group form (MC) and its bits."""
    K, R, rl_slots = 50, 6, (2, 9, 16)
    spec = ma_ring_experiment_spec(S.ENV_ACCEL_PO_MA, R, rl_slots, noise=0.2)
    spec["env"] = head
    acts = np.random.default_rng(6).uniform(-1.5, 1.5, (K, R, 3)).astype(np.float32)
    a, oa, ra, da = _ma_rollout(spec, K, acts)
    b, ob, rb, db = _ma_rollout(spec, K, acts, env={"FLOWSIM_FORCE_GENERIC": "1"})
    assert a.last_kernel.startswith("k_ring_pair") and b.last_kernel.startswith("k_steps")
    np.testing.assert_array_equal(oa, ob)
    np.testing.assert_array_equal(ra, rb)
    np.testing.assert_array_equal(da, db)
    np.testing.assert_array_equal(a.pos, b.pos)
    a.close(), b.close()


def test_multi_agent_figure_eight_runs_on_the_loop_rollout_kernel_bit_exact():
    """multiagent_figure_eight.py's population (2 x (6 noisy IDM + 1 RL), obey_safe_speed, MultiAgentAccelPOEnv) on
    k_rollout_loop: observations (12), the shared reward and the state equal the generic k_steps bit for bit; a crash at
    the crossing ends nothing (multiagent/base.py:188-190)."""
    from helpers import figure_eight_spec
    R, N, K = 21, 14, 150
    spec = figure_eight_spec(R=R, N=N, horizon=120, seed=5, num_rl=2, env=S.ENV_ACCEL_PO_MA)
    veh = []
    for g in range(2):
        veh += [idm_vehicle(speed_mode=1, max_decel=1.5, noise=0.2) for _ in range(6)]
        veh.append(idm_vehicle(controller=S.CTRL_RL, rl_index=1 - g, speed_mode=1, max_accel=3.0, max_decel=3.0))
    spec["vehicles"] = veh
    spec["seed"] = 31
    acts = np.random.default_rng(9).uniform(-3, 3, (K, R, 2)).astype(np.float32)
    a, oa, ra, da = _ma_rollout(spec, K, acts)
    b, ob, rb, db = _ma_rollout(spec, K, acts, env={"FLOWSIM_NO_LOOP_KERNEL": "1"})
    f, of, rf, df = _ma_rollout(spec, K, acts, env={"FLOWSIM_NO_LOOP_FULL": "1"})
    assert a.last_kernel == "k_rollout_loop<FULL,AccelMA>" and f.last_kernel == "k_rollout_loop<AccelMA>"
    assert b.last_kernel.startswith("k_steps")
    for o, r, d, sim in ((ob, rb, db, b), (of, rf, df, f)):
        np.testing.assert_array_equal(oa, o)
        np.testing.assert_array_equal(ra, r)
        np.testing.assert_array_equal(da, d)
        np.testing.assert_array_equal(a.pos, sim.pos)
        np.testing.assert_array_equal(a.vel, sim.vel)
    assert oa.shape[2] == 12 and not np.isnan(oa).any() and da.max() == 1
    a.close(), b.close(), f.close()
    hostile_spec = dict(spec, horizon=10 ** 6, vehicles=[dict(v, speed_mode=0) for v in veh])
    hostile = np.full((260, R, 2), 3.0, dtype=np.float32)
    c, oc, rc, dc = _ma_rollout(hostile_spec, 260, hostile)
    e, oe, re_, de = _ma_rollout(hostile_spec, 260, hostile, env={"FLOWSIM_NO_LOOP_KERNEL": "1"})
    np.testing.assert_array_equal(oc, oe)
    np.testing.assert_array_equal(rc, re_)
    assert not dc.any() and not de.any()
    c.close(), e.close()
