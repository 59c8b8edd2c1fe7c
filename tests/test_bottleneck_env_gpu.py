"""The reference-facing classes of the lane-drop row (BottleneckNetwork, BottleneckEnv,
BottleneckDesiredVelocityEnv) on the GPU step loop, against the oracle and the reference's own tests
(tests/fast_tests/test_environments.py:739-947)."""
import numpy as np
import pytest

from oracle import opennet as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def slot_order_kernels(monkeypatch):
    """This module holds the SLOT-order open-network kernels (k_steps_open / k_steps_wide) to the oracle; the queue-order
    kernels that take the same configurations by default have tests of their own (test_queue_gpu.py, test_dropq_gpu.py)."""
    monkeypatch.setenv("FLOWSIM_NO_QUEUE", "1")


def c4_flow_params(horizon=1000, warmup_steps=40, reset_inflow=False, flow_rate=2300, **sim_kw):
    """examples/exp_configs/rl/singleagent/singleagent_bottleneck.py:28-151."""
    from flow_amd.controllers import ContinuousRouter, RLController, SimLaneChangeController
    from flow_amd.core.params import (EnvParams, InFlows, InitialConfig, NetParams, SumoCarFollowingParams,
                                      SumoLaneChangeParams, SumoParams, VehicleParams)
    from flow_amd.envs import BottleneckDesiredVelocityEnv
    from flow_amd.networks import BottleneckNetwork
    vehicles = VehicleParams()
    vehicles.add(veh_id="human", lane_change_controller=(SimLaneChangeController, {}),
                 routing_controller=(ContinuousRouter, {}),
                 car_following_params=SumoCarFollowingParams(speed_mode="all_checks"),
                 lane_change_params=SumoLaneChangeParams(lane_change_mode=0), num_vehicles=1)
    vehicles.add(veh_id="followerstopper", acceleration_controller=(RLController, {}),
                 lane_change_controller=(SimLaneChangeController, {}), routing_controller=(ContinuousRouter, {}),
                 car_following_params=SumoCarFollowingParams(speed_mode=9),
                 lane_change_params=SumoLaneChangeParams(lane_change_mode=0), num_vehicles=1)
    add = {"target_velocity": 40, "disable_tb": True, "disable_ramp_metering": True,
           "controlled_segments": [("1", 1, False), ("2", 2, True), ("3", 2, True), ("4", 2, True), ("5", 1, False)],
           "symmetric": False, "observed_segments": [("1", 1), ("2", 3), ("3", 3), ("4", 3), ("5", 1)],
           "reset_inflow": reset_inflow, "lane_change_duration": 5, "max_accel": 3, "max_decel": 3,
           "inflow_range": [1000, 2000]}
    inflow = InFlows()
    inflow.add(veh_type="human", edge="1", vehs_per_hour=flow_rate * 0.9, departLane="random", departSpeed=10)
    inflow.add(veh_type="followerstopper", edge="1", vehs_per_hour=flow_rate * 0.1, departLane="random",
               departSpeed=10)
    return dict(exp_tag="DesiredVelocity", env_name=BottleneckDesiredVelocityEnv, network=BottleneckNetwork,
                simulator='traci',
                sim=SumoParams(sim_step=0.5, render=False, print_warnings=False, restart_instance=True, **sim_kw),
                env=EnvParams(warmup_steps=warmup_steps, sims_per_step=1, horizon=horizon, additional_params=add),
                net=NetParams(inflows=inflow, additional_params={"scaling": 1, "speed_limit": 23}), veh=vehicles,
                initial=InitialConfig(spacing="uniform", min_gap=5, lanes_distribution=float("inf"),
                                      edges_distribution=["2", "3", "4", "5"]))


def make_env(flow_params):
    from flow_amd.utils.registry import make_create_env
    return make_create_env(flow_params)[0]()


def test_reference_bottleneck_env_tests():
    """test_environments.py:739-810: ten SUMO-driven vehicles, dummy spaces, nobody on edges 3 / 4 after reset."""
    from flow_amd.core.params import EnvParams, NetParams, SumoParams, VehicleParams
    from flow_amd.envs import BottleneckAccelEnv, BottleneckEnv
    from flow_amd.networks import BottleneckNetwork
    vehicles = VehicleParams()
    vehicles.add(veh_id="human", num_vehicles=10)
    full = {"max_accel": 3, "max_decel": 3, "lane_change_duration": 5, "disable_tb": True,
            "disable_ramp_metering": True}
    net = BottleneckNetwork(name="bay_bridge_toll", vehicles=vehicles,
                            net_params=NetParams(additional_params={"scaling": 1, "speed_limit": 23}))
    sim_params = SumoParams(sim_step=0.5, restart_instance=True)
    for key in full:
        with pytest.raises(KeyError):
            BottleneckEnv(EnvParams(additional_params={k: v for k, v in full.items() if k != key}), sim_params, net)
    env = BottleneckEnv(EnvParams(additional_params=full), sim_params, net)
    env.reset()
    assert env.get_bottleneck_density() == 0
    for space in (env.observation_space, env.action_space):
        assert space.shape == (1,) and space.low[0] == -float('inf') and space.high[0] == float('inf')
    obs, rew, done, _ = env.step(None)
    assert list(obs) == [1] and rew >= 0.0 and not done
    env.additional_command()
    assert sum(len(lane) for lane in env.edge_dict["1"]) + sum(len(lane) for lane in env.edge_dict["2"]) >= 8
    env.terminate()
    with pytest.raises(NotImplementedError):
        BottleneckEnv(EnvParams(additional_params=dict(full, disable_tb=False)), sim_params, net)


def test_reference_bottleneck_accel_env_tests():
    """test_environments.py:813-878: ten humans, no RL vehicle: required params, observation space of 12 in [0, 1]
    (two numbers for each of the six edges, the rendering-only fake_edge included), an empty action space; the
    observation and the reward are the reference's formulas over the device state."""
    from flow_amd.core.params import EnvParams, NetParams, SumoParams, VehicleParams
    from flow_amd.controllers import RLController
    from flow_amd.envs import BottleneckAccelEnv
    from flow_amd.networks import BottleneckNetwork
    vehicles = VehicleParams()
    vehicles.add(veh_id="human", num_vehicles=10)
    full = {"max_accel": 3, "max_decel": 3, "lane_change_duration": 5, "disable_tb": True,
            "disable_ramp_metering": True, "target_velocity": 30, "add_rl_if_exit": True}
    net = BottleneckNetwork(name="bay_bridge_toll", vehicles=vehicles,
                            net_params=NetParams(additional_params={"scaling": 1, "speed_limit": 23}))
    sim_params = SumoParams(sim_step=0.5, restart_instance=True)
    for key in full:
        with pytest.raises(KeyError):
            BottleneckAccelEnv(EnvParams(additional_params={k: v for k, v in full.items() if k != key}), sim_params, net)
    env = BottleneckAccelEnv(EnvParams(additional_params=full), sim_params, net)
    obs = env.reset()
    space = env.observation_space
    assert space.shape == (12,) and (space.low == 0).all() and (space.high == 1).all()
    assert env.action_space.shape == (0,)
    assert env.k.network.get_edge_list() == ["1", "2", "3", "4", "5", "fake_edge"]
    assert obs.shape == (12,) and (obs[-2:] == 0).all()
    for _ in range(40):
        obs, rew, done, _ = env.step(None)
    veh = env.k.vehicle
    ids = veh.get_ids()
    speeds = np.array(veh.get_speed(ids))
    for k, edge in enumerate(["1", "2", "3", "4", "5"]):
        on = [v for v in ids if veh.get_edge(v) == edge]
        want = [np.mean(veh.get_speed(on)) / 23.0, len(on) / env.k.network.edge_length(edge)] if on else [0, 0]
        np.testing.assert_allclose(obs[2 * k:2 * k + 2], want, rtol=1e-6)
    cost = np.linalg.norm(speeds - 30.0)
    mx = np.linalg.norm(np.full(len(ids), 30.0))
    np.testing.assert_allclose(rew, max(mx - cost, 0) / (mx + np.finfo(np.float32).eps), rtol=1e-6)
    assert 0 < rew < 1 and not done
    env.terminate()
    rl = VehicleParams()
    rl.add(veh_id="human", num_vehicles=5)
    rl.add(veh_id="rl", acceleration_controller=(RLController, {}), num_vehicles=2)
    net_rl = BottleneckNetwork(name="bay_bridge_toll", vehicles=rl,
                               net_params=NetParams(additional_params={"scaling": 1, "speed_limit": 23}))
    env = BottleneckAccelEnv(EnvParams(additional_params=full), sim_params, net_rl)      # (with RL vehicles: below)
    assert env.observation_space.shape == (2 * 6 + 4 * 4 * 2 + 4 * 2,) and env.action_space.shape == (4,)
    env.terminate()


def accel_oracle(env):
    """oracle/bottleneck_accel.py on the env's own spec + the network description its walks need."""
    from oracle.bottleneck_accel import BottleneckAccelOracle
    net, spec = env.k.network, dict(env._spec)
    rl_names = {env._spec["init_slot"][v]: v for v in env.rl_id_list}
    add = env.env_params.additional_params
    spec["accel_env"] = dict(
        path=[(e, net.edge_length(e), net.num_lanes(e)) for e in net._drop_path],
        connections={k: {c["fromLane"]: c["toLane"] for c in v} for k, v in env.network.connections.items()},
        edge_list=list(net.get_edge_list()), edge_length={e: net.edge_length(e) for e in net.get_edge_list()},
        rl_names=rl_names, lane_change_duration=add["lane_change_duration"], scaling=env.scaling,
        add_rl_if_exit=add["add_rl_if_exit"], max_speed=net.max_speed(), max_accel=add["max_accel"],
        max_decel=add["max_decel"],
        lane_change_mode={i: int(v.get("lane_change_mode", 0)) for i, v in enumerate(spec["vehicles"])})
    return BottleneckAccelOracle(spec, env.sim.real)


@pytest.mark.parametrize("lc_mode,slots", [(512, 64), (0, 64), (512, 100)])
def test_bottleneck_accel_env_with_rl_vehicles_equals_the_oracle(lc_mode, slots):
    """BottleneckAccelEnv with RL vehicles (flow/envs/bottleneck.py:486-757): accelerations and lane-change commands of
    three RL vehicles among inflow traffic, 500 steps of 0.5 s -- RL vehicles leave the network and are put back
    (add_rl_if_exit).  Observation (rl block, per-lane leaders / followers across the lane drops, per-edge block) and
    reward against oracle/bottleneck_accel.py, which walks the reference's per-edge lists; state bit for bit."""
    from flow_amd import _lib as L
    from flow_amd.controllers import RLController
    from flow_amd.core.params import (EnvParams, InFlows, InitialConfig, NetParams, SumoLaneChangeParams, SumoParams,
                                      VehicleParams)
    from flow_amd.envs import BottleneckAccelEnv
    from flow_amd.networks import BottleneckNetwork
    vehicles = VehicleParams()
    vehicles.add(veh_id="human", num_vehicles=6)
    vehicles.add(veh_id="rl", acceleration_controller=(RLController, {}),
                 lane_change_params=SumoLaneChangeParams(lane_change_mode=lc_mode), num_vehicles=3)
    inflow = InFlows()
    inflow.add(veh_type="human", edge="1", vehs_per_hour=1800, departLane="random", departSpeed=10)
    add = {"max_accel": 3, "max_decel": 3, "lane_change_duration": 5, "disable_tb": True, "disable_ramp_metering": True,
           "target_velocity": 30, "add_rl_if_exit": True}
    net = BottleneckNetwork(name="bottleneck", vehicles=vehicles, initial_config=InitialConfig(spacing="uniform", edges_distribution=["2", "3"]),
                            net_params=NetParams(inflows=inflow, additional_params={"scaling": 1, "speed_limit": 23}))
    env = BottleneckAccelEnv(EnvParams(horizon=600, additional_params=add), SumoParams(sim_step=0.5, seed=7, max_vehicles=slots), net)
    ora = accel_oracle(env)
    obs = env.reset()
    ora.reset()
    np.testing.assert_allclose(obs, ora.accel_state(0), rtol=0, atol=1e-12)
    assert obs.shape == env.observation_space.shape == (12 + 48 + 12,)
    rng = np.random.default_rng(3)
    readded, changes, seen_internal = 0, 0, False
    for k in range(500):
        a = rng.uniform(-1, 1, 6) * np.tile([3.0, 1.4], 3)          # (clipped to the Box: |direction| <= 1 rounds to -1 / 0 / 1)
        if k % 3:
            a[1::2] = 0.0                                            # a lane-change command every third step
        before = set(env.k.vehicle.get_rl_ids())
        lanes0 = {v: env.k.vehicle.get_lane(v) for v in before}
        obs, rew, done, _ = env.step(a)
        o_ref, r_ref, d_ref = ora.step(a[None, :])
        np.testing.assert_allclose(obs, o_ref[0], rtol=0, atol=1e-12, err_msg="observation, step %d" % k)
        np.testing.assert_allclose(rew, r_ref[0], rtol=1e-12, atol=1e-12, err_msg="reward, step %d" % k)
        assert bool(done) == bool(d_ref[0])
        for f, ref in ((L.FS_FIELD_POS, ora.x), (L.FS_FIELD_VEL, ora.v), (L.FS_FIELD_ROUTE, ora.route)):
            got = env.sim.get_state(f)[0]
            alive = ora.route[0] >= 0
            np.testing.assert_array_equal(got[alive], ref[0][alive], err_msg="field %d, step %d" % (f, k))
        now = set(env.k.vehicle.get_rl_ids())
        readded += len(now - before)
        changes += sum(1 for v in now & before if env.k.vehicle.get_lane(v) != lanes0[v] and
                       env.k.vehicle.get_edge(v) in ("1", "2", "3"))
        seen_internal = seen_internal or any(env.k.vehicle.get_edge(v)[0] == ":" for v in now)
    assert readded >= 2 and changes >= 5 and seen_internal
    assert env.sim.last_kernel.startswith("k_steps_open" if slots == 64 else "k_steps_wide")   # (100 slots: two waves)
    env.terminate()


def test_desired_velocity_env_equals_oracle_on_the_c4_configuration():
    env = make_env(c4_flow_params(horizon=300))
    assert env.observation_space.shape == (141,) and env.action_space.shape == (20,)
    assert env.action_space.low[0] == -1.5 and env.action_space.high[0] == 1.5       # max_decel * sim_step
    spec = env._spec
    assert spec["num_vehicles"] == 64 and spec["num_rl"] == 20 and len(spec["obs_cells"]) == 35
    assert [f["route"] for f in spec["inflows"]] == [-1, -1] and spec["speed_limit"] == 23
    ora = O.MergeOracle(spec, np.float32)
    obs = env.reset()
    np.testing.assert_array_equal(obs, ora.reset()[0].astype(np.float32))
    rng = np.random.default_rng(0)
    for k in range(300):
        a = rng.uniform(-1.5, 1.5, 20).astype(np.float32)
        obs, rew, done, _ = env.step(a)
        o_ref, r_ref, d_ref = ora.step(a[None, :])
        np.testing.assert_array_equal(obs, o_ref[0].astype(np.float32))
        assert rew == np.float32(r_ref[0]) and done == bool(d_ref[0])
    veh = env.k.vehicle
    ids = veh.get_ids()
    assert len(ids) == int(ora.alive[0].sum()) > 20
    assert any(v.startswith("flow_1.") for v in veh.get_rl_ids())
    for v in ids[:10]:
        edge, lane = veh.get_edge(v), veh.get_lane(v)
        assert 0 <= lane < (env.k.network.num_lanes(edge) if edge[0] != ':' else 4)
    rl = veh.get_rl_ids()[0]
    assert 0.01 <= veh.get_max_speed(rl) <= 23.0
    veh.set_max_speed(rl, 11.0)
    assert veh.get_max_speed(rl) == 11.0
    env.additional_command()
    assert sum(len(lane) for e in "12345" for lane in env.edge_dict[e]) > 15
    assert veh.get_outflow_rate(100) > 500
    env.terminate()


def test_desired_velocity_env_with_200_vehicle_slots_holds_the_whole_queue():
    """SumoParams(max_vehicles=200): the replica runs on k_steps_wide (one workgroup of four waves); the queue upstream
    of the lane drops outgrows 64 vehicles and nothing is dropped for lack of a slot."""
    import random
    from flow_amd import _lib as L
    random.seed(5)                     # (restart_instance: the simulator's seed is drawn from `random` at reset)
    env = make_env(c4_flow_params(horizon=700, max_vehicles=200))
    spec = env._spec
    assert spec["num_vehicles"] == 200 and spec["num_rl"] == 20
    ora = O.MergeOracle(spec, np.float32)
    obs = env.reset()
    np.testing.assert_array_equal(obs, ora.reset()[0].astype(np.float32))
    rng = np.random.default_rng(1)
    for k in range(480):
        a = rng.uniform(-1.5, 1.5, 20).astype(np.float32)
        obs, rew, done, _ = env.step(a)
        o_ref, r_ref, d_ref = ora.step(a[None, :])
        np.testing.assert_array_equal(obs, o_ref[0].astype(np.float32))
        assert rew == np.float32(r_ref[0]) and done == bool(d_ref[0])
    veh = env.k.vehicle
    ids = veh.get_ids()
    assert len(ids) == int(ora.alive[0].sum()) > 64
    assert len(set(ids)) == len(ids) and all(veh.get_edge(v) for v in ids)
    h = np.array([veh.get_headway(v) for v in ids])          # (a zipper partner on the other lane may overlap)
    alive = ora.alive[0]
    np.testing.assert_array_equal(np.sort(h.astype(np.float32)), np.sort(ora.h[0][alive].astype(np.float32)))
    assert len(veh.get_rl_ids()) >= 3
    assert int(env.sim.get_state(L.FS_FIELD_COUNTERS)[0, 6]) == int(ora.total_departed[0])
    env.terminate()


def test_reset_inflow_draws_a_new_rate_like_the_reference():
    """test_environments.py:886-947: np.random.seed(123) -> the constructor's toll_wait_time draws, then reset()
    draws uniform(1000, 2000) = 1719.47 veh/h.  The reference then measures ~1353.6 veh/h entering (SUMO drops the
    random-lane vehicles it cannot insert); the loss depends on SUMO's insertion checks and is not pinned here."""
    np.random.seed(seed=123)
    fp = c4_flow_params(horizon=600, warmup_steps=0, reset_inflow=True, flow_rate=1500)
    env = make_env(fp)
    env.reset()
    rates = sorted(f["vehsPerHour"] for f in env.network.net_params.inflows.get())
    np.testing.assert_allclose(sum(rates), 1719.468969785563, rtol=1e-12)
    np.testing.assert_allclose(rates[0] / sum(rates), 0.1, rtol=1e-12)
    for _ in range(500):
        env.step(rl_actions=None)
    measured = env.k.vehicle.get_inflow_rate(250)
    from flow_amd import _lib as L
    dropped = int(env.sim.get_state(L.FS_FIELD_COUNTERS)[0, 7])
    assert 1000 <= measured <= 1719.5 + 15 and measured >= 0.75 * 1719.5 - dropped
    env.terminate()


def test_vec_env_runs_the_c4_configuration():
    import torch
    from flow_amd import _lib as L
    from flow_amd.envs import VecFlowEnv
    vec = VecFlowEnv(c4_flow_params(), num_replicas=128, device=0)
    obs = vec.reset()
    assert obs.shape == (128, 141)
    K = 200
    o = torch.empty((K, 128, 141), dtype=torch.float32, device=vec.device)
    r = torch.empty((K, 128), dtype=torch.float32, device=vec.device)
    d = torch.empty((K, 128), dtype=torch.uint8, device=vec.device)
    acts = (torch.rand((K, 128, 20), device=vec.device) * 2 - 1) * 1.5
    vec.sim.rollout_dev(K, o, r, d, actions=acts)
    vec.sim.sync()
    cnt = vec.sim.get_state(L.FS_FIELD_COUNTERS)
    assert (cnt[:, 0] == 1 + 40 + K).all() and (cnt[:, 6] > 50).all()
    assert torch.isfinite(o).all() and float(r.max()) > 0.3 and (o[-1, :, -1] > 0).any()


def test_simulate_script_runs_the_bottleneck_experiment(tmp_path):
    """examples/simulate.py bottleneck (the reference's examples/exp_configs/non_rl/bottleneck.py: BottleneckEnv, random
    initial placement, lane_change_mode 1621 -> the simplified lane-change model): runs through install_as_flow() +
    Experiment.run and reports an outflow."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "simulate.py"), "bottleneck"],
                         cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "steps/second" in out.stdout and "Round 0, return" in out.stdout
    m = re.search(r"Average, std outflows: ([0-9.]+)", out.stdout)
    assert m and 800 < float(m.group(1)) < 2400, out.stdout[-800:]


def test_symmetric_actions_and_the_evaluate_reward_of_the_desired_velocity_env():
    """bottleneck.py:802-851, 946-949: symmetric=True -- one action per segment for all its lanes (6 instead of 20), found
    at ``bucket + action_index[edge]`` where the offsets advance by ONE per controlled edge (the reference's own
    arithmetic: segments of neighbouring edges share an entry) -- steps exactly like the per-lane environment fed the
    gathered row.  :971-978: evaluate=True pays nothing until the horizon, then get_outflow_rate(500)."""
    from flow_amd.envs import VecFlowEnv
    fp_sym, fp_ref = c4_flow_params(horizon=200, warmup_steps=0), c4_flow_params(horizon=200, warmup_steps=0)
    fp_sym["env"].additional_params["symmetric"] = True
    import random
    envs = []
    for fp in (fp_sym, fp_ref):                      # (restart_instance draws the simulator's seed from `random`)
        random.seed(11)
        env = make_env(fp)
        envs.append((env, env.reset()))
    (sym, o_sym), (ref, o_ref) = envs
    assert sym.action_space.shape == (6,) and ref.action_space.shape == (20,)
    assert sym.action_index == {"2": [0], "3": [1], "4": [2]} and ref.action_index == {"2": [0], "3": [8], "4": [16]}
    np.testing.assert_array_equal(o_sym, o_ref)
    rng = np.random.default_rng(4)
    for k in range(150):
        a = rng.uniform(-1.5, 1.5, 6).astype(np.float32)
        wide = np.concatenate([np.repeat(a[0:2], 4), np.repeat(a[1:3], 4), np.repeat(a[2:4], 2)])   # edges 2, 3, 4
        o1, r1, d1, _ = sym.step(a)
        o2, r2, d2, _ = ref.step(wide)
        np.testing.assert_array_equal(o1, o2)
        assert r1 == r2 and d1 == d2
    assert len(sym.k.vehicle.get_rl_ids()) >= 2
    sym.terminate(), ref.terminate()
    with pytest.raises(NotImplementedError, match="scalar Env only"):
        VecFlowEnv(fp_sym, num_replicas=4, device=0)
    fp_ev = c4_flow_params(horizon=120, warmup_steps=0)
    fp_ev["env"].evaluate = True
    ev = make_env(fp_ev)
    ev.reset()
    for k in range(120):
        obs, rew, done, _ = ev.step(np.zeros(20, dtype=np.float32))
        if k < 119:
            assert rew == 0 and not done
    arrived = sum(ev.k.vehicle._num_arrived)
    assert done and arrived >= 3 and rew == pytest.approx(3600 * arrived / (120 * 0.5))
    ev.terminate()
