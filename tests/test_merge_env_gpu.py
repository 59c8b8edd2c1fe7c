"""The reference-facing classes of the open-network row (MergeNetwork, MergePOEnv, MultiAgentMergePOEnv,
InFlows) on the GPU step loop, checked against the oracle and against the reference's own tests
(tests/fast_tests/test_environments.py:616-700, 1140-1225)."""
import numpy as np
import pytest

from oracle import opennet as O

pytestmark = pytest.mark.gpu


def merge_flow_params(env_name, num_rl=5, horizon=600, sims_per_step=1, sim_step=0.2, noise=0.0, n_human=5, pre=500,
                      **sim_kw):
    """examples/exp_configs/rl/multiagent/multiagent_merge.py:38-132 / flow/benchmarks/merge0.py with
    selectable noise (the bit-exact comparisons need it off)."""
    from flow_amd.controllers import IDMController, RLController
    from flow_amd.core.params import (EnvParams, InFlows, InitialConfig, NetParams, SumoCarFollowingParams,
                                      SumoParams, VehicleParams)
    from flow_amd.networks.merge import ADDITIONAL_NET_PARAMS, MergeNetwork
    add = ADDITIONAL_NET_PARAMS.copy()
    add["pre_merge_length"] = pre
    vehicles = VehicleParams()
    vehicles.add(veh_id="human", acceleration_controller=(IDMController, {"noise": noise}),
                 car_following_params=SumoCarFollowingParams(speed_mode="obey_safe_speed"), num_vehicles=n_human)
    vehicles.add(veh_id="rl", acceleration_controller=(RLController, {}),
                 car_following_params=SumoCarFollowingParams(speed_mode="obey_safe_speed"), num_vehicles=0)
    inflow = InFlows()
    inflow.add(veh_type="human", edge="inflow_highway", vehs_per_hour=0.9 * 2000, departLane="free", departSpeed=10)
    inflow.add(veh_type="rl", edge="inflow_highway", vehs_per_hour=0.1 * 2000, departLane="free", departSpeed=10)
    inflow.add(veh_type="human", edge="inflow_merge", vehs_per_hour=100, departLane="free", departSpeed=7.5)
    env_add = {"max_accel": 1.5, "max_decel": 1.5, "target_velocity": 20}
    if num_rl is not None:
        env_add["num_rl"] = num_rl
    return dict(exp_tag="merge", env_name=env_name, network=MergeNetwork, simulator='traci',
                sim=SumoParams(sim_step=sim_step, render=False, restart_instance=True, **sim_kw),
                env=EnvParams(horizon=horizon, sims_per_step=sims_per_step, warmup_steps=0, additional_params=env_add),
                net=NetParams(inflows=inflow, additional_params=add), veh=vehicles, initial=InitialConfig())


def make_env(flow_params):
    from flow_amd.utils.registry import make_create_env
    create_env, _ = make_create_env(flow_params)
    return create_env()


def two_vehicle_network():
    from flow_amd.controllers import IDMController, RLController
    from flow_amd.core.params import NetParams, VehicleParams
    from flow_amd.networks.merge import ADDITIONAL_NET_PARAMS, MergeNetwork
    vehicles = VehicleParams()
    vehicles.add("rl", acceleration_controller=(RLController, {}), num_vehicles=1)
    vehicles.add("human", acceleration_controller=(IDMController, {}), num_vehicles=1)
    return MergeNetwork(name="test_merge", vehicles=vehicles,
                        net_params=NetParams(additional_params=ADDITIONAL_NET_PARAMS.copy()))


def test_reference_merge_po_env_space_and_observed_tests():
    """test_environments.py:616-695."""
    from flow_amd.core.params import EnvParams, SumoParams
    from flow_amd.envs import MergePOEnv
    env_params = EnvParams(additional_params={"max_accel": 3, "max_decel": 3, "target_velocity": 25, "num_rl": 5})
    env = MergePOEnv(sim_params=SumoParams(), network=two_vehicle_network(), env_params=env_params)
    assert env.observation_space.shape == (25,) and env.observation_space.low.min() == 0 \
        and env.observation_space.high.max() == 1
    assert env.action_space.shape == (5,) and env.action_space.low.min() == -3 and env.action_space.high.max() == 3
    env.reset()
    env.step(None)
    env.additional_command()
    assert env.k.vehicle.get_observed_ids() == ["human_0"]
    assert env.rl_veh == ["rl_0"] and list(env.rl_queue) == []
    env.terminate()


def test_reference_multiagent_merge_env_space_and_observed_tests():
    """test_environments.py:1140-1222."""
    from flow_amd.core.params import EnvParams, SumoParams
    from flow_amd.envs.multiagent import MultiAgentMergePOEnv
    env_params = EnvParams(additional_params={'max_accel': 1, 'max_decel': 1, "target_velocity": 25})
    env = MultiAgentMergePOEnv(sim_params=SumoParams(), network=two_vehicle_network(), env_params=env_params)
    assert env.observation_space.shape == (5,) and env.observation_space.low.min() == -5 \
        and env.observation_space.high.max() == 5
    assert env.action_space.shape == (1,) and env.action_space.low.min() == -1 and env.action_space.high.max() == 1
    obs = env.reset()
    assert set(obs.keys()) == {"rl_0"} and obs["rl_0"].shape == (5,)
    states, reward, done, infos = env.step(None)
    env.additional_command()
    assert env.k.vehicle.get_observed_ids() == ["human_0"]
    assert set(states) == {"rl_0"} and set(reward) == {"rl_0"} and done == {"rl_0": False, "__all__": False}
    env.terminate()


def test_merge_po_env_equals_oracle_and_names_vehicles_like_sumo():
    from flow_amd.envs import MergePOEnv
    env = make_env(merge_flow_params(MergePOEnv, num_rl=3, horizon=400, slot_capacity={"human": 50, "rl": 8}))
    ora = O.MergeOracle(env._spec, np.float32)
    obs = env.reset()
    np.testing.assert_array_equal(obs, ora.reset()[0].astype(np.float32))
    assert env.k.vehicle.get_ids() == ["human_%d" % i for i in range(5)]
    rng = np.random.default_rng(0)
    seen_rl, arrived = set(), 0
    for k in range(400):
        a = rng.uniform(0.0, 1.5, 3).astype(np.float32)
        obs, rew, done, _ = env.step(a)
        o_ref, r_ref, d_ref = ora.step(a[None, :])
        np.testing.assert_array_equal(obs, o_ref[0].astype(np.float32))
        assert rew == np.float32(r_ref[0]) and done == bool(d_ref[0])
        seen_rl.update(env.rl_veh)
        arrived += env.k.vehicle.get_num_arrived()
        # host mirror of rl_veh: the alive + ghost slots in joining order
        order = np.argsort(np.where(ora.ctl_seq[0] >= 0, ora.ctl_seq[0], 1 << 40))[:(ora.ctl_seq[0] >= 0).sum()]
        assert len(env.rl_veh) == len(order)
    ids = env.k.vehicle.get_ids()
    assert any(v.startswith("flow_0.") for v in ids) and any(v.startswith("flow_1.") for v in seen_rl)
    assert env.k.vehicle.num_vehicles == int(ora.alive[0].sum()) and arrived == int(ora.total_arrived[0])
    assert env.k.vehicle.get_rl_ids() == sorted(env.k.vehicle.get_rl_ids())
    v = ids[len(ids) // 2]
    lead = env.k.vehicle.get_leader(v)
    assert lead is None or env.k.vehicle.get_x_by_id(lead) != -1001
    assert env.k.vehicle.get_edge(v) in ("inflow_highway", ":left_0", "left", ":center_1", "center", "inflow_merge",
                                         ":bottom_0", "bottom", ":center_0")
    assert env.k.vehicle.get_outflow_rate(100) > 0 and env.k.vehicle.get_inflow_rate(100) > 0
    env.terminate()


def test_multiagent_merge_env_dicts_and_the_enumerate_quirk():
    from flow_amd.envs.multiagent import MultiAgentMergePOEnv

    class Fixed(MultiAgentMergePOEnv):
        APPLY_ENUMERATE_QUIRK = False

    outs = []
    for cls in (MultiAgentMergePOEnv, Fixed):
        env = make_env(merge_flow_params(cls, num_rl=None, horizon=300, sims_per_step=5,
                                         slot_capacity={"human": 50, "rl": 8}))
        ora = O.MergeOracle(env._spec, np.float32)
        states = env.reset()
        ora.reset()
        assert states == {}
        n_agents, n_done = 0, 0
        for k in range(120):
            acts = {rl: np.array([0.7], dtype=np.float32) for rl in states}
            row = np.full((1, env.sim.act_dim), np.nan, dtype=np.float32)
            for rl, col in env._rl_columns().items():      # the slot of each agent BEFORE the step
                if rl in acts:
                    row[0, col] = 0.7
            states, reward, done, _ = env.step(acts)
            o_ref, r_ref, d_ref = ora.step(row)
            live = [rl for rl in states if states[rl] is not None]
            for rl in live:
                col = env._rl_columns()[rl]
                np.testing.assert_array_equal(states[rl].astype(np.float32), o_ref[0, 5 * col:5 * col + 5].astype(np.float32))
                assert reward[rl] == np.float32(r_ref[0])
            assert done["__all__"] == bool(d_ref[0])
            n_agents = max(n_agents, len(live))
            n_done += sum(1 for rl in done if rl != "__all__" and done[rl])
        assert n_agents >= 2
        outs.append(ora.v.copy())
        env.terminate()
    assert not np.array_equal(outs[0], outs[1])          # applying the actions changes the traffic


def test_vec_env_runs_the_c5_merge_configuration():
    import torch
    from flow_amd.envs import VecFlowEnv
    from flow_amd.envs.multiagent import MultiAgentMergePOEnv
    fp = merge_flow_params(MultiAgentMergePOEnv, num_rl=None, horizon=600, sims_per_step=5, noise=0.2)
    vec = VecFlowEnv(fp, num_replicas=64, device=0)
    obs = vec.reset()
    assert obs.shape == (64, vec.obs_dim)
    K = 50
    o = torch.empty((K, 64, vec.obs_dim), dtype=torch.float32, device=vec.device)
    r = torch.empty((K, 64), dtype=torch.float32, device=vec.device)
    d = torch.empty((K, 64), dtype=torch.uint8, device=vec.device)
    vec.sim.rollout_dev(K, o, r, d)
    vec.sim.sync()
    from flow_amd import _lib as L
    cnt = vec.sim.get_state(L.FS_FIELD_COUNTERS)
    assert (cnt[:, 0] == 1 + 5 * K).all() and (cnt[:, 6] > 20).all()
    assert torch.isfinite(o).all() and (r >= 0).all()


def test_probabilistic_inflow_through_the_env_api():
    """InFlows.add(probability=p) (params.py:1103-1105, 1080-1213) reaches the kernel through flow_params: about
    p * T vehicles of that flow depart, differently in every replica."""
    from flow_amd import _lib as L
    from flow_amd.core.params import InFlows
    from flow_amd.envs import MergePOEnv, VecFlowEnv
    fp = merge_flow_params(MergePOEnv, num_rl=3, horizon=500, n_human=0)
    inflow = InFlows()
    inflow.add(veh_type="human", edge="inflow_highway", probability=0.3, departLane="free", departSpeed=10)
    inflow.add(veh_type="rl", edge="inflow_highway", vehs_per_hour=100, departLane="free", departSpeed=10)
    fp["net"].inflows = inflow
    vec = VecFlowEnv(fp, num_replicas=32, device=0)
    vec.reset()
    vec.rollout(500)                                     # 100 s of 0.2 s steps: ~0.3 * 99 vehicles of the first flow
    origin = vec.sim.get_state(L.FS_FIELD_ORIGIN)
    departed = vec.sim.get_state(L.FS_FIELD_COUNTERS)[:, 6]
    assert 15 <= departed.mean() <= 45 and len(set(departed.tolist())) > 3, departed
    assert (origin[origin >= 0] >> 20 <= 1).all()
    vec.close()


def test_simulate_script_runs_the_merge_experiment_and_writes_the_emission_file(tmp_path):
    """examples/simulate.py merge (the reference's examples/exp_configs/non_rl/merge.py experiment, 3600 steps of
    5 sub-steps): runs through install_as_flow() + Experiment.run and leaves a trajectory CSV whose vehicle ids and
    edges are those SUMO would report."""
    import csv
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "simulate.py"), "merge", "--gen_emission"],
                         cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "steps/second" in out.stdout and "Round 0, return" in out.stdout
    files = os.listdir(os.path.join(str(tmp_path), "data"))
    assert len(files) == 1 and files[0].endswith("-emission.csv")
    rows = list(csv.DictReader(open(os.path.join(str(tmp_path), "data", files[0]))))
    ids = {r["id"] for r in rows}
    edges = {r["edge_id"] for r in rows}
    assert {"human_0", "flow_0.0", "flow_1.0"} <= ids and len(ids) > 1500        # ~2100 veh/h for one hour
    assert {"inflow_highway", "left", "center", "inflow_merge", "bottom"} <= edges
    last = {}
    for r in rows:                                   # every vehicle only ever moves forward along its route
        key = r["id"]
        x = float(r["x"]) if "x" in r and r["x"] not in ("", None) else None
        t = float(r["time"])
        assert key not in last or t >= last[key]
        last[key] = t
