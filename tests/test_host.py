"""CPU tests of the host-side mirror of the reference interface: parameter classes and their
defaults (against defaults dumped from the imported reference), controller descriptors,
network geometry / placement (against the oracle's literal restatement), spec resolution,
and that constructing an environment without a GPU fails loudly."""
import inspect
import json
import os

import numpy as np
import pytest

from flow_amd import _lib as L
from flow_amd import controllers as FC
from flow_amd.core import params as P
from flow_amd.core.kernel.network import NetworkKernel
from flow_amd.networks import RingNetwork
from flow_amd.networks.ring import ADDITIONAL_NET_PARAMS, ring_start_positions
from oracle import network as ONet

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
DEF = json.load(open(os.path.join(GOLDEN, "defaults.json")))


def sig(cls):
    out = {}
    for k, v in inspect.signature(cls.__init__).parameters.items():
        if k in ("self", "kwargs") or v.default is inspect._empty:
            continue
        d = v.default
        out[k] = d if isinstance(d, (int, float, str, bool, type(None))) else repr(d)
    return out


@pytest.mark.parametrize("name", ["SumoParams", "EnvParams", "NetParams", "InitialConfig", "SimParams"])
def test_param_class_defaults_match_reference(name):
    mine = sig(getattr(P, name))
    for k, v in DEF[name].items():
        assert k in mine, (name, k)
        if not (isinstance(v, str) and v.startswith("<")):
            assert mine[k] == v or (isinstance(v, float) and np.isinf(v) and np.isinf(mine[k])), (name, k)


def test_mode_tables_and_sumo_param_dicts():
    assert P.SPEED_MODES == DEF["SPEED_MODES"] and P.LC_MODES == DEF["LC_MODES"]
    cf = P.SumoCarFollowingParams()
    assert cf.controller_params == DEF["SumoCarFollowingParams"]["controller_params"]
    assert cf.speed_mode == DEF["SumoCarFollowingParams"]["speed_mode"] == 25
    assert P.SumoCarFollowingParams(speed_mode="aggressive").speed_mode == 0
    lc = P.SumoLaneChangeParams()
    assert lc.controller_params == DEF["SumoLaneChangeParams"]["controller_params"]
    assert lc.lane_change_mode == 512


@pytest.mark.parametrize("name", list(DEF["controllers"]))
def test_controller_signatures_match_reference(name):
    mine = sig(getattr(FC, name))
    assert mine == DEF["controllers"][name]


def test_vehicle_params_bookkeeping():
    v = P.VehicleParams()
    v.add("human", acceleration_controller=(FC.IDMController, {}), num_vehicles=3)
    v.add("rl", acceleration_controller=(FC.RLController, {}), num_vehicles=2)
    d = DEF["VehicleParams"]
    assert v.ids == d["ids"] and v.num_vehicles == d["num_vehicles"]
    assert v.num_rl_vehicles == d["num_rl_vehicles"] and v.minGap == d["minGap"]
    assert v.get_type("rl_1") == "rl"


def test_inflows_validation():
    f = P.InFlows()
    with pytest.raises(ValueError):
        f.add("e", "human")
    with pytest.raises(ValueError):
        f.add("e", "human", vehs_per_hour=10, probability=0.2)
    f.add("e", "human", vehs_per_hour=10)
    assert f.get()[0]["vehsPerHour"] == 10


def test_controller_descriptors_pack_parameters():
    cf = P.SumoCarFollowingParams(accel=1.2, decel=-3.0)
    c = FC.IDMController("v", car_following_params=cf, noise=0.2, fail_safe="safe_velocity", time_delay=0.3)
    assert c.FS_ID == L.FS_CTRL_IDM and c.fs_params() == [30, 1, 1, 1.5, 4, 2]
    assert (c.max_accel, c.max_deaccel, c.accel_noise, c.delay) == (1.2, 3.0, 0.2, 0.3)
    f = FC.FollowerStopper("v", cf, v_des=7.5)
    assert f.fail_safe == "safe_velocity" and f.delay == 1.0 and f.fs_params() == [7.5]
    assert FC.RLController("v", cf).FS_ID == L.FS_CTRL_RL
    assert FC.SimCarFollowingController("v", cf).FS_ID == L.FS_CTRL_SIM
    with pytest.raises(ValueError):
        FC.IDMController("v", car_following_params=cf, fail_safe="bogus")
    pis = FC.PISaturation("v", cf)
    assert pis.FS_ID == L.FS_CTRL_PISATURATION and pis.delay == 1.0 and pis.fail_safe is None


def make_ring(length=230, n=22, **ic):
    v = P.VehicleParams()
    v.add("idm", acceleration_controller=(FC.IDMController, {}), routing_controller=(FC.ContinuousRouter, {}),
          num_vehicles=n)
    add = dict(ADDITIONAL_NET_PARAMS)
    add["length"] = length
    return RingNetwork("ring", v, P.NetParams(additional_params=add), P.InitialConfig(**ic))


def test_ring_network_requires_its_params():
    with pytest.raises(KeyError):
        RingNetwork("r", P.VehicleParams(), P.NetParams(additional_params={"length": 230}))


@pytest.mark.parametrize("length,n,ic", [(230, 22, dict(bunching=20)), (1000, 15, dict(x0=5)),
                                          (260, 22, dict(bunching=50, min_gap=0)), (100, 10, dict(min_gap=0.5))])
def test_network_kernel_matches_oracle_geometry(length, n, ic):
    net = make_ring(length, n, **ic)
    k = NetworkKernel(net, junction_length=0.1)
    o = ONet.ring_network(length, junction_length=0.1)
    assert abs(k.length() - o.length()) < 1e-12 and k.non_internal_length() == o.non_internal_length()
    assert k.max_speed() == 30 and k.edge_length("bottom") == length / 4
    for x in np.linspace(0, length + 0.39, 57):
        assert k.get_edge(x) == o.get_edge(x)
    pos, lanes = k.generate_starting_positions(net.initial_config, n)
    opos, olanes = o.gen_even_start_pos(n, **ic)
    assert pos == opos and lanes == olanes
    assert k.get_x("", 0) == -1001 and k.get_x(":top_0", 0.05) == o.get_x(":top_0", 0.05)
    assert k.next_edge("bottom", 0) == [(":right_0", 0)] and k.prev_edge("bottom", 0) == [(":bottom_0", 0)]


def test_not_enough_space_raises():
    from flow_amd.utils.exceptions import FatalFlowError
    net = make_ring(100, 30)
    with pytest.raises(FatalFlowError):
        NetworkKernel(net).generate_starting_positions(net.initial_config, 30)


def test_ring_start_positions_c1():
    x = ring_start_positions(22, length=230.0, bunching=20.0)
    np.testing.assert_allclose(x, np.arange(22) * (100 / 22 + 5), atol=1e-9)


class RecordingSim:
    """Stands in for FlowSim on the CPU: records the spec the env resolved."""
    last = None

    def __init__(self, spec, precision="f32", device=0):
        RecordingSim.last = (spec, precision)
        self.spec = spec
        self.R, self.N = spec["num_replicas"], spec["num_vehicles"]

    def close(self):
        pass


def build_env(monkeypatch, env_cls, env_params, sim_params, network):
    import flow_amd.envs.base as base
    monkeypatch.setattr(base, "FlowSim", RecordingSim)
    env = env_cls(env_params, sim_params, network)
    return env, RecordingSim.last[0]


def test_spec_resolution_c1(monkeypatch):
    from flow_amd.envs import AccelEnv
    from flow_amd.envs.ring.accel import ADDITIONAL_ENV_PARAMS
    net = make_ring(230, 22, bunching=20)
    env, spec = build_env(monkeypatch, AccelEnv, P.EnvParams(horizon=1500, additional_params=ADDITIONAL_ENV_PARAMS),
                          P.SumoParams(sim_step=0.1, render=False), net)
    assert spec["num_replicas"] == 1 and spec["num_vehicles"] == 22 and spec["num_rl"] == 0
    assert spec["env"] == L.FS_ENV_ACCEL and spec["horizon"] == 1500 and spec["max_speed"] == 30
    assert spec["target_velocity"] == 10 and abs(spec["slowdown_ramp"] - 0.1 / 0.101) < 1e-15
    v = spec["vehicles"][0]
    assert v["controller"] == L.FS_CTRL_IDM and v["p"] == [30, 1, 1, 1.5, 4, 2]
    assert (v["max_accel"], v["max_decel"], v["speed_mode"], v["sumo_min_gap"]) == (2.6, 4.5, 25, 2.5)
    np.testing.assert_allclose(spec["init_pos"][0], np.arange(22) * (100 / 22 + 5), atol=1e-9)
    assert env.initial_state["idm_7"][1] == "right" and env.observation_space.shape == (44,)
    assert env.action_space.shape == (0,) and env.initial_ids[0] == "idm_0"


def test_spec_resolution_rl_ring(monkeypatch):
    from flow_amd.envs import WaveAttenuationPOEnv
    v = P.VehicleParams()
    v.add("human", acceleration_controller=(FC.IDMController, {"noise": 0.2}),
          car_following_params=P.SumoCarFollowingParams(min_gap=0), routing_controller=(FC.ContinuousRouter, {}),
          num_vehicles=21)
    v.add("rl", acceleration_controller=(FC.RLController, {}), routing_controller=(FC.ContinuousRouter, {}),
          num_vehicles=1)
    add = dict(ADDITIONAL_NET_PARAMS)
    add["length"] = 260
    net = RingNetwork("stabilizing_the_ring", v, P.NetParams(additional_params=add), P.InitialConfig())
    ep = P.EnvParams(horizon=3000, warmup_steps=750, clip_actions=False,
                     additional_params={"max_accel": 1, "max_decel": 1, "ring_length": [220, 270]})
    env, spec = build_env(monkeypatch, WaveAttenuationPOEnv, ep, P.SumoParams(sim_step=0.1), net)
    assert spec["env"] == L.FS_ENV_WAVE_ATTENUATION_PO and spec["num_rl"] == 1
    assert spec["vehicles"][21]["controller"] == L.FS_CTRL_RL and spec["vehicles"][21]["rl_index"] == 0
    assert spec["vehicles"][0]["noise"] == 0.2 and spec["vehicles"][0]["sumo_min_gap"] == 0
    assert spec["po_max_length"] == 270 and spec["warmup_steps"] == 750 and spec["clip_actions"] is False
    assert (spec["action_low"], spec["action_high"]) == (-1.0, 1.0)
    assert env.observation_space.shape == (3,) and env.action_space.shape == (1,)
    assert env.k.vehicle.get_rl_ids() == ["rl_0"] and env.k.vehicle.get_leader("rl_0") == "human_0"


def test_missing_env_params_raise_keyerror(monkeypatch):
    from flow_amd.envs import AccelEnv, WaveAttenuationEnv
    net = make_ring()
    with pytest.raises(KeyError):                  # reference tests/fast_tests/test_environments.py:1333-1375
        build_env(monkeypatch, AccelEnv, P.EnvParams(additional_params={"max_accel": 1}), P.SumoParams(), net)
    with pytest.raises(KeyError):
        build_env(monkeypatch, WaveAttenuationEnv, P.EnvParams(additional_params={}), P.SumoParams(), net)


def test_unsupported_features_raise_at_construction(monkeypatch):
    from flow_amd.envs import AccelEnv
    from flow_amd.envs.ring.accel import ADDITIONAL_ENV_PARAMS
    v = P.VehicleParams()
    v.add("idm", acceleration_controller=(FC.IDMController, {}), num_vehicles=4)
    net = RingNetwork("ring", v, P.NetParams(additional_params=dict(ADDITIONAL_NET_PARAMS)),
                      P.InitialConfig(shuffle=True))
    # InitialConfig(shuffle=True) (envs/base.py:268-292): the start positions go to the ids in shuffled order; the
    # simulator keeps its slots in ring order and is told where each slot's vehicle sits in get_ids()
    import random
    random.seed(3)
    env, spec = build_env(monkeypatch, AccelEnv, P.EnvParams(additional_params=dict(ADDITIONAL_ENV_PARAMS,
                                                                                     sort_vehicles=True)),
                          P.SumoParams(), net)
    assert sorted(spec["obs_perm"].tolist()) == [0, 1, 2, 3] and spec["sort_vehicles"] is True
    assert [env.k.vehicle.get_ids()[q] for q in spec["obs_perm"]] == env.initial_ids
    assert env.k.vehicle.get_ids() == ["idm_0", "idm_1", "idm_2", "idm_3"]
    # slots are in ring order: the start positions along the loop ascend with the slot
    xs = [spec["init_pos"][0][env.k.vehicle._slot[v]] for v in env.k.vehicle._order]
    assert xs == sorted(xs)
    add3 = dict(ADDITIONAL_NET_PARAMS)
    add3["lanes"] = 3
    net3 = RingNetwork("ring", v, P.NetParams(additional_params=add3), P.InitialConfig(shuffle=True))
    from flow_amd.envs import LaneChangeAccelEnv
    from flow_amd.envs.ring.lane_change_accel import ADDITIONAL_ENV_PARAMS as LC_PARAMS
    # shuffle on a multi-lane ring: same permutation mechanism, the places are (position, lane) pairs
    env3, spec3 = build_env(monkeypatch, LaneChangeAccelEnv, P.EnvParams(additional_params=LC_PARAMS), P.SumoParams(),
                            net3)
    assert sorted(spec3["obs_perm"].tolist()) == [0, 1, 2, 3] and spec3["num_lanes"] == 3
    assert [env3.k.vehicle.get_ids()[q] for q in spec3["obs_perm"]] == env3.initial_ids
    inflow = P.InFlows()
    inflow.add("bottom", "idm", vehs_per_hour=100)
    net = RingNetwork("ring", v, P.NetParams(inflows=inflow, additional_params=dict(ADDITIONAL_NET_PARAMS)))
    with pytest.raises(NotImplementedError):
        build_env(monkeypatch, AccelEnv, P.EnvParams(additional_params=ADDITIONAL_ENV_PARAMS), P.SumoParams(), net)


def test_spec_resolution_multilane_lane_change_env(monkeypatch):
    """3-lane ring of reference tests/fast_tests/test_vehicles.py:199-253 (21 vehicles side by side) under
    LaneChangeAccelEnv: lanes 0,1,2 repeat, action space of test_environments.py:84-100."""
    from flow_amd.envs import LaneChangeAccelEnv
    from flow_amd.envs.ring.lane_change_accel import ADDITIONAL_ENV_PARAMS
    add = dict(ADDITIONAL_NET_PARAMS)
    add["lanes"] = 3
    v = P.VehicleParams()
    v.add("test", acceleration_controller=(FC.IDMController, {}), num_vehicles=19)
    v.add("rl", acceleration_controller=(FC.RLController, {}), num_vehicles=2,
          lane_change_params=P.SumoLaneChangeParams(lane_change_mode="aggressive"))
    net = RingNetwork("ring", v, P.NetParams(additional_params=add), P.InitialConfig(lanes_distribution=float("inf")))
    env, spec = build_env(monkeypatch, LaneChangeAccelEnv, P.EnvParams(additional_params=ADDITIONAL_ENV_PARAMS),
                          P.SumoParams(), net)
    assert spec["num_lanes"] == 3 and spec["env"] == L.FS_ENV_LANE_CHANGE_ACCEL and spec["num_rl"] == 2
    np.testing.assert_array_equal(spec["init_lane"][0], np.arange(21) % 3)
    np.testing.assert_allclose(spec["init_pos"][0][:6], [0, 0, 0, 230 / 7, 230 / 7, 230 / 7])
    assert spec["lane_change_duration"] == 5 and spec["lane_change_mode"] == 0 and spec["last_lc_quirk"] is True
    assert (spec["action_low"], spec["action_high"]) == (-3.0, 3.0)
    sp = env.action_space
    np.testing.assert_array_equal(sp.low, [-3, -1, -3, -1])
    np.testing.assert_array_equal(sp.high, [3, 1, 3, 1])
    assert env.observation_space.shape == (63,)
    with pytest.raises(ValueError):                       # vehicle/traci.py:973-975
        env.k.vehicle.apply_lane_change(["rl_0"], [0.5])


def test_env_construction_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from flow_amd.envs import AccelEnv
    from flow_amd.envs.ring.accel import ADDITIONAL_ENV_PARAMS
    from flow_amd.utils.exceptions import FatalFlowError
    from flow_amd.utils.registry import make_create_env
    v = P.VehicleParams()
    v.add("idm", acceleration_controller=(FC.IDMController, {}), num_vehicles=5)
    flow_params = dict(exp_tag="ring", env_name=AccelEnv, network=RingNetwork, simulator="traci",
                       sim=P.SumoParams(), env=P.EnvParams(additional_params=ADDITIONAL_ENV_PARAMS),
                       net=P.NetParams(additional_params=dict(ADDITIONAL_NET_PARAMS)), veh=v)
    create_env, name = make_create_env(flow_params)
    assert name.startswith("AccelEnv-v")
    with pytest.raises(FatalFlowError):
        create_env()


def test_figure_eight_geometry_known_answers_and_spec(monkeypatch):
    """reference tests/fast_tests/test_scenario_base_class.py:51-94 (get_x / get_edge on the figure eight),
    :656-693 (':center_*' length 9.40) and the netconvert lengths of tests/fast_tests/test_files/fig8_test.net.xml
    (total 421.94 m, non-internal 402.74 m)."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from helpers import figure_eight_tables
    from flow_amd.envs import AccelEnv
    from flow_amd.envs.ring.accel import ADDITIONAL_ENV_PARAMS
    from flow_amd.networks import FigureEightNetwork
    from flow_amd.networks.figure_eight import ADDITIONAL_NET_PARAMS as FIG8
    v = P.VehicleParams()
    v.add("idm", acceleration_controller=(FC.IDMController, {}), routing_controller=(FC.ContinuousRouter, {}),
          car_following_params=P.SumoCarFollowingParams(speed_mode="obey_safe_speed", decel=1.5), num_vehicles=14)
    net = FigureEightNetwork("figure8", v, P.NetParams(additional_params=dict(FIG8)))
    k = NetworkKernel(net)
    assert k.get_edge(5) == ("bottom", 4.72) and abs(k.get_x("bottom", 4.72) - 5) < 1e-12
    assert k.get_edge(0.1) == (":bottom", 0.1) and k.get_x(":bottom", 0.1) == 0.1
    np.testing.assert_allclose(k.edge_length(":center_0"), 9.40)
    np.testing.assert_allclose(k.edge_length("upper_ring"), 141.37, atol=5e-3)
    np.testing.assert_allclose(k.length(), 421.94, atol=5e-3)
    np.testing.assert_allclose(k.non_internal_length(), 402.74, atol=5e-3)
    assert k.locate(31.0) == (":center_1", 1.0) and k.locate(0.0) == ("bottom", 0.0)
    with pytest.raises(KeyError):
        FigureEightNetwork("f", v, P.NetParams(additional_params={"radius_ring": 30}))
    env, spec = build_env(monkeypatch, AccelEnv, P.EnvParams(horizon=1500, additional_params=ADDITIONAL_ENV_PARAMS),
                          P.SumoParams(), net)
    segs, junction, total, starts, onet = figure_eight_tables()
    np.testing.assert_allclose(np.array(spec["segments"], dtype=float), np.array(segs, dtype=float), atol=1e-12)
    for key, val in junction.items():
        np.testing.assert_allclose(spec["junction"][key], val, atol=1e-12)
    np.testing.assert_allclose(spec["ring_length"][0] + 0.4, total)
    assert spec["junction_mode"] == 1 and spec["vehicles"][0]["speed_mode"] == 1 and spec["vehicles"][0]["max_decel"] == 1.5
    pos, _ = onet.gen_even_start_pos(14)
    np.testing.assert_allclose(spec["init_pos"][0], [starts[e] + p for e, p in pos], atol=1e-12)
    assert env.observation_space.shape == (28,)


def test_flow_params_json_round_trip_and_reference_file(monkeypatch):
    """flow/utils/rllib.py FlowParamsEncoder / get_flow_params: (a) the reference's own stored file
    tests/fast_tests/test_files/ring_230.json (committed as tests/golden/ring_230_flow_params.json) loads into
    flow_amd objects and resolves to a spec; (b) encode -> decode round trip."""
    import json as _json
    from flow_amd.utils.rllib import FlowParamsEncoder, get_flow_params
    fp = get_flow_params(os.path.join(GOLDEN, "ring_230_flow_params.json"))
    from flow_amd.envs import WaveAttenuationPOEnv
    assert fp["env_name"] is WaveAttenuationPOEnv and fp["network"] is RingNetwork
    assert fp["veh"].num_vehicles == 22 and fp["veh"].num_rl_vehicles == 1
    assert fp["env"].warmup_steps == 750 and fp["env"].horizon == 3000 and fp["sim"].sim_step == 0.1
    assert fp["veh"].type_parameters["human"]["acceleration_controller"][0] is FC.IDMController
    assert fp["veh"].type_parameters["human"]["acceleration_controller"][1] == {"noise": 0.2}
    assert fp["veh"].type_parameters["human"]["car_following_params"].speed_mode == 25
    assert fp["initial"].lanes_distribution == float("inf")
    net = fp["network"](name=fp["exp_tag"], vehicles=fp["veh"], net_params=fp["net"], initial_config=fp["initial"])
    env, spec = build_env(monkeypatch, fp["env_name"], fp["env"], fp["sim"], net)
    assert spec["num_vehicles"] == 22 and spec["env"] == L.FS_ENV_WAVE_ATTENUATION_PO and spec["warmup_steps"] == 750
    assert spec["vehicles"][0]["noise"] == 0.2 and spec["vehicles"][21]["controller"] == L.FS_CTRL_RL

    text = _json.dumps(fp, cls=FlowParamsEncoder, sort_keys=True)
    again = get_flow_params({"env_config": {"flow_params": text}})
    assert again["env_name"] is WaveAttenuationPOEnv and again["network"] is RingNetwork
    assert again["veh"].ids == fp["veh"].ids and again["env"].additional_params == fp["env"].additional_params
    assert again["net"].additional_params == fp["net"].additional_params
    assert again["veh"].type_parameters["rl"]["acceleration_controller"][0] is FC.RLController


def test_package_alias_lets_reference_style_imports_resolve(monkeypatch):
    """after ``flow_amd.install_as_flow()`` the import lines of a reference experiment file
    (examples/exp_configs/non_rl/ring.py:6-10, examples/exp_configs/rl/singleagent/singleagent_ring.py:6-10,
    examples/simulate.py:9) resolve to this package."""
    import importlib
    import sys
    import flow_amd
    for name in [m for m in sys.modules if m == "flow" or m.startswith("flow.")]:
        monkeypatch.delitem(sys.modules, name)
    monkeypatch.setattr(sys, "meta_path", list(sys.meta_path))
    flow_amd.install_as_flow()
    ns = {}
    exec("from flow.controllers import IDMController, ContinuousRouter, RLController\n"
         "from flow.core.params import SumoParams, EnvParams, InitialConfig, NetParams\n"
         "from flow.core.params import VehicleParams, SumoCarFollowingParams\n"
         "from flow.envs.ring.accel import AccelEnv, ADDITIONAL_ENV_PARAMS\n"
         "from flow.envs import WaveAttenuationPOEnv\n"
         "from flow.networks.ring import RingNetwork, ADDITIONAL_NET_PARAMS\n"
         "from flow.networks import FigureEightNetwork\n"
         "from flow.core.experiment import Experiment\n"
         "from flow.utils.registry import make_create_env\n"
         "from flow.utils.rllib import FlowParamsEncoder, get_flow_params\n"
         "from flow.core import rewards\n", ns)
    assert ns["IDMController"] is FC.IDMController and ns["RingNetwork"] is RingNetwork
    assert ns["ADDITIONAL_ENV_PARAMS"]["target_velocity"] == 10 and ns["ADDITIONAL_NET_PARAMS"]["length"] == 230
    assert importlib.import_module("flow.envs").AccelEnv is ns["AccelEnv"]
    import flow_amd.envs
    assert ns["AccelEnv"] is flow_amd.envs.AccelEnv and ns["Experiment"].__module__ == "flow_amd.core.experiment"
    for name in [m for m in list(sys.modules) if m == "flow" or m.startswith("flow.")]:
        monkeypatch.delitem(sys.modules, name, raising=False)


def test_bench_roofline_traffic_comes_from_the_committed_pmc_summary():
    """bench.py's `roofline.traffic` is the HBM byte count of the newest committed PMC summary of the SAME kernel
    (profiles/rNN_pmc_summary_<dtype>.json); a summary of another kernel, size or fragment length -- or one whose launch
    time is more than 15 % off the launch the bench has just timed -- is refused, with the reason in place of the source."""
    import json
    import bench
    for prec in ("mixed", "f32"):
        traffic, src = bench.pmc_traffic(prec, 4096, 1500)
        nbytes, _ = bench.launch_bytes(4096, 22, 1500, prec)
        assert src.endswith("_pmc_summary_%s.json" % prec) and src[:1] == "r"
        assert 0.98 * nbytes < traffic < 1.10 * nbytes          # no wasted traffic: within 10 % of the algorithmic bytes
        ref_s = json.load(open(os.path.join(os.path.dirname(bench.__file__), "profiles", src)))["avg_launch_ns_kernel_trace"] * 1e-9
        assert bench.pmc_traffic(prec, 4096, 1500, ref_s * 1.1)[0] == traffic
        stale, why = bench.pmc_traffic(prec, 4096, 1500, ref_s * 1.3)
        assert stale is None and "stale" in why
        vi = bench.valu_issue(prec, 4096, 1500, ref_s)
        assert 0.5 < vi["frac"] < 1.0 and vi["frac"] > vi["frac_of_nominal"] and 1.8 < vi["effective_clock_GHz"] < 2.4
    assert bench.pmc_traffic("mixed", 4096, 20)[0] is None
    assert bench.pmc_traffic("mixed", 1024, 1500)[0] is None


def test_user_controller_header_and_library_tag():
    """flow_amd.build.user_header / build_user (the extension path for user-defined controllers): the header wraps the
    body in the documented signature; the library copy is keyed by the body and by the sources."""
    from flow_amd import build
    h = build.user_header("    return p[0] * (v_lead - v);\n")
    assert build.USER_SIGNATURE in h and "return p[0] * (v_lead - v);" in h and h.rstrip().endswith("}")
    from flow_amd.controllers import CompiledController
    from flow_amd.core.params import SumoCarFollowingParams
    with pytest.raises(ValueError):
        CompiledController("v", SumoCarFollowingParams())
    with pytest.raises(ValueError):
        CompiledController("v", SumoCarFollowingParams(), source="return T(0);", params=list(range(9)))
    c = CompiledController("v", SumoCarFollowingParams(), source="return T(0);", params=[1, 2])
    assert c.FS_ID == 12 and c.fs_params() == [1.0, 2.0]
