#!/usr/bin/env python3
"""Generate golden vectors by importing the reference's controllers/rewards.

Run ONLY in the build container (the reference tree is not present on the GPU
box, and nothing under tests/ reads it at test time):

    cd /tmp && SUMO_HOME=/tmp PYTHONDONTWRITEBYTECODE=1 \
        PYTHONPATH=/root/reference python3 /root/repo/tests/golden/gen_golden.py

Importable reference modules: flow.controllers, flow.core.params,
flow.core.rewards.  Everything else on the path (envs, kernel, networks) needs
gym/traci/sumolib and is restated from source text in oracle/.

Outputs (JSON, committed): controllers.json, failsafes.json, rewards.json.
The emission fixture ring_230_emission.csv is a column subset of the data file
the reference's own tests hold (tests/fast_tests/test_files/ring_230_emission.csv).
"""
import csv
import json
import os
import sys
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("FLOW_REFERENCE", "/root/reference")

os.environ.setdefault("SUMO_HOME", "/tmp")
sys.dont_write_bytecode = True
if REF not in sys.path:
    sys.path.insert(0, REF)

import flow.controllers as fc                      # noqa: E402
from flow.core import rewards as fr                # noqa: E402
from flow.core.params import SumoCarFollowingParams  # noqa: E402


class StubVehicle:
    """The slice of env.k.vehicle that controllers and rewards read."""

    def __init__(self, ids, speed, headway, leader, follower, edge="bottom", length=5.0,
                 position=0.0, prev_speed=None):
        self.ids = list(ids)
        self.speed = dict(zip(ids, speed))
        self.headway = dict(zip(ids, headway))
        self.leader = dict(zip(ids, leader))
        self.follower = dict(zip(ids, follower))
        self.edge = {i: edge for i in ids} if isinstance(edge, str) else dict(zip(ids, edge))
        self.length = length
        self.position = {i: position for i in ids}
        self.prev = dict(zip(ids, prev_speed if prev_speed is not None else speed))
        self.num_vehicles = len(ids)

    def get_ids(self):
        return self.ids

    def get_speed(self, v, error=-1001):
        if isinstance(v, (list, np.ndarray)):
            return [self.get_speed(x) for x in v]
        return self.speed.get(v, error)

    def get_previous_speed(self, v):
        return self.prev[v]

    def get_headway(self, v):
        return self.headway[v]

    def get_leader(self, v):
        return self.leader[v]

    def get_follower(self, v):
        return self.follower[v]

    def get_edge(self, v):
        return self.edge[v]

    def get_length(self, v):
        return self.length

    def get_position(self, v):
        return self.position[v]


def ring_env(speeds, headways, dt=0.1, additional=None):
    n = len(speeds)
    ids = ["test_%d" % i for i in range(n)]
    leader = [ids[(i + 1) % n] for i in range(n)] if n > 1 else [None]
    follower = [ids[(i - 1) % n] for i in range(n)] if n > 1 else [None]
    veh = StubVehicle(ids, speeds, headways, leader, follower)
    net = SimpleNamespace(edge_length=lambda e: 57.5)
    return SimpleNamespace(k=SimpleNamespace(vehicle=veh, network=net), sim_step=dt,
                           env_params=SimpleNamespace(additional_params=additional or {})), ids


CONTROLLERS = {
    # name: (class, kwargs, SumoCarFollowingParams kwargs) -- the reference's own test setups
    # (tests/fast_tests/test_controllers.py:28-45, 83-94, 132-149, 188-199, 237-246, 471-480,
    #  543-552, 667-682, 720-738) plus class defaults
    "CFM_test": (fc.CFMController, dict(time_delay=0, k_d=1, k_v=1, k_c=1, d_des=1, v_des=8, noise=0),
                 dict(accel=20, decel=5)),
    "BCM_test": (fc.BCMController, dict(time_delay=0, k_d=1, k_v=1, k_c=1, d_des=1, v_des=8, noise=0),
                 dict(accel=15, decel=5)),
    "OVM_test": (fc.OVMController, dict(time_delay=0, alpha=1, beta=1, h_st=2, h_go=15, v_max=30, noise=0),
                 dict(accel=15, decel=5)),
    "LinearOVM_test": (fc.LinearOVM, dict(time_delay=0, v_max=30, adaptation=0.65, h_st=5, noise=0),
                       dict(accel=15, decel=5)),
    "IDM_test": (fc.IDMController, dict(v0=30, b=1.5, delta=4, s0=2, noise=0),
                 dict(tau=1, accel=1, decel=5)),
    "IDM_default": (fc.IDMController, dict(), dict()),
    "FollowerStopper_test": (fc.FollowerStopper, dict(v_des=7.5), dict(accel=20, decel=5)),
    "NonLocalFollowerStopper_test": (fc.NonLocalFollowerStopper, dict(v_des=7.5), dict(accel=20, decel=5)),
    "LAC_test": (fc.LACController, dict(time_delay=0, k_1=0.3, k_2=0.4, h=1, tau=0.1, noise=0),
                 dict(accel=15, decel=5)),
    "Gipps_test": (fc.GippsController, dict(v0=30, acc=1.5, b=-1, b_l=-1, s0=2, tau=1, delay=0, noise=0),
                   dict(accel=15, decel=5)),
    "PISaturation_test": (fc.PISaturation, dict(), dict(accel=20, decel=5)),
}

# (speeds, headways, expected in the reference's own tests)
KNOWN = {
    "CFM_test": ([0] * 5, [5, 10, 15, 20, 25], [12., 17., 22., 27., 32.]),                     # :61-70
    "BCM_test": ([0] * 5, [5, 10, 15, 20, 25], [-12., 13., 13., 13., 13.]),                    # :110-119
    "OVM_test": ([0] * 5, [0, 10, 5, 5, 5], [0., 20.319073, 3.772339, 3.772339, 3.772339]),    # :165-174
    "LinearOVM_test": ([0] * 5, [5, 10, 10, 15, 0], [0., 12.992308, 12.992308, 25.984615, 0.]),  # :215-224
    "IDM_test": ([0] * 5, [10, 20, 30, 40, 50], [0.96, 0.99, 0.995556, 0.9975, 0.9984]),        # :262-271
    "FollowerStopper_test": ([5, 7.5, 7.5, 8, 7], [5, 10, 15, 20, 25], [0, 0, 0, -5, 5]),       # :496-507
    "NonLocalFollowerStopper_test": ([5, 7.5, 7.5, 8, 7], [5, 10, 15, 20, 25],
                                     [-3.33333333333333, -5.0, -5.0, -10.0, 0.0]),            # :568-579
    "PISaturation_test": ([5, 7.5, 7.5, 8, 7], [5, 10, 15, 20, 25],
                          [20., -36.847826, -35.76087, -37.173913, -31.086957]),              # :643-654
    "LAC_test": ([0] * 5, [5, 10, 15, 20, 25], [0., 1.5, 3., 4.5, 6.]),                         # :698-707
    "Gipps_test": ([0] * 5, [2, 4, 6, 8, 10], [0., 5.929271, 5.929271, 5.929271, 5.929271]),   # :754-763
}


def make_controllers(name, ids, **extra):
    cls, kw, cf = CONTROLLERS[name]
    kw = dict(kw)
    kw.update(extra)
    return [cls(i, car_following_params=SumoCarFollowingParams(**cf), **kw) for i in ids]


def run_controllers(name, speeds, headways, dt=0.1, **extra):
    env, ids = ring_env(speeds, headways, dt)
    ctrls = make_controllers(name, ids, **extra)
    out = []
    for c in ctrls:
        a = c.get_action(env)
        out.append(None if a is None else float(a))
    return out


def gen_controllers(rng):
    doc = {"known": {}, "random": {}}
    for name, (sp, hw, exp) in KNOWN.items():
        got = run_controllers(name, sp, hw)
        doc["known"][name] = {"speeds": sp, "headways": hw, "reference_test_expected": exp,
                              "reference_output": got}
    for name in CONTROLLERS:
        if name == "PISaturation_test":
            continue
        cases = []
        for _ in range(24):
            n = int(rng.integers(2, 9))
            sp = rng.uniform(0, 25, n).round(6).tolist()
            hw = rng.uniform(0.0005, 60, n).round(6).tolist()
            if rng.random() < 0.2:
                hw[int(rng.integers(0, n))] = 0.0
            cases.append({"speeds": sp, "headways": hw, "out": run_controllers(name, sp, hw)})
        doc["random"][name] = cases
    # PISaturation over a short trajectory (stateful)
    env, ids = ring_env([5, 7.5, 7.5, 8, 7], [5, 10, 15, 20, 25])
    ctrls = make_controllers("PISaturation_test", ids)
    traj = []
    for t in range(6):
        sp = (np.array([5, 7.5, 7.5, 8, 7]) + 0.3 * t).tolist()
        hw = (np.array([5, 10, 15, 20, 25]) - 0.5 * t).tolist()
        env.k.vehicle.speed = dict(zip(ids, sp))
        env.k.vehicle.headway = dict(zip(ids, hw))
        traj.append({"speeds": sp, "headways": hw, "out": [float(c.get_action(env)) for c in ctrls]})
    doc["pisaturation_traj"] = traj
    # LAC over a short trajectory (stateful self.a)
    env, ids = ring_env([0] * 5, [5, 10, 15, 20, 25])
    ctrls = make_controllers("LAC_test", ids)
    traj = []
    for t in range(5):
        sp = (np.array([0, 1, 2, 3, 4]) + 0.2 * t).tolist()
        hw = (np.array([5, 10, 15, 20, 25]) + 0.4 * t).tolist()
        env.k.vehicle.speed = dict(zip(ids, sp))
        env.k.vehicle.headway = dict(zip(ids, hw))
        traj.append({"speeds": sp, "headways": hw, "out": [float(c.get_action(env)) for c in ctrls]})
    doc["lac_traj"] = traj
    return doc


def gen_failsafes(rng):
    doc = {"cases": []}
    for fs in ("instantaneous", "safe_velocity"):
        for delay in (0.0, 0.5, 1.0):
            for _ in range(12):
                n = int(rng.integers(2, 9))
                sp = rng.uniform(0, 20, n).round(6).tolist()
                hw = rng.uniform(0.01, 12, n).round(6).tolist()
                env, ids = ring_env(sp, hw)
                ctrls = [fc.IDMController(i, car_following_params=SumoCarFollowingParams(),
                                          fail_safe=fs, time_delay=delay) for i in ids]
                raw = [float(c.get_accel(env)) for c in ctrls]
                out = [float(c.get_action(env)) for c in ctrls]
                doc["cases"].append({"fail_safe": fs, "delay": delay, "speeds": sp, "headways": hw,
                                     "raw_idm": raw, "out": out})
    # single vehicle: all actions are safe (base_controller.py:141-142, 191-193)
    env, ids = ring_env([3.0], [1000.0])
    for fs in ("instantaneous", "safe_velocity"):
        c = fc.IDMController(ids[0], car_following_params=SumoCarFollowingParams(), fail_safe=fs)
        doc["cases"].append({"fail_safe": fs, "delay": 0.0, "speeds": [3.0], "headways": [1000.0],
                             "raw_idm": [float(c.get_accel(env))], "out": [float(c.get_action(env))]})
    return doc


def gen_rewards(rng):
    doc = {"desired_velocity": [], "known": {}}
    for _ in range(40):
        n = int(rng.integers(1, 40))
        vel = rng.uniform(0, 20, n).round(6)
        if rng.random() < 0.1:
            vel[0] = -150.0
        tv = float(rng.choice([5, 8, 10, 20, 12.5]))
        fail = bool(rng.random() < 0.1)
        env, ids = ring_env(vel.tolist(), [10.0] * n, additional={"target_velocity": tv})
        doc["desired_velocity"].append({"vel": vel.tolist(), "target_velocity": tv, "fail": fail,
                                        "out": float(fr.desired_velocity(env, fail=fail)),
                                        "average_velocity": float(fr.average_velocity(env, fail=fail))})
    # tests/fast_tests/test_rewards.py:44-45: 10 stopped vehicles... target 10 -> 1 - sqrt(90)/10 when one moves at 10?
    env, ids = ring_env([0.0] * 10, [10.0] * 10, additional={"target_velocity": 10})
    doc["known"]["dv_all_stopped"] = float(fr.desired_velocity(env))
    env.k.vehicle.speed[ids[0]] = 10.0
    doc["known"]["dv_one_at_target"] = float(fr.desired_velocity(env))
    doc["known"]["dv_one_at_target_expected"] = 1 - np.sqrt(90) / 10
    env, ids = ring_env([1.0, 2.0, 3.0], [10.0] * 3)
    env.k.vehicle.prev = dict(zip(ids, [0.5, 2.5, 3.0]))
    doc["known"]["energy"] = {"speed": [1.0, 2.0, 3.0], "prev": [0.5, 2.5, 3.0],
                              "out": float(fr.energy_consumption(env))}
    return doc


def gen_defaults():
    """Default parameters of the reference's param / controller classes (for the host mirror)."""
    import inspect
    from flow.core import params as P
    doc = {"SPEED_MODES": P.SPEED_MODES, "LC_MODES": P.LC_MODES,
           "SumoCarFollowingParams": {"controller_params": P.SumoCarFollowingParams().controller_params,
                                      "speed_mode": P.SumoCarFollowingParams().speed_mode},
           "SumoLaneChangeParams": {"controller_params": P.SumoLaneChangeParams().controller_params,
                                    "lane_change_mode": P.SumoLaneChangeParams().lane_change_mode}}

    def sig(cls):
        out = {}
        for k, v in inspect.signature(cls.__init__).parameters.items():
            if k in ("self", "kwargs") or v.default is inspect._empty:
                continue
            d = v.default
            out[k] = d if isinstance(d, (int, float, str, bool, type(None))) else repr(d)
        return out
    for cls in (P.SumoParams, P.EnvParams, P.NetParams, P.InitialConfig, P.SimParams):
        doc[cls.__name__] = sig(cls)
    doc["controllers"] = {c.__name__: sig(c) for c in (
        fc.IDMController, fc.CFMController, fc.BCMController, fc.LACController, fc.OVMController,
        fc.LinearOVM, fc.GippsController, fc.FollowerStopper)}
    v = P.VehicleParams()
    v.add("human", acceleration_controller=(fc.IDMController, {}), num_vehicles=3)
    v.add("rl", acceleration_controller=(fc.RLController, {}), num_vehicles=2)
    doc["VehicleParams"] = {"ids": v.ids, "num_vehicles": v.num_vehicles, "num_rl_vehicles": v.num_rl_vehicles,
                            "minGap": v.minGap, "types": [t["veh_id"] for t in v.types]}
    return doc


def gen_inflows():
    """InFlows.add as the reference stores it (flow/core/params.py:1080-1213), for the calls the merge and bottleneck
    experiments make (deprecated spellings included), its error cases, and the vehicle-type dicts of those
    experiments -- the inputs of the open-network spec builder."""
    import warnings
    from flow.core import params as P
    warnings.simplefilter("ignore")
    calls = [
        dict(veh_type="human", edge="inflow_highway", vehs_per_hour=1800, departLane="free", departSpeed=10),
        dict(veh_type="rl", edge="inflow_highway", vehs_per_hour=200, depart_lane="free", depart_speed=10),
        dict(veh_type="human", edge="inflow_merge", vehsPerHour=100, departLane="free", departSpeed=7.5),
        dict(veh_type="human", edge="1", vehs_per_hour=2070.0, departLane="random", departSpeed=10),
        dict(veh_type="human", edge="1", period=3, number=7, begin=5),
        dict(veh_type="human", edge="1", probability=0.25, name="prob"),
    ]
    inflow = P.InFlows()
    for c in calls:
        inflow.add(**dict(c))
    errors = []
    for bad in (dict(), dict(vehs_per_hour=1, period=2), dict(probability=1.5), dict(vehs_per_hour=1, begin=0),
                dict(probability=-0.1)):
        try:
            P.InFlows().add(veh_type="human", edge="e", **bad)
            errors.append({"args": bad, "error": None})
        except Exception as e:                                        # noqa: BLE001
            errors.append({"args": bad, "error": type(e).__name__})
    v = P.VehicleParams()
    v.add(veh_id="human", lane_change_controller=(fc.SimLaneChangeController, {}),
          routing_controller=(fc.ContinuousRouter, {}),
          car_following_params=P.SumoCarFollowingParams(speed_mode="all_checks"),
          lane_change_params=P.SumoLaneChangeParams(lane_change_mode=0), num_vehicles=1)
    v.add(veh_id="followerstopper", acceleration_controller=(fc.RLController, {}),
          car_following_params=P.SumoCarFollowingParams(speed_mode=9),
          lane_change_params=P.SumoLaneChangeParams(lane_change_mode=1621), num_vehicles=1)
    types = {}
    for name, tp in v.type_parameters.items():
        types[name] = {"acceleration_controller": tp["acceleration_controller"][0].__name__,
                       "lane_change_controller": tp["lane_change_controller"][0].__name__,
                       "speed_mode": tp["car_following_params"].speed_mode,
                       "controller_params": tp["car_following_params"].controller_params,
                       "lane_change_mode": tp["lane_change_params"].lane_change_mode,
                       "initial_speed": tp["initial_speed"]}
    return {"calls": calls, "flows": inflow.get(), "errors": errors, "vehicle_types": types,
            "initial": [{k: (t[k] if not isinstance(t[k], tuple) else t[k][0].__name__)
                         for k in ("veh_id", "num_vehicles", "initial_speed")} for t in v.initial]}


def copy_flow_params():
    """The stored flow_params files the reference's own tests hold (data, not code):
    tests/fast_tests/test_files/ring_230.json and merge.json."""
    import shutil
    for src, dst in (("ring_230.json", "ring_230_flow_params.json"), ("merge.json", "merge_flow_params.json")):
        shutil.copyfile(os.path.join(REF, "tests/fast_tests/test_files", src), os.path.join(HERE, dst))


def copy_emission():
    src = os.path.join(REF, "tests/fast_tests/test_files/ring_230_emission.csv")
    keep = ["time", "id", "edge_id", "relative_position", "speed", "lane_number"]
    with open(src) as f, open(os.path.join(HERE, "ring_230_emission.csv"), "w", newline="") as g:
        rd = csv.DictReader(f)
        wr = csv.DictWriter(g, fieldnames=keep)
        wr.writeheader()
        for row in rd:
            wr.writerow({k: row[k] for k in keep})


def main():
    rng = np.random.default_rng(20261003)
    for name, fn in (("controllers", gen_controllers), ("failsafes", gen_failsafes),
                     ("rewards", gen_rewards), ("defaults", lambda r: gen_defaults()),
                     ("inflows", lambda r: gen_inflows())):
        with open(os.path.join(HERE, name + ".json"), "w") as f:
            json.dump(fn(rng), f, indent=1)
        print("wrote", name)
    copy_emission()
    print("wrote ring_230_emission.csv")
    copy_flow_params()
    print("wrote ring_230_flow_params.json, merge_flow_params.json")


if __name__ == "__main__":
    main()
