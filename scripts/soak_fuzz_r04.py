"""One-off soak of round 4's other new paths with random cases beyond the pinned tests:
  * the multi-agent ring heads of k_ring_pair (random RL slots / columns, noise, several action columns in the group form)
    against the generic kernel bit for bit and, without noise, against the oracle;
  * BottleneckAccelEnv with RL vehicles (random seeds, inflow rates, lane-change modes) against oracle/bottleneck_accel.py.
    python scripts/soak_fuzz_r04.py [first seed] [count]"""
import os
import sys
import traceback

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))


def ma_ring_case(seed):
    import test_multiagent_ring_gpu as tm
    from oracle import refsim as S
    rng = np.random.default_rng(7000 + seed)
    env = [S.ENV_WAVE_ATTENUATION_PO_MA, S.ENV_ACCEL_PO_MA][int(rng.integers(0, 2))]
    n_rl = int(rng.integers(1, 7))
    rl_slots = tuple(sorted(rng.choice(22, n_rl, replace=False).tolist()))
    K, R = int(rng.integers(20, 90)), int(rng.integers(1, 12))
    acts = rng.uniform(-1.5, 1.5, (K, R, n_rl)).astype(np.float32)
    noisy = tm.ma_ring_experiment_spec(env, R, rl_slots, noise=0.2, seed=seed)
    noisy["horizon"] = 10 ** 6
    a, oa, ra, da = tm._ma_rollout(noisy, K, acts)
    b, ob, rb, db = tm._ma_rollout(noisy, K, acts, env={"FLOWSIM_FORCE_GENERIC": "1"})
    assert a.last_kernel.startswith("k_ring_pair") and b.last_kernel.startswith("k_steps"), (a.last_kernel, b.last_kernel)
    np.testing.assert_array_equal(oa, ob)
    np.testing.assert_array_equal(ra, rb)
    np.testing.assert_array_equal(a.pos, b.pos)
    a.close(), b.close()
    quiet = tm.ma_ring_experiment_spec(env, R, rl_slots, noise=0.0, seed=seed)
    quiet["horizon"] = 10 ** 6
    c, oc, rc, dc = tm._ma_rollout(quiet, K, acts)
    ora = S.RingOracle(quiet, np.float32)
    ora.reset()
    for k in range(K):
        o_ref, r_ref, _ = ora.step(acts[k])
        np.testing.assert_array_equal(oc[k], o_ref.astype(np.float32))
        np.testing.assert_array_equal(rc[k], r_ref.astype(np.float32))
    c.close()
    return "ma ring env %d rl %s K %d R %d" % (env, rl_slots, K, R)


def accel_case(seed):
    import test_bottleneck_env_gpu as tb
    from flow_amd import _lib as L
    from flow_amd.controllers import RLController
    from flow_amd.core.params import (EnvParams, InFlows, InitialConfig, NetParams, SumoLaneChangeParams, SumoParams,
                                      VehicleParams)
    from flow_amd.envs import BottleneckAccelEnv
    from flow_amd.networks import BottleneckNetwork
    rng = np.random.default_rng(8000 + seed)
    n_rl = int(rng.integers(1, 6))
    vehicles = VehicleParams()
    vehicles.add(veh_id="human", num_vehicles=int(rng.integers(2, 9)))
    vehicles.add(veh_id="rl", acceleration_controller=(RLController, {}),
                 lane_change_params=SumoLaneChangeParams(lane_change_mode=[512, 0][seed % 2]), num_vehicles=n_rl)
    inflow = InFlows()
    inflow.add(veh_type="human", edge="1", vehs_per_hour=int(rng.integers(600, 2600)), departLane="random", departSpeed=10)
    add = {"max_accel": 3, "max_decel": 3, "lane_change_duration": 5, "disable_tb": True, "disable_ramp_metering": True,
           "target_velocity": 30, "add_rl_if_exit": True}
    net = BottleneckNetwork(name="bottleneck", vehicles=vehicles, initial_config=InitialConfig(),
                            net_params=NetParams(inflows=inflow, additional_params={"scaling": 1, "speed_limit": 23}))
    slots = [64, 100][int(rng.integers(0, 2))]
    env = BottleneckAccelEnv(EnvParams(horizon=10 ** 6, additional_params=add),
                             SumoParams(sim_step=0.5, seed=int(rng.integers(1, 10 ** 6)), max_vehicles=slots), net)
    ora = tb.accel_oracle(env)
    np.testing.assert_allclose(env.reset(), (ora.reset(), ora.accel_state(0))[1], rtol=0, atol=1e-12)
    readded = 0
    for k in range(400):
        a = rng.uniform(-1, 1, 2 * n_rl) * np.tile([3.0, 1.4], n_rl)
        if k % 2:
            a[1::2] = 0.0
        before = set(env.k.vehicle.get_rl_ids())
        obs, rew, done, _ = env.step(a)
        o_ref, r_ref, d_ref = ora.step(a[None, :])
        np.testing.assert_allclose(obs, o_ref[0], rtol=0, atol=1e-12, err_msg="obs, step %d" % k)
        np.testing.assert_allclose(rew, r_ref[0], rtol=1e-12, atol=1e-12)
        for f, ref in ((L.FS_FIELD_POS, ora.x), (L.FS_FIELD_VEL, ora.v), (L.FS_FIELD_ROUTE, ora.route)):
            got = env.sim.get_state(f)[0]
            alive = ora.route[0] >= 0
            np.testing.assert_array_equal(got[alive], ref[0][alive], err_msg="field %d, step %d" % (f, k))
        readded += len(set(env.k.vehicle.get_rl_ids()) - before)
    env.terminate()
    return "accel env rl %d slots %d re-added %d" % (n_rl, slots, readded)


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    bad = []
    for seed in range(first, first + count):
        for name, fn in (("ma_ring", ma_ring_case), ("accel", accel_case)):
            try:
                print("seed", seed, fn(seed), flush=True)
            except Exception:
                bad.append((name, seed))
                traceback.print_exc()
    print("FAILED:" if bad else "all ok", bad)
    sys.exit(1 if bad else 0)
