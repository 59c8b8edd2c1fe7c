"""GPU parity of the open-network path (FS_NET_MERGE, `k_steps_open`) against oracle/opennet.py on the
same seeded inputs, through the C ABI.

Bars: float32 kernel vs the float32 oracle twin -- bit-exact on every state field, observation, reward
and done flag for noise-free configurations (the only libm call on this path is the Gaussian noise);
float64 kernel vs the float64 oracle (the reference's arithmetic type) -- 1e-9.
"""
import numpy as np
import pytest

from conftest import seeds

from helpers import idm_vehicle, merge_spec
from oracle import opennet as O
from oracle import refsim as S

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def slot_order_kernels(monkeypatch):
    """This module holds the SLOT-order open-network kernels (k_steps_open / k_steps_wide) to the oracle; the queue-order
    kernels that take the same configurations by default have tests of their own (test_queue_gpu.py, test_dropq_gpu.py)."""
    monkeypatch.setenv("FLOWSIM_NO_QUEUE", "1")


def make(spec, precision):
    from flow_amd.sim import FlowSim
    return FlowSim(spec, precision=precision)


def quiet(spec):
    """The same spec with the acceleration noise switched off (bit-exact comparisons)."""
    spec = dict(spec)
    spec["vehicles"] = [dict(v, noise=0.0) for v in spec["vehicles"]]
    return spec


def compare_state(sim, ora, exact=True, atol=0.0):
    from flow_amd import _lib as L
    alive = ora.alive
    route = sim.get_state(L.FS_FIELD_ROUTE)
    np.testing.assert_array_equal(route, ora.route)
    cmp = (lambda a, b: np.testing.assert_array_equal(a[alive], b[alive])) if exact else \
        (lambda a, b: np.testing.assert_allclose(a[alive], b[alive], rtol=0, atol=atol))
    cmp(sim.pos, ora.x)
    cmp(sim.vel, ora.v)
    cmp(sim.get_state(L.FS_FIELD_PREV_VEL), ora.prev_v)
    cmp(sim.headway, ora.h)
    for field, ref in ((L.FS_FIELD_SEQ, ora.seq), (L.FS_FIELD_ORIGIN, ora.origin), (L.FS_FIELD_LEADER, ora.lead),
                       (L.FS_FIELD_FOLLOWER, ora.foll)):
        got = sim.get_state(field)
        np.testing.assert_array_equal(got[alive], ref[alive])
    np.testing.assert_array_equal(sim.get_state(L.FS_FIELD_CTL_SEQ), ora.ctl_seq)
    np.testing.assert_array_equal(sim.get_state(L.FS_FIELD_ARRIVED_RL), ora.arrived_rl.astype(np.int32))
    cnt = sim.get_state(L.FS_FIELD_COUNTERS)
    np.testing.assert_array_equal(cnt[:, 0], ora.sim_steps)
    np.testing.assert_array_equal(cnt[:, 1], ora.seq_ctr)
    np.testing.assert_array_equal(cnt[:, 2], ora.ctl_ctr)
    np.testing.assert_array_equal(cnt[:, 3], ora.num_arrived)
    np.testing.assert_array_equal(cnt[:, 4], ora.num_departed)
    np.testing.assert_array_equal(cnt[:, 5], ora.total_arrived)
    np.testing.assert_array_equal(cnt[:, 6], ora.total_departed)
    np.testing.assert_array_equal(cnt[:, 7], ora.total_dropped)


def run_pair(spec, precision, steps, action_fn=None, check_every=10, exact=True, atol=0.0):
    dtype = np.float32 if precision == "f32" else np.float64
    ora = O.MergeOracle(spec, dtype)
    sim = make(spec, precision)
    cmp = np.testing.assert_array_equal if exact else (lambda a, b: np.testing.assert_allclose(a, b, rtol=0, atol=atol))
    cmp(sim.reset(), ora.reset().astype(np.float32))
    compare_state(sim, ora, exact, atol)
    for k in range(steps):
        a = None if action_fn is None else action_fn(k)
        o_ref, r_ref, d_ref = ora.step(a)
        o_gpu, r_gpu, d_gpu = sim.step(a)
        cmp(o_gpu, o_ref.astype(np.float32))
        cmp(r_gpu, r_ref.astype(np.float32))
        np.testing.assert_array_equal(d_gpu, d_ref)
        if k % check_every == 0 or k == steps - 1:
            compare_state(sim, ora, exact, atol)
    np.testing.assert_array_equal(sim.time_counter, ora.time_counter)
    sim.close()
    return ora


def uniform_actions(spec, seed, lo=-1.0, hi=1.5):
    rng = np.random.default_rng(seed)
    R, A = spec["num_replicas"], spec["num_rl"]
    return lambda k: rng.uniform(lo, hi, (R, A)).astype(np.float32)


def test_merge_po_f32_bit_exact_with_inflows_arrivals_and_rl_queue():
    spec = quiet(merge_spec(R=5, cap_human=26, cap_rl=5, num_rl=2, horizon=500, seed=3))
    ora = run_pair(spec, "f32", 500, uniform_actions(spec, 7, 0.0, 1.5))
    # the run exercised the whole life cycle
    assert ora.total_departed.min() > 20 and ora.total_arrived.min() > 5
    assert (ora.ctl_ctr > 2).all()                      # RL vehicles joined and left rl_veh


def test_merge_po_f32_small_capacity_several_replicas_per_wave():
    """cap 8 -> SEG 8, eight replicas per wave; the slot pools overflow (vehicles wait outside)."""
    spec = quiet(merge_spec(R=19, cap_human=6, cap_rl=2, num_rl=1, horizon=300, seed=5, pre=120.0, q_highway=1500.0))
    run_pair(spec, "f32", 300, uniform_actions(spec, 11))


def test_merge_po_f32_sims_per_step_and_warmup():
    spec = quiet(merge_spec(R=4, cap_human=12, cap_rl=4, num_rl=3, horizon=60, seed=9, sims_per_step=5,
                            warmup_steps=7))
    run_pair(spec, "f32", 60, uniform_actions(spec, 2, 0.2, 1.5), check_every=5)


def test_merge_po_no_actions_means_sumo_drives_the_rl_vehicles():
    spec = quiet(merge_spec(R=3, cap_human=12, cap_rl=4, num_rl=2, horizon=200, seed=1))
    run_pair(spec, "f32", 200, None)


def test_merge_po_f64_matches_reference_arithmetic():
    spec = quiet(merge_spec(R=3, cap_human=20, cap_rl=4, num_rl=2, horizon=400, seed=4))
    run_pair(spec, "f64", 400, uniform_actions(spec, 5, 0.0, 1.5), check_every=25, exact=False, atol=1e-9)


def test_merge_po_noise_short_horizon_tolerance():
    """With the Gaussian acceleration noise (libm log / cos on both sides) the float32 paths agree to 1e-4
    over a horizon short enough that no discrete event (insertion, yield) flips."""
    spec = merge_spec(R=4, cap_human=12, cap_rl=4, num_rl=2, horizon=40, seed=8)
    run_pair(spec, "f32", 40, uniform_actions(spec, 3, 0.0, 1.0), exact=False, atol=1e-4)


def test_merge_multiagent_as_shipped_and_with_actions_applied():
    for apply in (False, True):
        spec = quiet(merge_spec(R=4, cap_human=14, cap_rl=4, num_rl=4, horizon=300, seed=6, env=O.ENV_MERGE_MA,
                                ma_apply_actions=apply))
        rng = np.random.default_rng(13)

        def acts(k, rng=rng):
            a = rng.uniform(0.0, 1.5, (4, 4)).astype(np.float32)
            a[rng.random((4, 4)) < 0.2] = np.nan          # "the vehicle just entered": no action
            return a
        run_pair(spec, "f32", 300, acts)


def test_merge_other_controllers_and_failsafes():
    spec = quiet(merge_spec(R=3, cap_human=12, cap_rl=3, num_rl=2, horizon=250, seed=12))
    veh = spec["vehicles"]
    for i in range(12):
        kind = i % 4
        if kind == 1:
            veh[i] = idm_vehicle(controller=S.CTRL_FOLLOWER_STOPPER, p=[12.0] + [0] * 7, speed_mode=1, type=0)
        elif kind == 2:
            veh[i] = idm_vehicle(controller=S.CTRL_GIPPS, p=[30, 1.5, -1, -1, 2, 1, 0, 0], speed_mode=1, type=0,
                                 fail_safe=S.FAILSAFE_SAFE_VELOCITY)
        elif kind == 3:
            veh[i] = idm_vehicle(controller=S.CTRL_SIM, type=0)
    run_pair(spec, "f32", 250, uniform_actions(spec, 21, 0.0, 1.5))


def test_merge_masked_reset_restarts_only_the_selected_replicas():
    spec = quiet(merge_spec(R=6, cap_human=12, cap_rl=3, num_rl=2, horizon=100, seed=2))
    ora = O.MergeOracle(spec, np.float32)
    sim = make(spec, "f32")
    np.testing.assert_array_equal(sim.reset(), ora.reset().astype(np.float32))
    acts = uniform_actions(spec, 4, 0.0, 1.5)
    for k in range(80):
        a = acts(k)
        ora.step(a), sim.step(a)
    mask = np.array([1, 0, 0, 1, 0, 1], dtype=bool)
    np.testing.assert_array_equal(sim.reset(mask), ora.reset(mask).astype(np.float32))
    compare_state(sim, ora)
    for k in range(60):
        a = acts(k)
        o_ref, r_ref, d_ref = ora.step(a)
        o_gpu, r_gpu, d_gpu = sim.step(a)
        np.testing.assert_array_equal(o_gpu, o_ref.astype(np.float32))
        np.testing.assert_array_equal(r_gpu, r_ref.astype(np.float32))
        np.testing.assert_array_equal(d_gpu, d_ref)
    compare_state(sim, ora)
    sim.close()


def test_merge_rollout_equals_stepping():
    """K steps in one launch == K single-step launches (same kernel, state kept in registers)."""
    import torch
    spec = quiet(merge_spec(R=8, cap_human=20, cap_rl=4, num_rl=2, horizon=200, seed=14))
    K, R = 120, 8
    rng = np.random.default_rng(3)
    acts = rng.uniform(0.0, 1.5, (K, R, 2)).astype(np.float32)
    a = make(spec, "f32")
    b = make(spec, "f32")
    a.reset(), b.reset()
    dev = torch.device("cuda:0")
    obs = torch.empty((K, R, a.obs_dim), dtype=torch.float32, device=dev)
    rew = torch.empty((K, R), dtype=torch.float32, device=dev)
    done = torch.empty((K, R), dtype=torch.uint8, device=dev)
    a.rollout_dev(K, obs, rew, done, actions=torch.from_numpy(acts).to(dev))
    a.sync()
    for k in range(K):
        o, r, d = b.step(acts[k])
        np.testing.assert_array_equal(obs[k].cpu().numpy(), o)
        np.testing.assert_array_equal(rew[k].cpu().numpy(), r)
        np.testing.assert_array_equal(done[k].cpu().numpy().astype(bool), d)
    np.testing.assert_array_equal(a.pos, b.pos)
    a.close(), b.close()


# ------------------------------------------------------------------ lane drops (BottleneckNetwork)
def bottleneck_actions(spec, seed, lo=-1.0, hi=1.0):
    rng = np.random.default_rng(seed)
    R, A = spec["num_replicas"], spec["num_rl"]
    return lambda k: rng.uniform(lo, hi, (R, A)).astype(np.float32)


def compare_vmax(sim, ora):
    from flow_amd import _lib as L
    got = sim.get_state(L.FS_FIELD_MAX_SPEED)
    np.testing.assert_array_equal(got[ora.alive], ora.vmax[ora.alive])


def test_merge_po_rl_veh_survives_reset_and_skips_while_removing_bit_exact():
    # O2 (merge.py:189-231): reset() never clears rl_veh -- the next episode starts with the old entries as ghost rows
    # of error values, removed by a loop that skips the entry behind each one it removes -- kernel and oracle agree bit
    # for bit over three episodes (tests/test_open_cpu.py holds the oracle to the reference's literal list operations)
    spec = quiet(merge_spec(R=7, cap_human=10, cap_rl=8, num_rl=4, horizon=10 ** 6, seed=5, q_rl=1500.0, q_highway=600.0))
    ora = O.MergeOracle(spec, np.float32)
    sim = make(spec, "f32")
    act = uniform_actions(spec, 3, lo=-0.5, hi=1.0)
    np.testing.assert_array_equal(sim.reset(), ora.reset().astype(np.float32))
    ghost_steps = 0
    for episode in range(3):
        for k in range(200):
            a = act(k)
            o_ref, r_ref, d_ref = ora.step(a)
            o_gpu, r_gpu, d_gpu = sim.step(a)
            np.testing.assert_array_equal(o_gpu, o_ref.astype(np.float32), err_msg="episode %d step %d" % (episode, k))
            np.testing.assert_array_equal(r_gpu, r_ref.astype(np.float32))
            if episode > 0 and k < 4:
                ghost_steps += int((o_ref[:, 0::5] < -30).any())
        compare_state(sim, ora)
        assert ((ora.ctl_seq >= 0).sum(axis=1) == 4).all()
        o_reset = ora.reset().astype(np.float32)
        np.testing.assert_array_equal(sim.reset(), o_reset)
        assert (o_reset[:, 0::5] < -30).all()                      # four stale entries: four rows of error values
        compare_state(sim, ora)
    assert ghost_steps >= 2                                        # the skipped entries outlive the first pass
    sim.close()


def with_probabilistic_inflows(spec, probs, number=None, window=None):
    """The spec with inflow f given a per-second probability probs[f] instead of its period (None keeps it)."""
    spec = dict(spec)
    flows = []
    for f, fl in enumerate(spec["inflows"]):
        fl = dict(fl)
        if f < len(probs) and probs[f] is not None:
            fl["probability"] = float(probs[f])
            fl.pop("period", None)
            if number is not None:
                fl["number"] = int(number)
            if window is not None:
                fl["begin"], fl["end"] = window
        flows.append(fl)
    spec["inflows"] = flows
    return spec


def test_merge_probabilistic_inflows_f32_bit_exact_incl_reset_and_limits():
    # InFlows.add(probability=p) (params.py:1103-1105): one Philox trial per flow and sub-step; a mix of a
    # probabilistic highway flow, a deterministic RL flow and a probabilistic on-ramp flow with a vehicle count limit
    base = quiet(merge_spec(R=9, cap_human=14, cap_rl=4, num_rl=3, horizon=260, seed=6))
    spec = with_probabilistic_inflows(base, [0.45, None, 0.2])
    spec["inflows"][2]["number"] = 4
    ora = run_pair(spec, "f32", 260, uniform_actions(spec, 4))
    assert ora.generated[:, 0].min() > 5 and (ora.generated[:, 2] <= 4).all() and ora.generated[:, 2].max() == 4
    assert len(set(ora.generated[:, 0].tolist())) > 1                   # replicas draw their own sequences
    # a second episode of the same handle draws a different sequence (episode-keyed), the oracle alike
    sim = make(spec, "f32")
    o2 = O.MergeOracle(spec, np.float32)
    sim.reset(), o2.reset()
    for _ in range(40):
        sim.step(None), o2.step(None)
    first = sim.get_state(__import__("flow_amd")._lib.FS_FIELD_COUNTERS)[:, 6].copy()
    np.testing.assert_array_equal(sim.reset(), o2.reset().astype(np.float32))
    for _ in range(40):
        og, _, _ = sim.step(None)
        orf, _, _ = o2.step(None)
    np.testing.assert_array_equal(og, orf.astype(np.float32))
    second = sim.get_state(__import__("flow_amd")._lib.FS_FIELD_COUNTERS)[:, 6]
    assert (first != second).any()
    sim.close()


def test_probabilistic_inflow_window_and_rollout_equals_stepping():
    import torch
    base = quiet(merge_spec(R=6, cap_human=14, cap_rl=4, num_rl=2, horizon=150, seed=8))
    spec = with_probabilistic_inflows(base, [1.0, None, None], window=(3.0, 9.0))     # p = 1: a trial succeeds iff in the window
    ora = run_pair(spec, "f32", 150, uniform_actions(spec, 2))
    # trials at now = 3.0 .. 9.0 in steps of 0.2 s, each with p * dt = 0.2
    assert 0 < ora.generated[:, 0].min() and ora.generated[:, 0].max() <= 31
    a, b = make(spec, "f32"), make(spec, "f32")
    a.reset(), b.reset()
    for _ in range(60):
        a.step(None)
    dev = torch.device("cuda", 0)
    o = torch.empty((60, 6, a.obs_dim), device=dev)
    r = torch.empty((60, 6), device=dev)
    d = torch.empty((60, 6), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    b.rollout_dev(60, o, r, d)
    b.sync()
    np.testing.assert_array_equal(a.pos, b.pos)
    np.testing.assert_array_equal(a.get_state(__import__("flow_amd")._lib.FS_FIELD_ROUTE), b.get_state(__import__("flow_amd")._lib.FS_FIELD_ROUTE))
    a.close(), b.close()


def test_bottleneck_probabilistic_random_lane_inflows_both_kernels():
    # the lane-drop network: 64-slot kernel and the wide kernel (one workgroup per replica), random entry lanes
    from helpers import bottleneck_spec
    for caps in ((48, 8), (90, 10)):
        spec = bottleneck_spec(R=3, cap_human=caps[0], cap_rl=caps[1], horizon=220, seed=11, q=2300.0)
        spec = with_probabilistic_inflows(spec, [0.55, 0.08])
        ora = run_pair(spec, "f32", 220, bottleneck_actions(spec, 3))
        assert ora.generated[:, 0].min() > 10 and ora.total_departed.min() > 10


def test_bottleneck_desired_velocity_f32_bit_exact():
    from helpers import bottleneck_spec
    spec = bottleneck_spec(R=5, cap_human=48, cap_rl=8, horizon=500, seed=3)
    ora = run_pair(spec, "f32", 500, bottleneck_actions(spec, 5))
    assert ora.total_arrived.min() > 50 and (ora.alive.sum(axis=1) > 25).all()
    assert (ora.vmax[:, 48:][ora.alive[:, 48:]] < 23.0).any()          # the actions moved some maxSpeed


def test_bottleneck_state_fields_and_warmup():
    from helpers import bottleneck_spec
    spec = bottleneck_spec(R=3, cap_human=40, cap_rl=6, horizon=120, seed=7, warmup_steps=40)
    dtype = np.float32
    ora = O.MergeOracle(spec, dtype)
    sim = make(spec, "f32")
    np.testing.assert_array_equal(sim.reset(), ora.reset().astype(np.float32))
    acts = bottleneck_actions(spec, 2)
    for k in range(120):
        a = acts(k)
        o_ref, r_ref, d_ref = ora.step(a)
        o_gpu, r_gpu, d_gpu = sim.step(a)
        np.testing.assert_array_equal(o_gpu, o_ref.astype(np.float32))
        np.testing.assert_array_equal(r_gpu, r_ref.astype(np.float32))
        np.testing.assert_array_equal(d_gpu, d_ref)
        if k % 20 == 0:
            compare_state(sim, ora)
            compare_vmax(sim, ora)
    assert d_ref.all()
    sim.close()


def test_bottleneck_base_env_and_no_actions():
    from helpers import bottleneck_spec
    spec = bottleneck_spec(R=3, cap_human=44, cap_rl=4, horizon=200, seed=9, env=O.ENV_BOTTLENECK, num_rl=0,
                           action_cells=[])
    ora = run_pair(spec, "f32", 200, None)
    assert ora.total_arrived.min() > 10


def test_bottleneck_f64_matches_reference_arithmetic():
    from helpers import bottleneck_spec
    spec = bottleneck_spec(R=2, cap_human=44, cap_rl=6, horizon=300, seed=4)
    run_pair(spec, "f64", 300, bottleneck_actions(spec, 8), check_every=25, exact=False, atol=1e-9)


def test_bottleneck_without_zipper_lookahead_crashes_at_the_joins():
    """zipper_distance = 0: vehicles only see the other lane once they are on it -- side-by-side arrivals at a join
    are collisions (the crash rule needs one physical lane), and the run stays bit-identical to the oracle."""
    from helpers import bottleneck_spec
    spec = bottleneck_spec(R=4, cap_human=48, cap_rl=4, horizon=400, seed=11, zipper_distance=0.0)
    ora = O.MergeOracle(spec, np.float32)
    sim = make(spec, "f32")
    np.testing.assert_array_equal(sim.reset(), ora.reset().astype(np.float32))
    crashed = np.zeros(4, dtype=bool)
    for k in range(400):
        o_ref, r_ref, d_ref = ora.step(None)
        o_gpu, r_gpu, d_gpu = sim.step(None)
        np.testing.assert_array_equal(o_gpu, o_ref.astype(np.float32))
        np.testing.assert_array_equal(d_gpu, d_ref)
        crashed |= d_ref & (ora.time_counter < 400)
    assert crashed.any()
    sim.close()


def test_sharded_open_network_handles_reproduce_the_unsharded_run():
    """Random entry lanes (M9) are drawn from the GLOBAL replica id: shards [0, 3) + [3, 8) == one handle of 8."""
    from helpers import bottleneck_spec
    spec = bottleneck_spec(R=8, cap_human=44, cap_rl=6, horizon=150, seed=21)
    whole = make(spec, "f32")
    parts = []
    for lo, hi in ((0, 3), (3, 8)):
        sub = dict(spec, num_replicas=hi - lo, replica_offset=lo)
        for key in ("init_alive", "init_pos", "init_vel", "init_route"):
            sub[key] = np.asarray(spec[key])[lo:hi]
        parts.append(make(sub, "f32"))
    whole.reset()
    [p.reset() for p in parts]
    acts = bottleneck_actions(spec, 6)
    for k in range(150):
        a = acts(k)
        o_w, r_w, d_w = whole.step(a)
        outs = [p.step(a[lo:hi]) for p, (lo, hi) in zip(parts, ((0, 3), (3, 8)))]
        np.testing.assert_array_equal(o_w, np.concatenate([o[0] for o in outs]))
        np.testing.assert_array_equal(r_w, np.concatenate([o[1] for o in outs]))
    from flow_amd import _lib as L
    np.testing.assert_array_equal(whole.get_state(L.FS_FIELD_ROUTE),
                                  np.concatenate([p.get_state(L.FS_FIELD_ROUTE) for p in parts]))
    for s in [whole] + parts:
        s.close()


def test_bottleneck_simplified_lane_changing_f32_bit_exact():
    """M11 on: the humans' lane_change_mode lets SUMO change lanes -> the simplified model runs in the kernel."""
    from flow_amd import _lib as L
    from helpers import bottleneck_spec
    spec = bottleneck_spec(R=5, cap_human=48, cap_rl=8, horizon=400, seed=13, lane_change_cooldown_steps=8,
                           lane_change_min_gain=8.0)
    for v in spec["vehicles"][:48]:
        v["lane_change_mode"] = 1621
    ora = run_pair(spec, "f32", 400, bottleneck_actions(spec, 4))
    assert (ora.num_lane_changes > 40).all()
    spec64 = dict(spec, num_replicas=2)
    for key in ("init_alive", "init_pos", "init_vel", "init_route"):
        spec64[key] = np.asarray(spec[key])[:2]
    run_pair(spec64, "f64", 200, bottleneck_actions(spec64, 4), check_every=25, exact=False, atol=1e-9)


def random_open_spec(seed):
    """A random open-network configuration: merge (two paths) or lane drop (four), random geometry knobs, inflow
    rates, slot pools, controllers, horizons -- everything the kernel has a branch for."""
    from helpers import bottleneck_spec
    rng = np.random.default_rng(1000 + seed)
    R = int(rng.integers(1, 9))
    if seed % 3 != 2:                                                   # ---- merge
        cap_rl = int(rng.integers(1, 6))
        cap_h = int(rng.choice([6, 11, 14, 27, 40])) if cap_rl + 40 <= 64 else 12
        ma = bool(rng.integers(0, 2))
        num_rl = cap_rl if ma else int(rng.integers(1, cap_rl + 1))
        spec = quiet(merge_spec(R=R, cap_human=cap_h, cap_rl=cap_rl, num_rl=num_rl, pre=float(rng.choice([80, 200, 500])),
                                merge=float(rng.choice([60, 100])), post=float(rng.choice([50, 100])),
                                horizon=int(rng.integers(40, 200)), seed=seed,
                                q_highway=float(rng.choice([600, 1500, 2400])), q_rl=float(rng.choice([150, 400, 900])),
                                q_merge=float(rng.choice([100, 400, 900])), n_init=int(rng.integers(0, 4)),
                                env=O.ENV_MERGE_MA if ma else None, time_gap=float(rng.choice([0.5, 1.0, 3.0])),
                                sims_per_step=int(rng.choice([1, 1, 2, 5])), warmup_steps=int(rng.choice([0, 0, 6])),
                                ma_apply_actions=bool(rng.integers(0, 2)), crash_gap=float(rng.choice([0.0, 0.5])),
                                slowdown_ramp=float(rng.choice([1.0, 0.2 / 0.201]))))
        spec["junction"]["enabled"] = int(rng.integers(0, 2))
        veh = spec["vehicles"]
        for i in range(cap_h):
            kind = int(rng.integers(0, 6))
            if kind == 1:
                veh[i] = idm_vehicle(controller=S.CTRL_SIM, speed_mode=int(rng.choice([0, 1, 7, 25, 31])), type=0)
            elif kind == 2:
                veh[i] = idm_vehicle(controller=S.CTRL_FOLLOWER_STOPPER, p=[float(rng.uniform(5, 20))] + [0] * 7,
                                     speed_mode=1, type=0)
            elif kind == 3:
                veh[i] = idm_vehicle(controller=S.CTRL_CFM, p=[1, 1, 1, 1, 8, 0, 0, 0], speed_mode=1, type=0,
                                     fail_safe=S.FAILSAFE_INSTANTANEOUS)
            else:
                veh[i] = idm_vehicle(p=[float(rng.uniform(15, 30)), 1, float(rng.uniform(0.8, 2)), 1.5, 4, 2, 0, 0],
                                     speed_mode=int(rng.choice([0, 1, 3])), type=0)
        A = spec["num_rl"]
        acts = (lambda k, r=np.random.default_rng(seed): r.uniform(-1.0, 1.5, (R, A)).astype(np.float32))
        return spec, acts
    cap_rl = int(rng.integers(2, 9))
    spec = bottleneck_spec(R=R, cap_human=int(rng.integers(33, 57 - cap_rl)), cap_rl=cap_rl,
                           horizon=int(rng.integers(60, 220)), seed=seed, q=float(rng.choice([1200, 2300, 3600])),
                           av_frac=float(rng.choice([0.1, 0.3])), zipper_distance=float(rng.choice([0.0, 20.0, 50.0, 120.0])),
                           warmup_steps=int(rng.choice([0, 0, 20])), lane_change_cooldown_steps=int(rng.choice([2, 8, 20])),
                           lane_change_min_gain=float(rng.choice([3.0, 10.0])), crash_gap=float(rng.choice([0.0, 1.0])))
    if rng.integers(0, 2):
        for v in spec["vehicles"][:int(spec["num_vehicles"]) - cap_rl]:
            v["lane_change_mode"] = 1621
    if rng.integers(0, 3) == 0:
        for v in spec["vehicles"][int(spec["num_vehicles"]) - cap_rl:]:
            v["lane_change_mode"] = 597
    A = spec["num_rl"]
    acts = (lambda k, r=np.random.default_rng(seed): r.uniform(-1.5, 1.5, (R, A)).astype(np.float32))
    return spec, acts


@pytest.mark.parametrize("seed", seeds(range(9), range(9, 18)))
def test_fuzz_random_open_network_configs_bit_exact(seed):
    spec, acts = random_open_spec(seed)
    steps = int(spec["horizon"])
    run_pair(spec, "f32", steps, acts if seed % 5 else None, check_every=max(1, steps // 4))


def test_equal_positions_take_the_exact_ranking_path():
    """Vehicles at exactly the same coordinate (two initial ones side by side; two inflows with the same schedule
    releasing at one coordinate in the same sub-step): the fast 32-bit ranking sees a tie and hands over to the
    64-bit one -- order 'higher slot first', as the oracle sorts."""
    from helpers import bottleneck_spec
    spec = bottleneck_spec(R=3, cap_human=40, cap_rl=8, horizon=160, seed=2)
    X = np.asarray(spec["init_pos"]).copy()
    X[:, 40] = X[:, 0]                                      # the RL vehicle starts level with the human, other lane
    spec["init_pos"] = X
    for f in spec["inflows"]:                               # same clock for both inflows: they release together
        f["period"], f["begin"] = 1.5, 1.0
    probe = O.MergeOracle(spec, np.float32)                 # the scenario really produces equal coordinates
    probe.reset()
    acts = bottleneck_actions(spec, 3)
    tie_steps = 0
    for k in range(160):
        probe.step(acts(k))
        tie_steps += sum(len(np.unique(probe.x[r][probe.alive[r]])) < probe.alive[r].sum() for r in range(3))
    assert tie_steps > 20
    ora = run_pair(spec, "f32", 160, bottleneck_actions(spec, 3), check_every=5)
    assert ora.total_departed.min() > 40
    mspec = quiet(merge_spec(R=3, cap_human=20, cap_rl=6, num_rl=3, horizon=120, seed=5))
    mspec["init_pos"] = np.asarray(mspec["init_pos"]).copy()
    alive0 = np.nonzero(np.asarray(mspec["init_alive"])[0])[0]
    if len(alive0) >= 2:                                    # two initial vehicles on the two routes at one coordinate
        mspec["init_pos"][:, alive0[1]] = mspec["init_pos"][:, alive0[0]]
    for f in mspec["inflows"]:
        f["period"], f["begin"] = 4.0, 1.0
    try:
        run_pair(mspec, "f32", 120, uniform_actions(mspec, 2, 0.0, 1.0), check_every=5)
    except ValueError as e:                                 # the placement check may refuse overlapping same-route starts
        assert "init" in str(e)


def test_merge_mixed_precision_holds_1e_4_against_float64_where_float32_does_not():
    """FS_MIXED on the open networks (k_steps_open<double, ., ., CSET = 2>): the float64 kernel -- positions, geometry, the
    inflow clocks, every decision in float64 -- with the float32 car-following models (idm_fd / sumo_acc_fd on the rounded
    speeds and gaps).  No bit-twin: held against the float64 kernel, (a) without noise over a whole 600-step episode of 5
    sub-steps: every discrete event identical (routes / arrivals / departures), positions and speeds within 1e-4 (measured: 3e-6 m), the
    float32 run an order of magnitude further off; (b) with noise 0.2 and the same Philox streams."""
    from flow_amd import _lib as L
    K = 600
    for noise in (0.0, 0.2):
        spec = merge_spec(R=24, cap_human=56, cap_rl=8, num_rl=8, pre=500.0, horizon=K, seed=33, env=O.ENV_MERGE_MA,
                          sims_per_step=5)
        spec["vehicles"] = [dict(v, noise=noise if v["controller"] != S.CTRL_RL else 0.0) for v in spec["vehicles"]]
        sims = {p: make(spec, p) for p in ("mixed", "f64", "f32")}
        for sim in sims.values():
            sim.reset()
        rng = np.random.default_rng(3)
        for k in range(K):
            a = rng.uniform(-1.0, 1.0, (24, 8)).astype(np.float32)
            outs = {p: sim.step(a) for p, sim in sims.items()}
            if k % 100 == 99 or k == K - 1:
                np.testing.assert_array_equal(sims["mixed"].get_state(L.FS_FIELD_ROUTE), sims["f64"].get_state(L.FS_FIELD_ROUTE))
                np.testing.assert_array_equal(sims["mixed"].get_state(L.FS_FIELD_COUNTERS), sims["f64"].get_state(L.FS_FIELD_COUNTERS))
                np.testing.assert_allclose(outs["mixed"][0], outs["f64"][0], rtol=0, atol=2e-6)     # normalised observations
                np.testing.assert_allclose(outs["mixed"][1], outs["f64"][1], rtol=0, atol=1e-5)
        assert sims["mixed"].last_kernel == "k_steps_open<mixed>", sims["mixed"].last_kernel
        alive = sims["f64"].get_state(L.FS_FIELD_ROUTE) >= 0
        dx_m = np.abs(sims["mixed"].pos - sims["f64"].pos)[alive].max()
        dv_m = np.abs(sims["mixed"].vel - sims["f64"].vel)[alive].max()
        assert dx_m < 1e-4 and dv_m < 1e-4, (noise, dx_m, dv_m)
        if noise == 0.0:
            same = (sims["f32"].get_state(L.FS_FIELD_ROUTE) == sims["f64"].get_state(L.FS_FIELD_ROUTE)).all(axis=1)
            dx_f = np.abs(sims["f32"].pos.astype(np.float64) - sims["f64"].pos)[alive & same[:, None]].max()
            assert dx_f > 10 * dx_m, (dx_f, dx_m)               # (float32: 1e-4 .. 7e-4 m on this network)
        assert sims["f64"].get_state(L.FS_FIELD_COUNTERS)[:, 5].min() > 100          # a whole episode of traffic
        for sim in sims.values():
            sim.close()


def test_lane_drop_mixed_precision_matches_float64():
    """The same form on the lane-drop network (k_steps_open<double, 64, 4, 2>, up to 64 vehicle slots): discrete events as in
    float64, positions within 1e-4 over 300 steps with random desired-velocity actions."""
    from flow_amd import _lib as L
    from helpers import bottleneck_spec
    spec = bottleneck_spec(R=6, cap_human=52, cap_rl=10, horizon=300, seed=9, q=2000.0)
    spec["vehicles"] = [dict(v, noise=0.0) for v in spec["vehicles"]]
    a, b = make(spec, "mixed"), make(spec, "f64")
    a.reset(), b.reset()
    rng = np.random.default_rng(4)
    for k in range(300):
        act = rng.uniform(-1.0, 1.0, (6, spec["num_rl"])).astype(np.float32)
        oa, ra, da = a.step(act)
        ob, rb, db = b.step(act)
        np.testing.assert_array_equal(da, db)
    np.testing.assert_array_equal(a.get_state(L.FS_FIELD_ROUTE), b.get_state(L.FS_FIELD_ROUTE))
    np.testing.assert_array_equal(a.get_state(L.FS_FIELD_COUNTERS), b.get_state(L.FS_FIELD_COUNTERS))
    alive = b.get_state(L.FS_FIELD_ROUTE) >= 0
    assert np.abs(a.pos - b.pos)[alive].max() < 1e-4 and np.abs(a.vel - b.vel)[alive].max() < 1e-4
    np.testing.assert_allclose(oa, ob, rtol=0, atol=2e-5)
    assert b.get_state(L.FS_FIELD_COUNTERS)[:, 6].min() > 20
    a.close(), b.close()


def test_float64_open_kernel_ranks_on_float32_images_and_counts_exactly_when_they_collide():
    """k_steps_open<double>: the ranking runs on the float32 IMAGES of the float64 positions (try-and-prove, count on 32-bit
    words) and hands over to the exact float64 count when images tie -- vehicles at one coordinate (released together), and
    vehicles a nanometre apart whose slot order contradicts their position order."""
    from helpers import bottleneck_spec
    spec = bottleneck_spec(R=3, cap_human=40, cap_rl=8, horizon=160, seed=2)
    X = np.asarray(spec["init_pos"], dtype=np.float64).copy()
    A, R_ = np.asarray(spec["init_alive"]).copy(), np.asarray(spec["init_route"]).copy()
    X[:, 40] = X[:, 0]                                      # the RL vehicle starts level with the human, other lane
    A[:, 1], X[:, 1], R_[:, 1] = True, 150.0, 0             # slots 1 / 2: same float32 image, the HIGHER slot is ahead
    A[:, 2], X[:, 2], R_[:, 2] = True, 150.0 + 1e-9, 3
    assert np.float32(X[0, 1]) == np.float32(X[0, 2]) and X[0, 1] < X[0, 2]
    spec["init_pos"], spec["init_alive"], spec["init_route"] = X, A, R_
    for f in spec["inflows"]:                               # same clock for both inflows: they release together
        f["period"], f["begin"] = 1.5, 1.0
    ora = run_pair(spec, "f64", 160, bottleneck_actions(spec, 3), check_every=5, exact=False, atol=1e-9)
    assert ora.total_departed.min() > 40


@pytest.mark.parametrize("seed", seeds([0, 2, 5, 11], [1, 4, 7, 14]))
def test_fuzz_random_open_network_configs_float64(seed):
    """The float64 kernels (ranking on float32 images with the exact count on ties; branch-free controller selection for
    IDM / RL / Sim populations) on the random configurations of the float32 fuzz, without noise, against the float64 oracle."""
    spec, acts = random_open_spec(seed)
    spec = quiet(spec)
    steps = int(spec["horizon"])
    run_pair(spec, "f64", steps, acts if seed % 5 else None, check_every=max(1, steps // 4), exact=False, atol=1e-9)
