"""GPU parity of `k_steps_wide` (flowsim_wide.h): BottleneckNetwork replicas with MORE than 64 vehicle slots, one
workgroup of 2 / 4 waves per replica, against oracle/opennet.py through the C ABI.

Same bars as test_open_gpu.py: float32 kernel vs the float32 oracle twin bit-exact on every state field,
observation, reward and done flag; float64 kernel vs the float64 oracle within 1e-9.
"""
import numpy as np
import pytest

from conftest import seeds

from oracle import opennet as O
from test_open_gpu import bottleneck_actions, compare_state, compare_vmax, make, run_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def slot_order_kernels(monkeypatch):
    """This module holds the SLOT-order open-network kernels (k_steps_open / k_steps_wide) to the oracle; the queue-order
    kernels that take the same configurations by default have tests of their own (test_queue_gpu.py, test_dropq_gpu.py)."""
    monkeypatch.setenv("FLOWSIM_NO_QUEUE", "1")


@pytest.mark.slow                  # (the 100 / 128 / 200-slot cases below and in test_bottleneck_env_gpu.py stay in the fast set)
def test_wide_desired_velocity_f32_bit_exact_192_slots():
    """C4's demand: the queue upstream of the lane drops outgrows one wave (more than 64 vehicles in the network)."""
    from helpers import bottleneck_spec
    spec = bottleneck_spec(R=2, cap_human=170, cap_rl=22, horizon=700, seed=3)
    ora = run_pair(spec, "f32", 450, bottleneck_actions(spec, 5), check_every=50)
    assert (ora.alive.sum(axis=1) > 64).all()                          # really beyond the 64-slot kernel
    assert ora.total_arrived.min() > 60


def test_wide_two_waves_and_a_full_block():
    from helpers import bottleneck_spec
    for cap_h, cap_rl in ((90, 10), (116, 12)):                        # 100 slots (two waves, 28 idle lanes) / 128
        spec = bottleneck_spec(R=2, cap_human=cap_h, cap_rl=cap_rl, horizon=400, seed=cap_h)
        steps = 400 if cap_h == 90 else 260              # (the first case fills past one wave)
        ora = run_pair(spec, "f32", steps, bottleneck_actions(spec, 1), check_every=40)
        assert (ora.alive.sum(axis=1) > (64 if cap_h == 90 else 45)).any()


def test_wide_state_fields_warmup_and_max_speed():
    from helpers import bottleneck_spec
    spec = bottleneck_spec(R=2, cap_human=120, cap_rl=30, horizon=200, seed=7, warmup_steps=60)
    ora = O.MergeOracle(spec, np.float32)
    sim = make(spec, "f32")
    np.testing.assert_array_equal(sim.reset(), ora.reset().astype(np.float32))
    compare_state(sim, ora)
    acts = bottleneck_actions(spec, 2)
    for k in range(200):
        a = acts(k)
        o_ref, r_ref, d_ref = ora.step(a)
        o_gpu, r_gpu, d_gpu = sim.step(a)
        np.testing.assert_array_equal(o_gpu, o_ref.astype(np.float32))
        np.testing.assert_array_equal(r_gpu, r_ref.astype(np.float32))
        np.testing.assert_array_equal(d_gpu, d_ref)
        if k % 25 == 0:
            compare_state(sim, ora)
            compare_vmax(sim, ora)
    assert d_ref.all()
    sim.close()


def test_wide_base_env_no_actions_256_slots():
    from helpers import bottleneck_spec
    spec = bottleneck_spec(R=2, cap_human=236, cap_rl=20, horizon=500, seed=9, env=O.ENV_BOTTLENECK, num_rl=0,
                           action_cells=[], q=3600.0)
    ora = run_pair(spec, "f32", 500, None, check_every=50)
    assert ora.total_arrived.min() > 50


def test_wide_f64_matches_reference_arithmetic():
    from helpers import bottleneck_spec
    spec = bottleneck_spec(R=2, cap_human=130, cap_rl=14, horizon=400, seed=4)
    run_pair(spec, "f64", 400, bottleneck_actions(spec, 8), check_every=50, exact=False, atol=1e-9)


def test_wide_simplified_lane_changing_and_followers():
    """M11 and the sticky-follower bookkeeping (O1) across waves."""
    from helpers import bottleneck_spec
    spec = bottleneck_spec(R=3, cap_human=150, cap_rl=20, horizon=500, seed=13, lane_change_cooldown_steps=8,
                           lane_change_min_gain=8.0, track_followers=True)
    for v in spec["vehicles"][:150]:
        v["lane_change_mode"] = 1621
    ora = run_pair(spec, "f32", 300, bottleneck_actions(spec, 4), check_every=50)
    assert (ora.num_lane_changes > 40).all()


def test_wide_without_zipper_lookahead_crashes_at_the_joins():
    from helpers import bottleneck_spec
    spec = bottleneck_spec(R=3, cap_human=100, cap_rl=8, horizon=400, seed=11, zipper_distance=0.0)
    ora = O.MergeOracle(spec, np.float32)
    sim = make(spec, "f32")
    np.testing.assert_array_equal(sim.reset(), ora.reset().astype(np.float32))
    crashed = np.zeros(3, dtype=bool)
    for k in range(400):
        o_ref, r_ref, d_ref = ora.step(None)
        o_gpu, r_gpu, d_gpu = sim.step(None)
        np.testing.assert_array_equal(o_gpu, o_ref.astype(np.float32))
        np.testing.assert_array_equal(d_gpu, d_ref)
        crashed |= d_ref & (ora.time_counter < 400)
    assert crashed.any()
    sim.close()


def test_wide_masked_reset_and_rollout_equal_stepping():
    import torch
    from helpers import bottleneck_spec
    spec = bottleneck_spec(R=4, cap_human=100, cap_rl=12, horizon=200, seed=5)
    K, R, A = 80, 4, spec["num_rl"]
    acts = np.random.default_rng(3).uniform(-1.0, 1.0, (K, R, A)).astype(np.float32)
    a, b = make(spec, "f32"), make(spec, "f32")
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
    a.close()
    # masked reset against the oracle: replicas 0 and 3 restart, 1 and 2 keep going
    ora = O.MergeOracle(spec, np.float32)
    ora.reset()
    for k in range(K):
        ora.step(acts[k])
    mask = np.array([1, 0, 0, 1], dtype=bool)
    np.testing.assert_array_equal(b.reset(mask), ora.reset(mask).astype(np.float32))
    compare_state(b, ora)
    for k in range(40):
        o_ref, r_ref, d_ref = ora.step(acts[k])
        o_gpu, r_gpu, d_gpu = b.step(acts[k])
        np.testing.assert_array_equal(o_gpu, o_ref.astype(np.float32))
        np.testing.assert_array_equal(r_gpu, r_ref.astype(np.float32))
        np.testing.assert_array_equal(d_gpu, d_ref)
    compare_state(b, ora)
    b.close()


@pytest.mark.parametrize("seed", seeds([0], [1, 2, 3, 4, 5, 6, 7]))
def test_wide_fuzz_random_lane_drop_configs_bit_exact(seed):
    from helpers import bottleneck_spec
    rng = np.random.default_rng(7000 + seed)
    R = int(rng.integers(1, 5))
    cap_rl = int(rng.integers(2, 30))
    N = int(rng.integers(65, 257))
    spec = bottleneck_spec(R=R, cap_human=N - cap_rl, cap_rl=cap_rl, horizon=int(rng.integers(150, 420)), seed=seed,
                           q=float(rng.choice([2300, 3600, 5000])), av_frac=float(rng.choice([0.1, 0.3])),
                           zipper_distance=float(rng.choice([0.0, 20.0, 50.0, 120.0])),
                           warmup_steps=int(rng.choice([0, 0, 20])), lane_change_cooldown_steps=int(rng.choice([2, 8, 20])),
                           lane_change_min_gain=float(rng.choice([3.0, 10.0])), crash_gap=float(rng.choice([0.0, 1.0])),
                           track_followers=bool(rng.integers(0, 2)), sims_per_step=int(rng.choice([1, 1, 2])))
    if rng.integers(0, 2):
        for v in spec["vehicles"][:N - cap_rl]:
            v["lane_change_mode"] = 1621
    A = spec["num_rl"]
    acts = (lambda k, r=np.random.default_rng(seed): r.uniform(-1.5, 1.5, (R, A)).astype(np.float32))
    run_pair(spec, "f32", int(spec["horizon"]), acts, check_every=30)


def test_wide_other_controllers_and_generic_instantiation(monkeypatch):
    """k_steps_wide exists twice in float32: CSET = 1 for IDM / RL / Sim populations (every case above) and the generic
    one.  A mixed population (CFM with the instantaneous fail-safe, FollowerStopper, IDM) takes the generic kernel; the
    Sim / RL population forced onto the generic kernel (FLOWSIM_FORCE_GENERIC) gives the same bits as on CSET = 1."""
    from helpers import bottleneck_spec, idm_vehicle
    from oracle import refsim as S
    spec = bottleneck_spec(R=2, cap_human=110, cap_rl=12, horizon=300, seed=17)
    veh = spec["vehicles"]
    for i in range(110):
        if i % 3 == 0:
            veh[i] = idm_vehicle(controller=S.CTRL_CFM, p=[1, 1, 1, 1, 8, 0, 0, 0], speed_mode=1, type=0,
                                 fail_safe=S.FAILSAFE_INSTANTANEOUS)
        elif i % 3 == 1:
            veh[i] = idm_vehicle(controller=S.CTRL_FOLLOWER_STOPPER, p=[12.0] + [0] * 7, speed_mode=1, type=0)
        else:
            veh[i] = idm_vehicle(p=[20.0, 1, 1.2, 1.5, 4, 2, 0, 0], speed_mode=1, type=0, noise=0.0)
    run_pair(spec, "f32", 300, bottleneck_actions(spec, 2), check_every=30)
    plain = bottleneck_spec(R=2, cap_human=110, cap_rl=12, horizon=200, seed=18)
    outs = []
    for force in ("0", "1"):
        monkeypatch.setenv("FLOWSIM_FORCE_GENERIC", force)
        sim = make(plain, "f32")
        sim.reset()
        acts = bottleneck_actions(plain, 6)
        for k in range(200):
            o, r, d = sim.step(acts(k))
        outs.append((o, r, sim.pos.copy(), sim.vel.copy()))
        sim.close()
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)


def test_wide_scaling_two_eight_entry_lanes():
    """BottleneckNetwork scaling 2 (flow/benchmarks/bottleneck2.py): 8 -> 4 -> 2 lanes, 70 observed and 40 controlled
    lane-segments, twice the demand; num_paths = 8 in the workgroup-per-replica kernel."""
    from flow_amd import _lib as L
    from helpers import bottleneck_spec
    spec = bottleneck_spec(R=1, cap_human=200, cap_rl=40, horizon=400, seed=6, q=4000.0, scaling=2)
    assert spec["num_paths"] == 8 and len(spec["obs_cells"]) == 70 and spec["num_rl"] == 40
    ora = run_pair(spec, "f32", 200, bottleneck_actions(spec, 3), check_every=50)
    alive = ora.alive[0]
    assert set(ora.route[0][alive]) == set(range(8))                       # every entry lane is in use
    assert set((ora.route[0][alive & (ora.x[0] > spec["merge2_x"])] >> 2)) == {0, 1}    # two lanes leave the network
    assert ora.total_arrived.min() > 40
    # with the simplified lane changing, in float64 and on two waves
    spec = bottleneck_spec(R=2, cap_human=100, cap_rl=20, horizon=300, seed=8, q=4000.0, scaling=2,
                           lane_change_cooldown_steps=8, lane_change_min_gain=8.0)
    for v in spec["vehicles"][:100]:
        v["lane_change_mode"] = 1621
    ora = run_pair(spec, "f32", 160, bottleneck_actions(spec, 5), check_every=40)
    assert (ora.num_lane_changes > 10).all()
    run_pair(spec, "f64", 60, bottleneck_actions(spec, 5), check_every=30, exact=False, atol=1e-9)
    with pytest.raises(NotImplementedError, match="more than 64 vehicle slots"):
        make(bottleneck_spec(R=1, cap_human=50, cap_rl=10, scaling=2), "f32")


def test_full_size_properties_of_the_c4_and_c5_configurations():
    """BASELINE configs[3] / [4] at their per-GPU sizes (128 lane-drop replicas x 256 slots on k_steps_wide, 1024 merge
    replicas x 64 slots on k_steps_open), too large for the oracle: size-independent properties instead --
    a K-step launch equals K one-step launches bit for bit, vehicles are conserved (in network + arrived = initial +
    departed), nobody overlaps its leader on its own lane, two shards of 64 reproduce the handle of 128."""
    import torch
    from flow_amd import _lib as L
    from helpers import bottleneck_spec, merge_spec
    dev = torch.device("cuda:0")

    def rollout(sim, K, acts):
        obs = torch.empty((K, sim.R, sim.obs_dim), dtype=torch.float32, device=dev)
        rew = torch.empty((K, sim.R), dtype=torch.float32, device=dev)
        done = torch.empty((K, sim.R), dtype=torch.uint8, device=dev)
        sim.rollout_dev(K, obs, rew, done, actions=acts)
        sim.sync()
        return obs, rew, done

    def conserved(sim, n_init):
        cnt = sim.get_state(L.FS_FIELD_COUNTERS)
        route = sim.get_state(L.FS_FIELD_ROUTE)
        np.testing.assert_array_equal((route >= 0).sum(axis=1) + cnt[:, 5], n_init + cnt[:, 6])
        lead = sim.get_state(L.FS_FIELD_LEADER)
        h = sim.headway
        alive = route >= 0
        assert (lead[alive] < sim.N).all() and np.isfinite(h[alive]).all()
        return cnt

    K = 120
    for name, spec, n_init in (("c4", bottleneck_spec(R=128, cap_human=230, cap_rl=26, horizon=1000, seed=31), 2),
                               ("c5", merge_spec(R=1024, cap_human=56, cap_rl=8, num_rl=8, horizon=600, seed=32,
                                                 env=O.ENV_MERGE_MA), None)):
        if name == "c5":
            spec["vehicles"] = [dict(v, noise=0.0) for v in spec["vehicles"]]
            n_init = np.asarray(spec["init_alive"]).sum(axis=1)
        a, b = make(spec, "f32"), make(spec, "f32")
        a.reset(), b.reset()
        gen = torch.Generator(device=dev).manual_seed(5)
        acts = (torch.rand((K, a.R, max(a.act_dim, 1)), device=dev, generator=gen) * 2 - 1)[:, :, :a.act_dim].contiguous()
        obs, rew, done = rollout(a, K, acts if a.act_dim else None)
        for k in range(K):
            o = torch.empty((a.R, a.obs_dim), dtype=torch.float32, device=dev)
            r = torch.empty((a.R,), dtype=torch.float32, device=dev)
            d = torch.empty((a.R,), dtype=torch.uint8, device=dev)
            b.step_dev(o, r, d, actions=acts[k] if a.act_dim else None)
            if k % 40 == 39 or k == K - 1:
                b.sync()
                assert torch.equal(o, obs[k]) and torch.equal(r, rew[k]) and torch.equal(d, done[k])
        np.testing.assert_array_equal(a.pos, b.pos)
        cnt = conserved(a, n_init)
        assert (cnt[:, 6] > 10).all()                          # the inflows were at work
        if name == "c4":
            halves = []
            for lo, hi in ((0, 64), (64, 128)):
                sub = dict(spec, num_replicas=hi - lo, replica_offset=lo)
                for key in ("init_alive", "init_pos", "init_vel", "init_route"):
                    sub[key] = np.asarray(spec[key])[lo:hi]
                s = make(sub, "f32")
                s.reset()
                o2, r2, d2 = rollout(s, K, acts[:, lo:hi].contiguous())
                assert torch.equal(o2, obs[:, lo:hi]) and torch.equal(r2, rew[:, lo:hi])
                halves.append(s.get_state(L.FS_FIELD_ROUTE))
                s.close()
            np.testing.assert_array_equal(np.concatenate(halves), a.get_state(L.FS_FIELD_ROUTE))
        a.close(), b.close()


def test_wide_equal_positions_take_the_exact_ranking_path():
    """Two inflows on one clock release side by side at one coordinate: the 32-bit ranking of k_steps_wide sees the
    collision in the scatter and repeats the sub-step's ranking with the exact 64-bit keys."""
    from helpers import bottleneck_spec
    spec = bottleneck_spec(R=2, cap_human=110, cap_rl=18, horizon=220, seed=4, q=3600.0)
    X = np.asarray(spec["init_pos"]).copy()
    X[:, 110] = X[:, 0]
    spec["init_pos"] = X
    for f in spec["inflows"]:
        f["period"], f["begin"] = 1.0, 1.0
    probe = O.MergeOracle(spec, np.float32)
    probe.reset()
    acts = bottleneck_actions(spec, 3)
    tie_steps = 0
    for k in range(220):
        probe.step(acts(k))
        tie_steps += sum(len(np.unique(probe.x[r][probe.alive[r]])) < probe.alive[r].sum() for r in range(2))
    assert tie_steps > 30 and (probe.alive.sum(axis=1) > 64).any()
    run_pair(spec, "f32", 220, bottleneck_actions(spec, 3), check_every=10)


def test_wide_float64_ranks_on_float32_images_and_counts_exactly_when_they_collide():
    """The float64 kernel ranks by updating 64-bit keys that hold the FLOAT32 image of a position; two neighbours of the
    ranking with equal images -- here: vehicles released side by side (equal positions) and vehicles a nanometre apart
    (equal images, different float64 positions, the HIGHER slot ahead where the key's slot field says behind) -- fail the proof and the block counts on
    the float64 positions.  Leaders, headways and everything downstream must be the float64 oracle's."""
    from helpers import bottleneck_spec
    spec = bottleneck_spec(R=2, cap_human=110, cap_rl=18, horizon=200, seed=4, q=3600.0)
    X = np.asarray(spec["init_pos"], dtype=np.float64).copy()
    A, R_ = np.asarray(spec["init_alive"]).copy(), np.asarray(spec["init_route"]).copy()
    # slots 1 / 2: same float32 image, float64 says the HIGHER slot is ahead (the key's slot field says the opposite)
    A[:, 1], X[:, 1], R_[:, 1] = True, 150.0, 0
    A[:, 2], X[:, 2], R_[:, 2] = True, 150.0 + 1e-9, 3
    assert np.float32(X[0, 1]) == np.float32(X[0, 2]) and X[0, 1] < X[0, 2]
    A[:, 3], X[:, 3], R_[:, 3] = True, X[:, 0], 3       # and a vehicle at exactly the position of another one
    spec["init_alive"], spec["init_route"] = A, R_
    spec["init_pos"] = X
    for f in spec["inflows"]:
        f["period"], f["begin"] = 1.0, 1.0
    ora = run_pair(spec, "f64", 200, bottleneck_actions(spec, 3), check_every=10, exact=False, atol=1e-9)
    assert (ora.alive.sum(axis=1) > 64).any()


@pytest.mark.parametrize("seed", seeds([], [0, 3, 5]))     # (float64 on this kernel: test_wide_f64_matches_reference_arithmetic)
def test_wide_fuzz_random_lane_drop_configs_float64(seed):
    """k_steps_wide<double> (ranking on float32 images, exact count on ties) on random lane-drop configurations against the
    float64 oracle."""
    from helpers import bottleneck_spec
    rng = np.random.default_rng(7000 + seed)
    R = int(rng.integers(1, 4))
    cap_rl = int(rng.integers(2, 30))
    N = int(rng.integers(65, 257))
    spec = bottleneck_spec(R=R, cap_human=N - cap_rl, cap_rl=cap_rl, horizon=int(rng.integers(150, 300)), seed=seed,
                           q=float(rng.choice([2300, 3600, 5000])), av_frac=float(rng.choice([0.1, 0.3])),
                           zipper_distance=float(rng.choice([0.0, 20.0, 50.0, 120.0])),
                           lane_change_cooldown_steps=int(rng.choice([2, 8, 20])),
                           lane_change_min_gain=float(rng.choice([3.0, 10.0])), crash_gap=float(rng.choice([0.0, 1.0])),
                           track_followers=bool(rng.integers(0, 2)))
    if rng.integers(0, 2):
        for v in spec["vehicles"][:N - cap_rl]:
            v["lane_change_mode"] = 1621
    A = spec["num_rl"]
    acts = (lambda k, r=np.random.default_rng(seed): r.uniform(-1.5, 1.5, (R, A)).astype(np.float32))
    run_pair(spec, "f64", int(spec["horizon"]), acts, check_every=30, exact=False, atol=1e-9)
