"""GPU parity of the queue-order lane-drop kernel `k_drop_queue` (flow_amd/csrc/flowsim_dropq.h: one wave per entry lane)
against the float32 oracle (oracle/opennet.py with spec['cell_sum'] = 'fixed': the kernel adds the speeds of a lane-segment
as exact integers of 2^-16 m/s) -- bit for bit -- and against the slot-order kernels `k_steps_wide` / `k_steps_open` on the
same handle configuration (every state field bit for bit, the mean-speed observations to 1e-6: float sums in slot order
there), through the C ABI."""
import numpy as np
import pytest

from helpers import bottleneck_spec
from oracle import opennet as O
from test_open_gpu import compare_state, make

pytestmark = pytest.mark.gpu


def actions(spec, seed, lo=-1.0, hi=1.0):
    rng = np.random.default_rng(seed)
    R, A = spec["num_replicas"], spec["num_rl"]
    return lambda k: rng.uniform(lo, hi, (R, A)).astype(np.float32)


def run(spec, steps, action_fn=None, check_every=10, expect="k_drop_queue"):
    ora = O.MergeOracle(dict(spec, cell_sum="fixed"), np.float32)
    sim = make(spec, "f32")
    np.testing.assert_array_equal(sim.reset(), ora.reset().astype(np.float32))
    compare_state(sim, ora)
    for k in range(steps):
        a = None if action_fn is None else action_fn(k)
        o_ref, r_ref, d_ref = ora.step(a)
        o_gpu, r_gpu, d_gpu = sim.step(a)
        if k == 0:
            assert sim.last_kernel == expect
        np.testing.assert_array_equal(o_gpu, o_ref.astype(np.float32), err_msg="obs, step %d" % k)
        np.testing.assert_array_equal(r_gpu, r_ref.astype(np.float32), err_msg="reward, step %d" % k)
        np.testing.assert_array_equal(d_gpu, d_ref, err_msg="done, step %d" % k)
        if k % check_every == 0 or k == steps - 1:
            compare_state(sim, ora)
    sim.close()
    return ora


def test_drop_queue_desired_velocity_bit_exact_on_64_slots():
    spec = bottleneck_spec(R=5, cap_human=40, cap_rl=8, horizon=400, seed=3)
    ora = run(spec, 400, actions(spec, 7))
    assert ora.total_departed.min() > 100 and ora.total_arrived.min() > 40


def test_drop_queue_beyond_64_slots_with_a_standing_queue():
    """The workgroup-per-replica sizes (launch_wide): the demand of singleagent_bottleneck.py fills the four lanes."""
    spec = bottleneck_spec(R=3, cap_human=130, cap_rl=30, horizon=500, seed=5, q=2600.0)
    ora = run(spec, 500, actions(spec, 2), check_every=25)
    assert (ora.route >= 0).sum(axis=1).max() > 64


def test_drop_queue_base_env_sub_steps_and_no_actions():
    spec = bottleneck_spec(R=3, cap_human=50, cap_rl=6, horizon=150, seed=8, env=O.ENV_BOTTLENECK, sims_per_step=3)
    run(spec, 150, None)
    spec = bottleneck_spec(R=2, cap_human=50, cap_rl=6, horizon=150, seed=9, sims_per_step=2)
    run(spec, 150, None)


def test_drop_queue_fixed_entry_lanes_and_two_inflows_on_one_lane():
    spec = bottleneck_spec(R=3, cap_human=56, cap_rl=8, horizon=300, seed=11)
    fl = spec["inflows"]
    spec["inflows"] = [dict(fl[0], route=1, period=3.0), dict(fl[1], route=1, period=7.0), dict(fl[0], route=2, period=2.5),
                       dict(fl[0], route=-1, period=4.0)]
    run(spec, 300, actions(spec, 4))


def test_drop_queue_without_zipper_lookahead_collides_at_the_joins_and_goes_on():
    """zipper_distance = 0: side-by-side arrivals at a join collide (done flag); the launch goes on with overlapping
    vehicles, so a path is no longer in driving order and is re-sorted."""
    spec = bottleneck_spec(R=4, cap_human=56, cap_rl=8, horizon=300, seed=13, zipper_distance=0.0)
    ora = run(spec, 300, actions(spec, 5), check_every=5)
    assert ora.total_arrived.min() > 10


def test_drop_queue_rollout_equals_stepping_and_the_slot_order_kernel(monkeypatch):
    import torch
    from flow_amd import _lib as L
    spec = bottleneck_spec(R=4, cap_human=100, cap_rl=20, horizon=300, seed=21, sims_per_step=1)
    K, R, A = 200, 4, spec["num_rl"]
    acts = np.random.default_rng(1).uniform(-1, 1, (K, R, A)).astype(np.float32)
    dev = torch.device("cuda:0")

    def rollout(sim):
        out = (torch.empty((K, R, sim.obs_dim), dtype=torch.float32, device=dev),
               torch.empty((K, R), dtype=torch.float32, device=dev), torch.empty((K, R), dtype=torch.uint8, device=dev))
        sim.reset()
        sim.rollout_dev(K, *out, actions=torch.from_numpy(acts).to(dev))
        sim.sync()
        return [t.cpu().numpy() for t in out]

    a = make(spec, "f32")
    ra = rollout(a)
    assert a.last_kernel == "k_drop_queue"
    b = make(spec, "f32")
    b.reset()
    for k in range(K):
        o, r, d = b.step(acts[k])
        np.testing.assert_array_equal(ra[0][k], o, err_msg="obs %d" % k)
        np.testing.assert_array_equal(ra[1][k], r)
        np.testing.assert_array_equal(ra[2][k].astype(bool), d)
    monkeypatch.setenv("FLOWSIM_NO_QUEUE", "1")
    c = make(spec, "f32")
    rc = rollout(c)
    assert c.last_kernel == "k_steps_wide"
    C = len(spec["obs_cells"])
    np.testing.assert_array_equal(ra[0][:, :, :2 * C], rc[0][:, :, :2 * C])              # vehicle counts
    np.testing.assert_allclose(ra[0][:, :, 2 * C:4 * C], rc[0][:, :, 2 * C:4 * C], rtol=0, atol=1e-6)   # mean speeds
    np.testing.assert_array_equal(ra[0][:, :, 4 * C:], rc[0][:, :, 4 * C:])
    np.testing.assert_array_equal(ra[1], rc[1])
    np.testing.assert_array_equal(ra[2], rc[2])
    alive = a.get_state(L.FS_FIELD_ROUTE) >= 0
    for f in (L.FS_FIELD_POS, L.FS_FIELD_VEL, L.FS_FIELD_PREV_VEL, L.FS_FIELD_ACCEL, L.FS_FIELD_ROUTE, L.FS_FIELD_SEQ,
              L.FS_FIELD_ORIGIN, L.FS_FIELD_LEADER, L.FS_FIELD_HEADWAY, L.FS_FIELD_ARRIVED_RL, L.FS_FIELD_COUNTERS,
              L.FS_FIELD_MAX_SPEED):
        fa, fb, fc = a.get_state(f), b.get_state(f), c.get_state(f)
        np.testing.assert_array_equal(fa, fb, err_msg="rollout vs stepping, field %d" % f)
        if f in (L.FS_FIELD_COUNTERS, L.FS_FIELD_ROUTE, L.FS_FIELD_ARRIVED_RL):
            np.testing.assert_array_equal(fa, fc, err_msg="queue vs slot order, field %d" % f)
        else:
            np.testing.assert_array_equal(fa[alive], fc[alive], err_msg="queue vs slot order, field %d" % f)
    a.close(), b.close(), c.close()


def test_drop_queue_path_overflow_is_reported_by_the_step_itself():
    """A path that outgrows its wave's 64 lanes (243 slots, saturated network: the soak's seed 520) invalidates the handle:
    the step that overflowed raises (fs_step checks the kernel's flag after its own synchronisation), every step before it
    equals the oracle, and the same configuration runs through on the slot-order kernel (FLOWSIM_NO_QUEUE=1)."""
    rng = np.random.default_rng(7000 + 520)
    R = int(rng.integers(1, 5)); cap_rl = int(rng.integers(2, 30)); N = int(rng.integers(65, 257))
    spec = bottleneck_spec(R=R, cap_human=N - cap_rl, cap_rl=cap_rl, horizon=int(rng.integers(150, 420)), seed=520,
                           q=float(rng.choice([2300, 3600, 5000])), av_frac=float(rng.choice([0.1, 0.3])),
                           zipper_distance=float(rng.choice([0.0, 20.0, 50.0, 120.0])),
                           warmup_steps=int(rng.choice([0, 0, 20])), lane_change_cooldown_steps=int(rng.choice([2, 8, 20])),
                           lane_change_min_gain=float(rng.choice([3.0, 10.0])), crash_gap=float(rng.choice([0.0, 1.0])),
                           track_followers=bool(rng.integers(0, 2)), sims_per_step=int(rng.choice([1, 1, 2])))
    assert (R, N) == (3, 243)
    ora = O.MergeOracle(dict(spec, cell_sum="fixed"), np.float32)
    sim = make(spec, "f32")
    np.testing.assert_array_equal(sim.reset(), ora.reset().astype(np.float32))
    act = actions(spec, 520, -1.5, 1.5)
    raised_at = -1
    for k in range(int(spec["horizon"])):
        a = act(k)
        o_ref, r_ref, d_ref = ora.step(a)
        try:
            o_gpu, r_gpu, d_gpu = sim.step(a)
        except NotImplementedError as e:
            assert "k_drop_queue" in str(e) and "FLOWSIM_NO_QUEUE" in str(e)
            raised_at = k
            break
        assert sim.last_kernel == "k_drop_queue"
        np.testing.assert_array_equal(o_gpu, o_ref.astype(np.float32), err_msg="obs, step %d" % k)
    assert raised_at > 0
    assert max(int((ora.route[r][ora.alive[r]] == p).sum()) for r in range(R) for p in range(4)) > 64
    with pytest.raises(NotImplementedError, match="k_drop_queue"):
        sim.pos                                          # (sticky: the handle stays unusable)
    sim.close()
