"""GPU parity of the queue-order merge kernel `k_merge_queue` (flow_amd/csrc/flowsim_queue.h): against the float32
oracle (oracle/opennet.py, bit for bit, noise off) and against the slot-order kernel `k_steps_open` on the same handle
configuration (bit for bit, noise ON: both evaluate the same Philox draws with the same device functions), through the
C ABI.  The queue formulation itself is proven on the CPU in tests/test_queue_model.py."""
import os

import numpy as np
import pytest

from helpers import merge_spec
from oracle import opennet as O
from test_open_gpu import compare_state, make, quiet, run_pair

pytestmark = pytest.mark.gpu


def ma_spec(**kw):
    kw.setdefault("env", O.ENV_MERGE_MA)
    return merge_spec(**kw)


def nan_actions(R, A, seed, p_nan=0.2):
    rng = np.random.default_rng(seed)

    def acts(k):
        a = rng.uniform(-1.0, 1.5, (R, A)).astype(np.float32)
        a[rng.random((R, A)) < p_nan] = np.nan              # "the vehicle just entered": no action
        return a
    return acts


def test_queue_kernel_is_chosen_and_bit_exact_against_the_oracle():
    for apply in (False, True):
        spec = quiet(ma_spec(R=5, cap_human=26, cap_rl=5, num_rl=5, horizon=400, seed=3, ma_apply_actions=apply))
        sim = make(spec, "f32")
        sim.reset()
        sim.step(None)
        assert sim.last_kernel == "k_merge_queue"
        sim.close()
        ora = run_pair(spec, "f32", 400, nan_actions(5, 5, 13))
        assert ora.total_departed.min() > 20 and ora.total_arrived.min() > 5


@pytest.mark.parametrize("seed", [2, 5, 8, 11])
def test_queue_kernel_through_collisions_ties_and_full_slot_pools(seed):
    """speed_mode 'aggressive' (nobody obeys SUMO's safe speed) on a short, busy merge: vehicles run into each other --
    the multi-agent env does not end the episode (flow/envs/multiagent/base.py:188-190) -- so the queues are re-sorted,
    the slot pools run full and the ramp queue spills back."""
    spec = quiet(ma_spec(R=4, cap_human=18 + seed, cap_rl=4, num_rl=4, horizon=300, seed=seed, pre=150.0,
                         q_highway=1500 + 100 * seed, q_merge=200 + 80 * seed, sims_per_step=1 + seed % 3,
                         ma_apply_actions=bool(seed % 2)))
    spec["vehicles"] = [dict(v, speed_mode=0) for v in spec["vehicles"]]
    run_pair(spec, "f32", 300, nan_actions(4, 4, seed), check_every=7)


def test_queue_kernel_sim_step_half_second_and_five_sub_steps():
    spec = quiet(ma_spec(R=3, cap_human=30, cap_rl=6, num_rl=6, horizon=120, seed=4, sim_step=0.5, sims_per_step=5,
                         q_merge=400.0))
    run_pair(spec, "f32", 120, None, check_every=5)


def test_queue_kernel_small_pool():
    spec = quiet(ma_spec(R=7, cap_human=6, cap_rl=2, num_rl=2, horizon=300, seed=5, pre=120.0, q_highway=1500.0))
    run_pair(spec, "f32", 300, None)


def test_queue_rollout_equals_stepping_and_the_slot_order_kernel_with_noise(monkeypatch):
    """One K-step launch == K one-step launches == the slot-order kernel, noise included (GPU against GPU: every array,
    free slots too)."""
    import torch
    from flow_amd import _lib as L
    spec = ma_spec(R=6, cap_human=24, cap_rl=5, num_rl=5, horizon=300, seed=14, ma_apply_actions=True, sims_per_step=2)
    K, R, A = 150, 6, 5
    rng = np.random.default_rng(3)
    acts = rng.uniform(-1.0, 1.5, (K, R, A)).astype(np.float32)
    acts[rng.random((K, R, A)) < 0.2] = np.nan
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
    assert a.last_kernel == "k_merge_queue"
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
    assert c.last_kernel == "k_steps_open"
    for x, y in zip(ra, rc):
        np.testing.assert_array_equal(x, y)
    fields = (L.FS_FIELD_POS, L.FS_FIELD_VEL, L.FS_FIELD_PREV_VEL, L.FS_FIELD_ACCEL, L.FS_FIELD_ROUTE, L.FS_FIELD_SEQ,
              L.FS_FIELD_ORIGIN, L.FS_FIELD_FOLLOWER, L.FS_FIELD_LEADER, L.FS_FIELD_HEADWAY, L.FS_FIELD_ARRIVED_RL,
              L.FS_FIELD_COUNTERS, L.FS_FIELD_MAX_SPEED)
    alive = a.get_state(L.FS_FIELD_ROUTE) >= 0
    for f in fields:
        fa, fb, fc = a.get_state(f), b.get_state(f), c.get_state(f)
        np.testing.assert_array_equal(fa, fb, err_msg="rollout vs stepping, field %d" % f)
        if f in (L.FS_FIELD_COUNTERS, L.FS_FIELD_ROUTE, L.FS_FIELD_ARRIVED_RL):
            np.testing.assert_array_equal(fa, fc, err_msg="queue vs slot order, field %d" % f)
        else:
            np.testing.assert_array_equal(fa[alive], fc[alive], err_msg="queue vs slot order, field %d" % f)
    a.close(), b.close(), c.close()


def test_queue_kernel_half_precision_state():
    """FS_F16S (BASELINE configs[4]: fp16 state, fp32 integrator): the same launch on both kernels."""
    import torch
    spec = ma_spec(R=4, cap_human=24, cap_rl=5, num_rl=5, horizon=200, seed=9, sims_per_step=5)
    K, R = 100, 4
    dev = torch.device("cuda:0")
    res = []
    for env in ("0", "1"):
        os.environ["FLOWSIM_NO_QUEUE"] = env
        try:
            sim = make(spec, "f16s")
        finally:
            os.environ.pop("FLOWSIM_NO_QUEUE")
        out = (torch.empty((K, R, sim.obs_dim), dtype=torch.float32, device=dev),
               torch.empty((K, R), dtype=torch.float32, device=dev), torch.empty((K, R), dtype=torch.uint8, device=dev))
        sim.reset()
        sim.rollout_dev(K, *out)
        sim.sync()
        res.append(([t.cpu().numpy() for t in out], sim.pos.copy(), sim.vel.copy(), sim.last_kernel))
        sim.close()
    assert res[0][3] == "k_merge_queue" and res[1][3] == "k_steps_open"
    for x, y in zip(res[0][0], res[1][0]):
        np.testing.assert_array_equal(x, y)
