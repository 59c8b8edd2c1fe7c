"""BASELINE configs[3] (C4) and configs[4] (C5) exactly as bench.py runs them -- same flow_params, same replica count,
whole episode in one launch (on the queue-order kernels k_drop_queue / k_merge_queue) -- with SAMPLED replicas checked bit for bit against the numpy oracle (which is far too slow for
all of them): observation, reward and done of every step, and the state at the end.  The oracle runs the sampled replicas
alone; the Philox streams of the inflows (random entry lanes) are keyed by the global replica index, which the oracle takes
from `replica_ids`."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import opennet as O                                     # noqa: E402
from test_open_gpu import compare_state                             # noqa: E402

pytestmark = pytest.mark.gpu


class Rows:
    """The sampled rows of a FlowSim handle, presented like a handle of their own to compare_state."""

    def __init__(self, sim, rows):
        self.sim, self.rows = sim, rows

    def get_state(self, field):
        return self.sim.get_state(field)[self.rows]

    pos = property(lambda s: s.sim.pos[s.rows])
    vel = property(lambda s: s.sim.vel[s.rows])
    headway = property(lambda s: s.sim.headway[s.rows])


def sampled_parity(fp, R, K, rows, act_seed, **oracle_kw):
    import torch
    from flow_amd.envs import VecFlowEnv
    dev = torch.device("cuda:0")
    vec = VecFlowEnv(fp, num_replicas=R, device=0)
    spec = vec.env._spec
    A = vec.act_dim
    gen = torch.Generator(device=dev).manual_seed(act_seed)
    tape = ((torch.rand((K, R, max(A, 1)), device=dev, generator=gen) * 2 - 1) * 1.5)[:, :, :A].contiguous()
    out = (torch.empty((K, R, vec.obs_dim), dtype=torch.float32, device=dev),
           torch.empty((K, R), dtype=torch.float32, device=dev), torch.empty((K, R), dtype=torch.uint8, device=dev))
    obs0 = vec.reset()
    vec.sim.rollout_dev(K, *out, actions=tape if A else None)
    torch.cuda.synchronize()
    kernel = vec.sim.last_kernel
    sub = dict(spec, num_replicas=len(rows), replica_ids=np.asarray(rows) + int(spec.get("replica_offset", 0)), **oracle_kw)
    for key in ("init_alive", "init_pos", "init_vel", "init_route"):
        sub[key] = np.asarray(spec[key])[rows]
    ora = O.MergeOracle(sub, np.float32)
    np.testing.assert_array_equal(obs0[rows].cpu().numpy(), ora.reset().astype(np.float32))
    obs, rew, done = (t[:, rows].cpu().numpy() for t in out)
    acts = tape[:, rows].cpu().numpy()
    for k in range(K):
        o, r, d = ora.step(acts[k] if A else None)
        np.testing.assert_array_equal(obs[k], o.astype(np.float32), err_msg="obs, step %d" % k)
        np.testing.assert_array_equal(rew[k], r.astype(np.float32), err_msg="reward, step %d" % k)
        np.testing.assert_array_equal(done[k], d, err_msg="done, step %d" % k)
    compare_state(Rows(vec.sim, rows), ora)
    vec.close()
    return kernel, ora


def test_c4_full_size_sampled_replicas_equal_the_oracle():
    import bench
    # (k_drop_queue adds the speeds of a lane-segment as exact integers of 2^-16 m/s: the oracle's cell_sum = 'fixed')
    kernel, ora = sampled_parity(bench.c4_flow_params(256), R=128, K=1000, rows=[0, 37, 90, 127], act_seed=3, cell_sum="fixed")
    assert kernel == "k_drop_queue"
    assert ora.total_departed.min() > 250 and ora.total_arrived.min() > 150      # the whole episode was traffic


def test_c5_full_size_sampled_replicas_equal_the_oracle():
    """The bench's configuration WITH its acceleration noise (IDMController(noise=0.2), as the reference ships the
    experiment): SumoParams(noise_math='exact') makes the Box-Muller draws fixed float32 sequences, the same in the numpy
    oracle (with the hardware's log2 / cos the comparison is a tolerance test: test_merge_po_noise_short_horizon_tolerance)."""
    import bench
    fp = bench.c5_flow_params("f32", noise=0.2)
    fp["sim"].noise_math = "exact"
    kernel, ora = sampled_parity(fp, R=1024, K=350, rows=[301, 1023], act_seed=4)     # (the numpy Philox is the slow part)
    assert kernel == "k_merge_queue"
    assert ora.total_departed.min() > 150 and ora.total_arrived.min() > 60


def test_c4_lane_change_leg_full_size_sampled_replicas_equal_the_oracle():
    """bench.py's c4_bottleneck_lane_change leg (flow/benchmarks/bottleneck1: lane_change_mode 1621 -> the simplified lane
    changing M11, on k_steps_wide: the ranked slot-order path) at the bench's replica count, two sampled replicas."""
    import bench
    kernel, ora = sampled_parity(bench.c4_flow_params(256, lane_change_mode=1621), R=128, K=300, rows=[5, 120], act_seed=6)
    assert kernel == "k_steps_wide"
    assert ora.num_lane_changes.min() > 10 and ora.total_arrived.min() > 20


def test_rl_ring_full_size_rollout_and_fused_policy_fragment():
    """The reference's RL ring at the bench's size (4096 replicas x 22 vehicles, a ring length per replica):
    (a) with noise 0.2, the 1500-step k_ring_pair rollout equals the generic kernel bit for bit in EVERY replica;
    (b) without noise, 6 sampled replicas of the same rollout equal the numpy oracle bit for bit, every step;
    (c) a 500-step fused policy fragment (k_ring_policy, noise 0.2, in-fragment resets) equals eager stepping
        (fs_policy_act_dev + fs_step_dev + masked fs_reset_dev) bit for bit in every replica."""
    import torch
    from oracle import refsim as S
    from test_policy_gpu import buffers, eager_obs0, make_policy
    from test_ringrl_gpu import make, rl_ring_spec, rollout, tape
    R, K = 4096, 1500
    dev = torch.device("cuda", 0)
    acts = tape(K, R, 1, seed=8)
    # (a)
    spec = rl_ring_spec(R=R, N=22, noise=0.2, warmup=0, horizon=K, seed=4)
    fast, slow = make(spec, "f32"), make(spec, "f32", FLOWSIM_NO_RING_RL=1)
    fast.reset(), slow.reset()
    a, b = rollout(fast, K, acts), rollout(slow, K, acts)
    assert fast.last_kernel.startswith("k_ring_pair") and slow.last_kernel.startswith("k_steps")
    for u, w, what in zip(a, b, ("obs", "reward", "done")):
        assert np.array_equal(u, w), what
    np.testing.assert_array_equal(fast.pos, slow.pos)
    np.testing.assert_array_equal(fast.vel, slow.vel)
    fast.close(), slow.close()
    # (b)
    rows = [0, 1, 777, 2048, 3333, 4095]
    quiet = rl_ring_spec(R=R, N=22, noise=0.0, warmup=0, horizon=K, seed=4)
    sim = make(quiet, "f32")
    sim.reset()
    o, r, d = rollout(sim, K, acts)
    sub = dict(quiet, num_replicas=len(rows), ring_length=np.asarray(quiet["ring_length"])[rows],
               init_pos=np.asarray(quiet["init_pos"])[rows])
    ora = S.RingOracle(sub, np.float32)
    ora.reset()
    for k in range(K):
        o_ref, r_ref, d_ref = ora.step(acts[k][rows])
        np.testing.assert_array_equal(o[k][rows], o_ref.astype(np.float32), err_msg="obs, step %d" % k)
        np.testing.assert_array_equal(r[k][rows], r_ref.astype(np.float32), err_msg="reward, step %d" % k)
    np.testing.assert_array_equal(sim.pos[rows], ora.x)
    sim.close()
    # (c)
    from flow_amd import _lib as L
    Kf = 500
    spec = rl_ring_spec(R=R, N=22, noise=0.2, warmup=5, horizon=300, seed=4)
    pol_a, pol_b = make_policy(3, False, seed=5), make_policy(3, False, seed=5)
    fused, eager = make(spec, "f32"), make(spec, "f32")
    fused.reset(), eager.reset()
    fo, fa, flp, fr, fd = buffers(Kf, R, dev)
    fused.policy_rollout_dev(pol_a.struct, Kf, fo, fa, flp, fr, fd, reset_done=True)
    fused.sync()
    assert fused.last_kernel == "k_ring_policy"
    eo, ea, elp, er, ed = buffers(Kf, R, dev)
    eo[0].copy_(torch.as_tensor(eager_obs0(eager), device=dev))
    torch.cuda.synchronize()
    for k in range(Kf):
        eager.policy_act_dev(pol_b.struct, eo[k], ea[k], elp[k])
        eager.step_dev(eo[k + 1], er[k], ed[k], ea[k].reshape(R, 1))
        eager.reset_dev(eo[k + 1], ed[k])
    eager.sync()
    for name, x, y in (("obs", fo, eo), ("act", fa, ea), ("logp", flp, elp), ("rew", fr, er), ("done", fd, ed)):
        assert torch.equal(x, y), name
    np.testing.assert_array_equal(fused.pos, eager.pos)
    assert int((fd != 0).sum()) >= R                       # every replica finished an episode inside the fragment
    fused.close(), eager.close()
