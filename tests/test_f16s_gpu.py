"""FS_F16S -- BASELINE configs[4] "fp16 state with fp32 integrator" on the merge network: the positions and speeds a
handle keeps in HBM between launches are IEEE halves (a position two of them), a launch steps in float32.

* within ONE launch the arithmetic is FS_F32's: starting from a state that the halves represent exactly, a K-step
  rollout writes bit for bit the observations and rewards of the float32 handle;
* across launches the state is rounded at every boundary: the stored state IS what halves can hold, and stepping step
  by step stays within the stated budget of the float32 run (speeds: 11 bits, positions: 22);
* host access (fs_get_state / fs_set_state) speaks float32 and round-trips through the halves."""
import numpy as np
import pytest

from helpers import merge_spec

pytestmark = pytest.mark.gpu


def make(spec, precision):
    from flow_amd.sim import FlowSim
    return FlowSim(spec, precision=precision)


def to_half_pair(x):
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32) + lo.astype(np.float32)


def rollout(sim, K, act):
    import torch
    dev = torch.device("cuda", 0)
    o = torch.zeros((K, sim.R, sim.obs_dim), device=dev)
    r = torch.zeros((K, sim.R), device=dev)
    d = torch.zeros((K, sim.R), dtype=torch.uint8, device=dev)
    a = torch.as_tensor(act, device=dev).contiguous()
    torch.cuda.synchronize()
    sim.rollout_dev(K, o, r, d, actions=a, action_stride_steps=0)
    sim.sync()
    return o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy()


def test_half_state_float32_integrator_on_the_merge():
    from flow_amd import _lib as L
    spec = merge_spec(R=6, cap_human=24, cap_rl=6, num_rl=3, horizon=400, seed=4)
    spec["vehicles"] = [dict(v, noise=0.0) for v in spec["vehicles"]]
    h16, f32 = make(spec, "f16s"), make(spec, "f32")
    h16.reset(), f32.reset()
    act = np.full((6, 3), 0.4, np.float32)
    # (1) one launch = float32 arithmetic: the float32 handle is given exactly the state the halves hold after the reset
    # (everything else -- routes, ids, counters, inflow clocks -- is equal: both were just reset), then ONE 150-step launch
    for field in (L.FS_FIELD_POS, L.FS_FIELD_VEL):
        f32.set_state(field, h16.get_state(field))
    a, b = rollout(h16, 150, act), rollout(f32, 150, act)
    for u, w in zip(a, b):
        np.testing.assert_array_equal(u, w)
    alive = h16.get_state(L.FS_FIELD_ROUTE) >= 0
    assert alive.sum() > 40                        # the inflows have filled the network
    np.testing.assert_array_equal(alive, f32.get_state(L.FS_FIELD_ROUTE) >= 0)
    # (2) what the handle keeps between launches is what halves hold (the float32 handle keeps more)
    x16, v16 = h16.pos.copy(), h16.vel.copy()
    np.testing.assert_array_equal(x16[alive], to_half_pair(f32.pos[alive]))
    np.testing.assert_array_equal(v16[alive], f32.vel[alive].astype(np.float16).astype(np.float32))
    assert not np.array_equal(f32.vel[alive], v16[alive])
    # (3) fs_set_state rounds through the halves, fs_get_state returns float32; the other field is left alone
    x = x16.copy()
    x[alive] += np.float32(0.123456)
    h16.set_state(L.FS_FIELD_POS, x)
    np.testing.assert_array_equal(h16.pos[alive], to_half_pair(x[alive]))
    np.testing.assert_array_equal(h16.vel[alive], v16[alive])
    h16.close(), f32.close()


def test_step_by_step_stays_within_the_rounding_budget_of_float32():
    """Every fs_step is a launch boundary: speeds are rounded to 11 bits (<= 2^-7 m/s below 32 m/s) each time."""
    from flow_amd import _lib as L
    spec = merge_spec(R=4, cap_human=20, cap_rl=4, num_rl=2, horizon=200, seed=9)
    spec["vehicles"] = [dict(v, noise=0.0) for v in spec["vehicles"]]
    h16, f32 = make(spec, "f16s"), make(spec, "f32")
    np.testing.assert_array_equal(h16.reset(), f32.reset())
    act = np.full((4, 2), 0.2, np.float32)
    for k in range(40):
        o16, r16, d16 = h16.step(act)
        o32, r32, d32 = f32.step(act)
    a16, a32 = h16.get_state(L.FS_FIELD_ROUTE) >= 0, f32.get_state(L.FS_FIELD_ROUTE) >= 0
    np.testing.assert_array_equal(a16, a32)                      # the same vehicles are in the network
    dv = np.abs(h16.vel[a16] - f32.vel[a16]).max()
    dx = np.abs(h16.pos[a16] - f32.pos[a16]).max()
    assert 0 < dv < 0.25 and dx < 1.0, (dv, dx)                  # 40 roundings of <= 0.008 m/s, amplified by the followers
    assert np.abs(r16 - r32).max() < 0.05
    h16.close(), f32.close()


def test_f16s_is_refused_outside_the_merge_network():
    from helpers import ring_spec
    with pytest.raises(NotImplementedError, match="FS_F16S"):
        make(ring_spec(R=2, N=22), "f16s")
