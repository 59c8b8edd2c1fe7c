"""Policy in the loop (flow_amd/csrc/flowsim_policy.h): K x (policy -> action -> Env.step -> reset of finished episodes)
as one launch -- what examples/train.py:110-212 does with one Python call and socket round trips per step.

* the fused fragment equals eager stepping (fs_policy_act_dev, fs_step_dev, masked fs_reset_dev) bit for bit, float32
  with acceleration noise and FS_MIXED, resets with warm-up steps and a new ring length included;
* the in-kernel network agrees with the same torch module (tolerance: different summation order), its samples follow
  N(mean, std) and its log-probabilities are those of the sampled actions;
* the simulator inside the fragment is the oracle's: replaying the fragment's own actions through oracle/refsim.py
  reproduces its observations and rewards exactly."""
import numpy as np
import pytest

from oracle import refsim as S
from test_ringrl_gpu import make, rl_ring_spec

pytestmark = pytest.mark.gpu


def make_policy(num_hidden=3, free_log_std=False, seed=0, dev="cuda:0"):
    import torch
    from flow_amd.utils.device_policy import DevicePolicy
    g = torch.Generator().manual_seed(seed)
    dims = [3] + [32] * num_hidden
    hidden = [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(num_hidden)]
    head = torch.nn.Linear(32, 1 if free_log_std else 2)
    for l in hidden + [head]:
        with torch.no_grad():
            l.weight.copy_(torch.randn(l.weight.shape, generator=g) * 0.4)
            l.bias.copy_(torch.randn(l.bias.shape, generator=g) * 0.2)
    with torch.no_grad():
        head.weight.mul_(0.3)
    for l in hidden + [head]:
        l.to(dev)
    ls = torch.nn.Parameter(torch.tensor([-0.7], device=dev)) if free_log_std else None
    return DevicePolicy(hidden, head, log_std=ls, seed=77 + seed)


def buffers(K, R, dev):
    import torch
    out = (torch.zeros((K + 1, R, 3), device=dev), torch.zeros((K, R), device=dev), torch.zeros((K, R), device=dev),
           torch.zeros((K, R), device=dev), torch.zeros((K, R), dtype=torch.uint8, device=dev))
    torch.cuda.synchronize()          # (the handles launch on streams of their own)
    return out


@pytest.mark.parametrize("precision,noise,num_hidden,free", [("f32", 0.2, 3, False), ("f32", 0.0, 2, True),
                                                               ("mixed", 0.0, 3, False), ("f32", 0.2, 1, True),
                                                               ("mixed", 0.2, 3, False)])
def test_fused_fragment_equals_eager_stepping(precision, noise, num_hidden, free):
    import torch
    from flow_amd import _lib as L
    K, R = 70, 9
    dev = torch.device("cuda", 0)
    # horizon 40 + warm-up 7: every replica finishes an episode inside the fragment and is reset in it
    spec = rl_ring_spec(R=R, N=22, noise=noise, warmup=7, horizon=40, seed=21)
    pol_a, pol_b = make_policy(num_hidden, free, seed=3), make_policy(num_hidden, free, seed=3)
    fused, eager = make(spec, precision), make(spec, precision)
    for sim in (fused, eager):                     # a pending ring length: the in-fragment resets must take it
        sim.reset()
        sim.set_state(L.FS_FIELD_INIT_RING_LENGTH, sim.get_state(L.FS_FIELD_RING_LENGTH) + 3.0)
    o, a, lp, r, d = buffers(K, R, dev)
    fused.policy_rollout_dev(pol_a.struct, K, o, a, lp, r, d, reset_done=True)
    fused.sync()
    assert fused.last_kernel == "k_ring_policy"
    eo, ea, elp, er, ed = buffers(K, R, dev)
    eo[0].copy_(torch.as_tensor(eager_obs0(eager), device=dev))       # observation of the current state
    torch.cuda.synchronize()
    for k in range(K):
        eager.policy_act_dev(pol_b.struct, eo[k], ea[k], elp[k])
        eager.step_dev(eo[k + 1], er[k], ed[k], ea[k].reshape(R, 1))
        eager.reset_dev(eo[k + 1], ed[k])
    eager.sync()
    for name, x, y in (("obs", o, eo), ("act", a, ea), ("logp", lp, elp), ("rew", r, er), ("done", d, ed)):
        np.testing.assert_array_equal(x.cpu().numpy(), y.cpu().numpy(), err_msg=name)
    np.testing.assert_array_equal(fused.pos, eager.pos)
    np.testing.assert_array_equal(fused.vel, eager.vel)
    np.testing.assert_array_equal(fused.time_counter, eager.time_counter)
    np.testing.assert_array_equal(fused.get_state(L.FS_FIELD_RING_LENGTH), eager.get_state(L.FS_FIELD_RING_LENGTH))
    assert (d.cpu().numpy() != 0).sum() >= R              # episodes did end (and were reset) inside the fragment
    # a second fragment continues the sampling streams
    o2, a2, lp2, r2, d2 = buffers(5, R, dev)
    fused.policy_rollout_dev(pol_a.struct, 5, o2, a2, lp2, r2, d2, reset_done=True)
    fused.sync()
    np.testing.assert_array_equal(o2[0].cpu().numpy(), o[K].cpu().numpy())
    assert not np.array_equal(a2[0].cpu().numpy(), a[0].cpu().numpy())
    fused.close(), eager.close()


def eager_obs0(sim):
    """Observation of the current state through the C ABI's zero-step form (what Env.reset returns without warm-up)."""
    import torch
    dev = torch.device("cuda", 0)
    o = torch.zeros((sim.R, sim.obs_dim), device=dev)
    r = torch.zeros((sim.R,), device=dev)
    d = torch.zeros((sim.R,), dtype=torch.uint8, device=dev)
    m = torch.zeros((sim.R,), dtype=torch.uint8, device=dev)          # nobody selected: nothing moves, everyone observed
    # (a masked reset of no replica: placement untouched, the observation of every replica written)
    sim.reset_dev(o, m)
    sim.sync()
    return o.cpu().numpy()


def test_network_matches_torch_and_samples_are_gaussian():
    import torch
    R = 4096
    dev = torch.device("cuda", 0)
    spec = rl_ring_spec(R=R, N=22, seed=5)
    sim = make(spec, "f32")
    sim.reset()
    pol = make_policy(3, False, seed=9)
    obs = (torch.rand((R, 3), device=dev) * 2 - 1) * torch.tensor([1.0, 0.3, 0.5], device=dev)
    act, logp = torch.zeros(R, device=dev), torch.zeros(R, device=dev)
    torch.cuda.synchronize()
    sim.policy_act_dev(pol.struct, obs, act, logp)
    sim.sync()
    with torch.no_grad():
        mu, ls = pol.reference(obs)
    g = (act - mu) / ls.exp()                                  # the draws behind the actions
    lp_ref = -0.5 * g * g - ls - 0.9189385332
    np.testing.assert_allclose(logp.cpu().numpy(), lp_ref.cpu().numpy(), atol=2e-3, rtol=0)
    gn = g.cpu().numpy()
    assert abs(gn.mean()) < 0.06 and abs(gn.std() - 1.0) < 0.05 and np.abs(gn).max() < 6
    # a second call draws again; the network itself: freeze the noise by comparing means through a zero-std policy
    act2, logp2 = torch.zeros(R, device=dev), torch.zeros(R, device=dev)
    torch.cuda.synchronize()
    sim.policy_act_dev(pol.struct, obs, act2, logp2)
    sim.sync()
    assert not torch.equal(act, act2)
    free = make_policy(3, True, seed=9)
    with torch.no_grad():
        free.log_std_param.fill_(-30.0)                          # std ~ 1e-13: the action is the mean
    free.sync()
    torch.cuda.synchronize()
    sim.policy_act_dev(free.struct, obs, act, logp)
    sim.sync()
    with torch.no_grad():
        mu_f, _ = free.reference(obs)
    np.testing.assert_allclose(act.cpu().numpy(), mu_f.cpu().numpy(), atol=2e-5, rtol=0)
    sim.close()


def test_fragment_simulator_is_the_oracles():
    """Replaying the fragment's own actions through oracle/refsim.py reproduces its observations and rewards."""
    import torch
    K, R = 60, 6
    dev = torch.device("cuda", 0)
    spec = rl_ring_spec(R=R, N=22, noise=0.0, warmup=0, horizon=500, seed=2)
    sim, ora = make(spec, "f32"), S.RingOracle(spec, np.float32)
    sim.reset()
    o_ref = ora.reset()
    pol = make_policy(3, False, seed=1)
    o, a, lp, r, d = buffers(K, R, dev)
    sim.policy_rollout_dev(pol.struct, K, o, a, lp, r, d, reset_done=False)
    sim.sync()
    on, an, rn = o.cpu().numpy(), a.cpu().numpy(), r.cpu().numpy()
    np.testing.assert_array_equal(on[0], o_ref.astype(np.float32))
    for k in range(K):
        o_ref, r_ref, d_ref = ora.step(an[k].reshape(R, 1))
        np.testing.assert_array_equal(on[k + 1], o_ref.astype(np.float32))
        np.testing.assert_array_equal(rn[k], r_ref.astype(np.float32))
    np.testing.assert_array_equal(sim.pos, ora.x)
    sim.close()


def test_unsupported_configurations_are_refused_by_name():
    import torch
    from helpers import ring_spec
    dev = torch.device("cuda", 0)
    sim = make(ring_spec(R=4, N=22, junction_length=0.1), "f32")          # AccelEnv, no RL vehicle
    pol = make_policy(3, False)
    o, a, lp, r, d = buffers(3, 4, dev)
    with pytest.raises(NotImplementedError, match="WaveAttenuationPOEnv"):
        sim.policy_rollout_dev(pol.struct, 3, o, a, lp, r, d)
    sim.close()


# ------------------------------------------------------------------ the figure eight (k_loop_policy)
def fig8_rl_spec(R, head, noise, horizon, seed):
    """BASELINE's C3 population (examples/exp_configs/rl/singleagent/singleagent_figure_eight.py): 13 IDM + 1 RL vehicle on
    the figure eight, obey_safe_speed; head 'po' = WaveAttenuationPOEnv (BASELINE's pairing), 'accel' = AccelEnv (the
    reference's own)."""
    from helpers import figure_eight_spec, idm_vehicle
    spec = figure_eight_spec(R=R, N=14, horizon=horizon, seed=seed, num_rl=1, action_low=-3.0, action_high=3.0,
                             env=S.ENV_WAVE_ATTENUATION_PO if head == "po" else S.ENV_ACCEL, po_max_length=421.94,
                             track_aux=False)
    spec["seed"] = 40 + seed
    spec["vehicles"] = [idm_vehicle(speed_mode=1, max_decel=1.5, noise=noise) for _ in range(13)] + \
                       [idm_vehicle(controller=S.CTRL_RL, rl_index=0, speed_mode=1, max_decel=1.5)]
    return spec


def make_policy_in(in_dim, num_hidden=3, free_log_std=False, seed=0, dev="cuda:0"):
    import torch
    from flow_amd.utils.device_policy import DevicePolicy
    g = torch.Generator().manual_seed(seed)
    dims = [in_dim] + [32] * num_hidden
    hidden = [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(num_hidden)]
    head = torch.nn.Linear(32, 1 if free_log_std else 2)
    for l in hidden + [head]:
        with torch.no_grad():
            l.weight.copy_(torch.randn(l.weight.shape, generator=g) * (0.4 if l.in_features <= 4 or l is head else 0.25))
            l.bias.copy_(torch.randn(l.bias.shape, generator=g) * 0.2)
    with torch.no_grad():
        head.weight.mul_(0.3)
    for l in hidden + [head]:
        l.to(dev)
    ls = torch.nn.Parameter(torch.tensor([-0.7], device=dev)) if free_log_std else None
    return DevicePolicy(hidden, head, log_std=ls, seed=77 + seed)


@pytest.mark.parametrize("head,noise,num_hidden,free", [("po", 0.2, 3, False), ("accel", 0.2, 3, False), ("po", 0.0, 1, True),
                                                          ("accel", 0.0, 2, True)])
def test_figure_eight_fused_fragment_equals_eager_stepping(head, noise, num_hidden, free):
    import torch
    K, R = 90, 7
    dev = torch.device("cuda", 0)
    spec = fig8_rl_spec(R, head, noise, horizon=35, seed=4)       # every replica finishes episodes inside the fragment
    D = 3 if head == "po" else 28
    pol_a, pol_b = make_policy_in(D, num_hidden, free, seed=3), make_policy_in(D, num_hidden, free, seed=3)
    fused, eager = make(spec, "f32"), make(spec, "f32")
    fused.reset(), eager.reset()

    def bufs():
        out = (torch.zeros((K + 1, R, D), device=dev), torch.zeros((K, R), device=dev), torch.zeros((K, R), device=dev),
               torch.zeros((K, R), device=dev), torch.zeros((K, R), dtype=torch.uint8, device=dev))
        torch.cuda.synchronize()
        return out
    o, a, lp, r, d = bufs()
    fused.policy_rollout_dev(pol_a.struct, K, o, a, lp, r, d, reset_done=True)
    fused.sync()
    assert fused.last_kernel == "k_loop_policy"
    eo, ea, elp, er, ed = bufs()
    eo[0].copy_(torch.as_tensor(eager_obs0(eager), device=dev))
    torch.cuda.synchronize()
    for k in range(K):
        eager.policy_act_dev(pol_b.struct, eo[k], ea[k], elp[k])
        eager.step_dev(eo[k + 1], er[k], ed[k], ea[k].reshape(R, 1))
        eager.reset_dev(eo[k + 1], ed[k])
    eager.sync()
    for name, x, y in (("obs", o, eo), ("act", a, ea), ("logp", lp, elp), ("rew", r, er), ("done", d, ed)):
        np.testing.assert_array_equal(x.cpu().numpy(), y.cpu().numpy(), err_msg=name)
    np.testing.assert_array_equal(fused.pos, eager.pos)
    np.testing.assert_array_equal(fused.vel, eager.vel)
    np.testing.assert_array_equal(fused.time_counter, eager.time_counter)
    assert (d.cpu().numpy() != 0).sum() >= 2 * R
    fused.close(), eager.close()


def test_wide_first_layer_matches_torch():
    """AccelEnv's 28 observations through policy_eval<., WIDE>: the mean of a (nearly) zero-std policy is torch's."""
    import torch
    R = 512
    dev = torch.device("cuda", 0)
    sim = make(fig8_rl_spec(R, "accel", 0.0, horizon=100, seed=1), "f32")
    sim.reset()
    pol = make_policy_in(28, 3, True, seed=9)
    with torch.no_grad():
        pol.log_std_param.fill_(-30.0)
    pol.sync()
    obs = torch.rand((R, 28), device=dev)
    act, logp = torch.zeros(R, device=dev), torch.zeros(R, device=dev)
    torch.cuda.synchronize()
    sim.policy_act_dev(pol.struct, obs, act, logp)
    sim.sync()
    with torch.no_grad():
        mu, _ = pol.reference(obs)
    np.testing.assert_allclose(act.cpu().numpy(), mu.cpu().numpy(), atol=3e-5, rtol=0)
    sim.close()
