"""k_ring_pair (flow_amd/csrc/flowsim_ringrl.h): rings of IDMControllers + RLControllers on the two-vehicles-per-lane
kernel -- the reference's RL ring experiment (examples/exp_configs/rl/singleagent/singleagent_ring.py:17-65:
21 x IDMController(noise=0.2) + 1 x RLController, WaveAttenuationPOEnv, ring length 220..270, warm-up steps).

float32: bit-exact against the numpy oracle (no noise) and against the generic kernel k_steps (with noise: the two
kernels share the hardware's log / cos).  FS_MIXED: bit-exact against its C twin (oracle/csim/refsim_rl.c) and within
1e-4 of the float64 oracle -- the reference's arithmetic -- after 1500 steps of a fixed action tape."""
import os

import numpy as np
import pytest

from helpers import idm_vehicle, ring_spec
from oracle import cbuild
from oracle import refsim as S

pytestmark = pytest.mark.gpu


def make(spec, precision, **env):
    from flow_amd.sim import FlowSim
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return FlowSim(spec, precision=precision)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def rl_ring_spec(R=9, N=22, n_rl=1, po=True, noise=0.0, speed_mode=25, warmup=0, horizon=400, seed=0, clip=False,
                 lengths=(220, 270), min_gap=0.0):
    """The reference's RL ring: N - n_rl IDM humans (minGap 0, default speed mode 'right_of_way' = 25) followed by n_rl RL
    vehicles; one ring length per replica drawn from `lengths`, InitialConfig(bunching=50) placement for that length."""
    rng = np.random.default_rng(seed)
    Ls = rng.integers(lengths[0], lengths[1] + 1, R).astype(np.float64)
    pos = np.stack([ring_spec(R=1, N=N, length=float(L), bunching=50, junction_length=0.1)["init_pos"][0] for L in Ls])
    pos = pos + np.abs(rng.normal(0, 0.3, pos.shape))
    spec = ring_spec(R=R, N=N, length=260.0, bunching=50, junction_length=0.1, horizon=horizon,
                     env=S.ENV_WAVE_ATTENUATION_PO if po else S.ENV_ACCEL, num_rl=n_rl, action_low=-1.0, action_high=1.0,
                     po_max_length=float(lengths[1]), warmup_steps=warmup, clip_actions=clip, seed=1234)
    spec["ring_length"] = Ls
    spec["init_pos"] = pos
    veh = [idm_vehicle(sumo_min_gap=min_gap, noise=noise, speed_mode=speed_mode) for _ in range(N - n_rl)]
    # RL vehicles: rl ids sorted lexicographically give the columns (S2); here simply in slot order
    veh += [idm_vehicle(controller=S.CTRL_RL, rl_index=k, speed_mode=speed_mode) for k in range(n_rl)]
    spec["vehicles"] = veh
    return spec


def tape(K, R, n_rl, seed=5, scale=1.3):
    return np.random.default_rng(seed).uniform(-scale, scale, (K, R, n_rl)).astype(np.float32)


def rollout(sim, K, actions, obs_every_step=True):
    import torch
    dev = torch.device("cuda", 0)
    o = torch.zeros((K if obs_every_step else 1, sim.R, sim.obs_dim), device=dev)
    r = torch.zeros((K if obs_every_step else 1, sim.R), device=dev)
    d = torch.zeros((K if obs_every_step else 1, sim.R), dtype=torch.uint8, device=dev)
    a = None if actions is None else torch.as_tensor(actions, device=dev).contiguous()
    torch.cuda.synchronize()          # (the handle launches on a stream of its own)
    sim.rollout_dev(K, o, r, d, actions=a, action_stride_steps=None if a is None else sim.R * sim.num_rl,
                    obs_every_step=obs_every_step)
    sim.sync()
    return o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy()


@pytest.mark.parametrize("po,n_rl,N,clip", [(True, 1, 22, False), (True, 1, 22, True), (False, 1, 22, True),
                                              (True, 3, 14, True), (False, 2, 8, False), (True, 1, 40, False)])
def test_f32_rollout_bit_exact_against_the_oracle(po, n_rl, N, clip):
    K, R = 120, 7
    spec = rl_ring_spec(R=R, N=N, n_rl=n_rl, po=po, clip=clip, lengths=(220, 270) if N <= 22 else (420, 470), seed=N)
    acts = tape(K, R, n_rl, seed=N + 1)
    sim, ora = make(spec, "f32"), S.RingOracle(spec, np.float32)
    np.testing.assert_array_equal(sim.reset(), ora.reset().astype(np.float32))
    o, r, d = rollout(sim, K, acts)
    assert sim.last_kernel.startswith("k_ring_pair"), sim.last_kernel
    for k in range(K):
        o_ref, r_ref, d_ref = ora.step(acts[k])
        np.testing.assert_array_equal(o[k], o_ref.astype(np.float32), err_msg="obs, step %d" % k)
        np.testing.assert_array_equal(r[k], r_ref.astype(np.float32), err_msg="reward, step %d" % k)
        np.testing.assert_array_equal(d[k] != 0, d_ref)
    np.testing.assert_array_equal(sim.pos, ora.x)
    np.testing.assert_array_equal(sim.vel, ora.v)
    np.testing.assert_array_equal(sim.time_counter, ora.time_counter)
    sim.close()


def test_f32_reference_experiment_with_noise_equals_the_generic_kernel():
    """singleagent_ring as shipped (noise 0.2, speed mode 25, warm-up, ring length per replica): rollout on k_ring_pair
    == the same handle forced onto the generic kernel, bit for bit, across two launches that split a Philox block."""
    K, R = 150, 11
    spec = rl_ring_spec(R=R, N=22, noise=0.2, warmup=30, seed=3)
    acts = tape(K, R, 1, seed=9)
    fast, slow = make(spec, "f32"), make(spec, "f32", FLOWSIM_NO_RING_RL=1)
    np.testing.assert_array_equal(fast.reset(), slow.reset())                 # (warm-up steps: RL vehicle uncommanded)
    assert fast.last_kernel.startswith("k_ring_pair") and slow.last_kernel.startswith("k_steps")
    for k0, k1 in ((0, 37), (37, 150)):
        a, b = rollout(fast, k1 - k0, acts[k0:k1]), rollout(slow, k1 - k0, acts[k0:k1])
        for u, w in zip(a, b):
            np.testing.assert_array_equal(u, w)
        np.testing.assert_array_equal(fast.pos, slow.pos)
        np.testing.assert_array_equal(fast.vel, slow.vel)
    assert np.abs(a[1]).max() > 0 and fast.vel.max() > 1.0
    # single steps (fs_step) land on the same kernel and continue the same trajectory
    for k in range(5):
        oa, ra, da = fast.step(acts[k])
        ob, rb, db = slow.step(acts[k])
        np.testing.assert_array_equal(oa, ob)
        np.testing.assert_array_equal(ra, rb)
    assert fast.last_kernel.startswith("k_ring_pair")
    fast.close(), slow.close()


def test_masked_reset_with_warmup_and_last_step_observation():
    K, R = 40, 10
    spec = rl_ring_spec(R=R, N=22, warmup=12, seed=4, horizon=25)
    acts = tape(K, R, 1, seed=2)
    sim, ora = make(spec, "f32"), S.RingOracle(spec, np.float32)
    np.testing.assert_array_equal(sim.reset(), ora.reset().astype(np.float32))
    o, r, d = rollout(sim, 25, acts[:25], obs_every_step=False)                # observation of the last step only
    for k in range(25):
        o_ref, r_ref, d_ref = ora.step(acts[k])
    np.testing.assert_array_equal(o[0], o_ref.astype(np.float32))
    np.testing.assert_array_equal(r[0], r_ref.astype(np.float32))
    assert (d[0] != 0).all() and d_ref.all()
    mask = np.zeros(R, np.uint8)
    mask[[0, 3, 4, 9]] = 1
    np.testing.assert_array_equal(sim.reset(mask), ora.reset(mask.astype(bool)).astype(np.float32))
    assert sim.last_kernel.startswith("k_ring_pair")
    np.testing.assert_array_equal(sim.pos, ora.x)
    np.testing.assert_array_equal(sim.time_counter, ora.time_counter)
    o, r, d = rollout(sim, 10, acts[25:35])
    for k in range(10):
        o_ref, r_ref, d_ref = ora.step(acts[25 + k])
        np.testing.assert_array_equal(o[k], o_ref.astype(np.float32))
        np.testing.assert_array_equal(r[k], r_ref.astype(np.float32))
    np.testing.assert_array_equal(sim.vel, ora.v)
    sim.close()


@pytest.mark.parametrize("po", [True, False])
def test_mixed_bit_exact_against_its_c_twin_and_within_1e_4_of_float64(po):
    """FS_MIXED with an RL vehicle: reset with warm-up steps, then 1500 steps of a fixed action tape."""
    K, R = 1500, 6
    spec = rl_ring_spec(R=R, N=22, po=po, warmup=50, seed=7, horizon=3000)
    acts = tape(K, R, 1, seed=11, scale=0.8)
    sim, twin, ref = make(spec, "mixed"), cbuild.CRingRLMixed(spec), S.RingOracle(spec, np.float64)
    np.testing.assert_array_equal(sim.reset(), twin.reset())
    assert sim.last_kernel.startswith("k_ring_pair")
    ref.reset()
    o, r, d = rollout(sim, K, acts)
    to, tr, td = twin.rollout(K, acts)
    np.testing.assert_array_equal(o, to)
    np.testing.assert_array_equal(r, tr)
    np.testing.assert_array_equal(d, td)
    np.testing.assert_array_equal(sim.pos, twin.x)
    np.testing.assert_array_equal(sim.vel, twin.v)
    for k in range(K):
        ref.step(acts[k].astype(np.float64))
    L = spec["ring_length"][:, None] + 0.4
    dx = np.abs(sim.pos - ref.x)
    dx = np.minimum(dx, L - dx)
    assert dx.max() < 1e-4 and np.abs(sim.vel - ref.v).max() < 1e-4, (dx.max(), np.abs(sim.vel - ref.v).max())
    assert ref.v.max() > 1.0
    # the float32 twin of the same run drifts past the bar (why FS_MIXED exists)
    f32 = make(spec, "f32")
    f32.reset()
    rollout(f32, K, acts)
    d32 = np.abs(f32.pos - ref.x)
    assert np.minimum(d32, L - d32).max() > dx.max()
    f32.close()
    # masked reset on the mixed handle
    mask = np.array([1, 0, 0, 1, 0, 1], np.uint8)
    np.testing.assert_array_equal(sim.reset(mask), twin.reset(mask))
    np.testing.assert_array_equal(sim.pos, twin.x)
    sim.close()


def test_pending_ring_length_is_taken_at_the_next_reset_only():
    """FS_FIELD_INIT_RING_LENGTH: what VecFlowEnv.redraw_ring_lengths writes between replays of a captured fragment."""
    from flow_amd import _lib as L
    R = 6
    spec = rl_ring_spec(R=R, N=22, warmup=0, seed=12)
    sim = make(spec, "f32")
    sim.reset()
    acts = tape(20, R, 1, seed=1)
    before = sim.get_state(L.FS_FIELD_RING_LENGTH).copy()
    new_len = before + 7.0
    sim.set_state(L.FS_FIELD_INIT_RING_LENGTH, new_len)
    rollout(sim, 20, acts)
    np.testing.assert_array_equal(sim.get_state(L.FS_FIELD_RING_LENGTH), before)          # mid-episode: untouched
    mask = np.array([1, 1, 0, 0, 1, 0], np.uint8)
    sim.reset(mask)
    want = np.where(mask != 0, new_len, before)
    np.testing.assert_array_equal(sim.get_state(L.FS_FIELD_RING_LENGTH), want)
    # and the trajectory follows the new length: the oracle on the mixed lengths agrees
    spec2 = dict(spec, ring_length=want)
    ora = S.RingOracle(spec2, np.float32)
    ora.reset()
    ora.x[mask == 0] = sim.pos[mask == 0]
    ora.v[mask == 0] = sim.vel[mask == 0]
    ora.time_counter[mask == 0] = sim.time_counter[mask == 0]
    o, r, d = rollout(sim, 20, acts)
    for k in range(20):
        o_ref, r_ref, d_ref = ora.step(acts[k])
    np.testing.assert_array_equal(sim.pos, ora.x)
    np.testing.assert_array_equal(o[-1], o_ref.astype(np.float32))
    sim.close()


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_fuzz_ring_pair_equals_generic_kernel(seed):
    """Random RL rings (vehicle count, RL count and places, noise, speed modes, clipping, lengths, per-slot parameters,
    launch lengths that split the 16-step groups and the Philox blocks): k_ring_pair == k_steps bit for bit -- observations,
    rewards, done flags of every step and the state after every launch."""
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.choice([8, 14, 22, 22, 30]))
    n_rl = int(rng.choice([1, 1, 1, 2, 3]))
    po = bool(rng.integers(0, 2))
    R = int(rng.integers(3, 14))
    spec = rl_ring_spec(R=R, N=N, n_rl=n_rl, po=po, noise=float(rng.choice([0.0, 0.2, 0.5])),
                        speed_mode=int(rng.choice([0, 1, 25, 31])), warmup=int(rng.choice([0, 0, 17])),
                        clip=bool(rng.integers(0, 2)), lengths=(8 * N + 60, 10 * N + 66), seed=seed,
                        min_gap=float(rng.choice([0.0, 2.5])))
    # per-slot IDM parameters and lengths; RL vehicles at random places (columns in slot order)
    veh = spec["vehicles"]
    for v in veh:
        if v["controller"] != S.CTRL_RL:
            v["p"] = [float(rng.uniform(20, 35)), float(rng.uniform(0.8, 1.5)), float(rng.uniform(0.8, 1.6)),
                      float(rng.uniform(1.0, 2.5)), 4.0, float(rng.uniform(1.5, 3.0))] + list(v["p"][6:])
        v["length"] = float(rng.choice([4.0, 5.0, 6.5]))
    order = rng.permutation(N)
    veh = [veh[j] for j in order]
    col = 0
    for v in veh:
        if v["controller"] == S.CTRL_RL:
            v["rl_index"] = col
            col += 1
    spec["vehicles"] = veh
    K = int(rng.integers(40, 130))
    acts = tape(K, R, n_rl, seed=seed + 50, scale=float(rng.choice([0.8, 1.6])))
    fast, slow = make(spec, "f32"), make(spec, "f32", FLOWSIM_NO_RING_RL=1)
    np.testing.assert_array_equal(fast.reset(), slow.reset())
    cut = int(rng.integers(1, K - 1))
    for k0, k1 in ((0, cut), (cut, K)):
        a, b = rollout(fast, k1 - k0, acts[k0:k1]), rollout(slow, k1 - k0, acts[k0:k1])
        assert fast.last_kernel.startswith("k_ring_pair") and slow.last_kernel.startswith("k_steps"), \
            (fast.last_kernel, slow.last_kernel)
        for u, w, what in zip(a, b, ("obs", "reward", "done")):
            np.testing.assert_array_equal(u, w, err_msg="%s, launch %d..%d" % (what, k0, k1))
        np.testing.assert_array_equal(fast.pos, slow.pos)
        np.testing.assert_array_equal(fast.vel, slow.vel)
    fast.close(), slow.close()


def test_mixed_with_noise_holds_1e_4_against_the_float64_kernel_where_float32_does_not():
    """The reference's RL ring experiment AS SHIPPED -- IDMController(noise=0.2) -- in FS_MIXED (float64 state, float32
    controller + float32 noise term).  The hardware's log / cos have no bit-twin on the CPU, so this form is held against
    the float64 generic kernel (the reference's arithmetic type) running the SAME Philox streams: 1e-4 m / m/s after 1500
    steps of a fixed action tape, which the float32 run of the same tape and streams misses; and the all-IDM noisy ring
    (no RL vehicle) takes the same kernel."""
    K, R = 1500, 16
    spec = rl_ring_spec(R=R, N=22, noise=0.2, warmup=0, horizon=K, seed=6)
    acts = tape(K, R, 1, seed=3, scale=1.0)
    runs = {}
    for prec in ("mixed", "f64", "f32"):
        sim = make(spec, prec)
        sim.reset()
        o, r, d = rollout(sim, K, acts)
        runs[prec] = (sim.pos.astype(np.float64), sim.vel.astype(np.float64), o, r, sim.last_kernel)
        sim.close()
    assert runs["mixed"][4].startswith("k_ring_pair") and runs["f64"][4].startswith("k_steps")
    Ls = np.asarray(spec["ring_length"])[:, None] + 0.4

    def ring_dist(a, b):
        dd = np.abs(a - b)
        return np.minimum(dd, Ls - dd)
    dx_m, dv_m = ring_dist(runs["mixed"][0], runs["f64"][0]).max(), np.abs(runs["mixed"][1] - runs["f64"][1]).max()
    dx_f = ring_dist(runs["f32"][0], runs["f64"][0]).max()
    assert dx_m < 1e-4 and dv_m < 1e-4, (dx_m, dv_m)
    assert dx_f > 3 * dx_m, (dx_f, dx_m)                       # float32 state is what loses the bar, not the noise
    assert np.abs(runs["mixed"][2] - runs["f64"][2]).max() < 1e-5 and np.abs(runs["mixed"][3] - runs["f64"][3]).max() < 1e-5
    assert runs["mixed"][1].max() > 1.0
    # all-IDM ring with noise: FS_MIXED steps it on k_ring_pair as well (k_rollout_pair's noisy form is float32 only)
    spec2 = rl_ring_spec(R=4, N=22, noise=0.2, po=False, seed=2)
    spec2["vehicles"] = [dict(v, controller=S.CTRL_IDM, rl_index=-1) if v["controller"] == S.CTRL_RL else v
                         for v in spec2["vehicles"]]
    spec2["num_rl"] = 0
    for v in spec2["vehicles"]:
        if v["controller"] == S.CTRL_IDM and not v.get("p"):
            v.update(idm_vehicle(noise=0.2, speed_mode=25, sumo_min_gap=0.0))
    a, b = make(spec2, "mixed"), make(spec2, "f64")
    a.reset(), b.reset()
    rollout(a, 300, None), rollout(b, 300, None)
    assert a.last_kernel.startswith("k_ring_pair"), a.last_kernel
    assert np.abs(a.pos - b.pos).max() < 1e-4
    a.close(), b.close()
