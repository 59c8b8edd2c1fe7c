"""SumoParams(noise_math='exact') / spec['noise_math'] = 'exact' (VERDICT r03 item 7): the Box-Muller transform of the
acceleration noise (flow/controllers/base_controller.py:109-110) runs as fixed float32 operation sequences
(flowsim_kernels.h bm_ln_exact / bm_cos_exact = oracle/refsim.py exact_ln_f32 / exact_cos_turns_f32), so the NOISY
configurations -- the reference's experiments as shipped (IDMController(noise=0.2)) -- are held against the numpy oracle BIT
FOR BIT in float32, kernel by kernel; with the default hardware log2 / cos they agree to a few ulp of the draw only."""
import numpy as np
import pytest

from helpers import figure_eight_spec, idm_vehicle, merge_spec, ring_spec
from oracle import opennet as O
from oracle import refsim as S
from test_parity_gpu import _rollout

pytestmark = pytest.mark.gpu


def ring_case(kind, R=9):
    N = 22
    if kind == "all_idm":                                  # C2 with the human model of the RL experiments
        spec = ring_spec(R=R, N=N, horizon=200)
        spec["vehicles"] = [idm_vehicle(noise=0.2) for _ in range(N)]
        acts = None
    else:                                                  # C1: singleagent_ring.py (21 noisy IDM + 1 RL, PO head)
        spec = ring_spec(R=R, N=N, horizon=200, env=S.ENV_WAVE_ATTENUATION_PO, num_rl=1, po_max_length=270.0,
                         action_low=-1.0, action_high=1.0, clip_actions=False)
        spec["vehicles"] = [idm_vehicle(noise=0.2, sumo_min_gap=0.0) for _ in range(N - 1)] + \
                           [idm_vehicle(controller=S.CTRL_RL, rl_index=0)]
        acts = np.random.default_rng(2).uniform(-1, 1, (70, R, 1)).astype(np.float32)
    spec["ring_length"] = np.random.default_rng(1).uniform(220, 270, R)
    spec["init_pos"] = np.asarray(spec["init_pos"]) * (spec["ring_length"][:, None] / 230.0)
    spec.update(seed=123456789012, noise_math="exact")
    return spec, acts


@pytest.mark.parametrize("kind,kernel,env", [("all_idm", "k_rollout_pair+noise", None), ("rl", "k_ring_pair<PO>", None),
                                               ("all_idm", "k_steps", {"FLOWSIM_FORCE_GENERIC": "1"})])
def test_noisy_rings_equal_the_numpy_oracle_bit_for_bit(kind, kernel, env):
    K = 70
    spec, acts = ring_case(kind)
    sim, obs, rew, done = _rollout(spec, K, acts, env=env)
    assert sim.last_kernel.startswith(kernel)
    ora = S.RingOracle(spec, np.float32)
    ora.reset()
    for k in range(K):
        o_ref, r_ref, d_ref = ora.step(None if acts is None else acts[k])
        np.testing.assert_array_equal(obs[k], o_ref.astype(np.float32), err_msg="obs, step %d" % k)
        np.testing.assert_array_equal(rew[k], r_ref.astype(np.float32), err_msg="reward, step %d" % k)
    np.testing.assert_array_equal(sim.pos, ora.x)
    np.testing.assert_array_equal(sim.vel, ora.v)
    sim.close()
    # and the default math is NOT bit-equal (it is the hardware's log2 / cos): the flag does something
    hw, o_hw, _, _ = _rollout(dict(spec, noise_math="hw"), K, acts, env=env)
    assert not np.array_equal(o_hw, obs) and np.abs(o_hw - obs).max() < 1e-3
    hw.close()


def test_noisy_figure_eight_equals_the_numpy_oracle_bit_for_bit():
    R, N, K = 11, 14, 90
    spec = figure_eight_spec(R=R, N=N, horizon=200, seed=5, num_rl=1)
    spec["vehicles"] = [idm_vehicle(speed_mode=1, max_decel=1.5, noise=0.2) for _ in range(N - 1)] + \
                       [idm_vehicle(controller=S.CTRL_RL, rl_index=0, speed_mode=1, max_decel=1.5)]
    spec.update(seed=77, noise_math="exact")
    acts = np.random.default_rng(9).uniform(-3, 3, (K, R, 1)).astype(np.float32)
    sim, obs, rew, done = _rollout(spec, K, acts)
    assert sim.last_kernel == "k_rollout_loop<FULL>"
    ora = S.RingOracle(spec, np.float32)
    ora.reset()
    for k in range(K):
        o_ref, r_ref, d_ref = ora.step(acts[k])
        np.testing.assert_array_equal(obs[k], o_ref.astype(np.float32), err_msg="obs, step %d" % k)
        np.testing.assert_array_equal(rew[k], r_ref.astype(np.float32), err_msg="reward, step %d" % k)
    np.testing.assert_array_equal(sim.pos, ora.x)
    sim.close()


@pytest.mark.parametrize("env,kernel", [({}, "k_merge_queue"), ({"FLOWSIM_NO_QUEUE": "1"}, "k_steps_open")])
def test_noisy_merge_equals_the_numpy_oracle_bit_for_bit(env, kernel, monkeypatch):
    from test_open_gpu import compare_state, make
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    spec = merge_spec(R=5, cap_human=40, cap_rl=8, num_rl=8, horizon=150, seed=4, env=O.ENV_MERGE_MA, sims_per_step=2,
                      ma_apply_actions=True)
    spec.update(noise_math="exact")
    ora = O.MergeOracle(spec, np.float32)
    sim = make(spec, "f32")
    np.testing.assert_array_equal(sim.reset(), ora.reset().astype(np.float32))
    rng = np.random.default_rng(3)
    for k in range(150):
        a = rng.uniform(-1, 1, (5, 8)).astype(np.float32)
        o_ref, r_ref, d_ref = ora.step(a)
        o, r, d = sim.step(a)
        if k == 0:
            assert sim.last_kernel.startswith(kernel)
        np.testing.assert_array_equal(o, o_ref.astype(np.float32), err_msg="obs, step %d" % k)
        np.testing.assert_array_equal(r, r_ref.astype(np.float32), err_msg="reward, step %d" % k)
    compare_state(sim, ora)
    assert ora.total_arrived.min() > 5
    sim.close()


@pytest.mark.parametrize("po", [True, False])
def test_mixed_with_the_experiments_noise_equals_its_c_twin_bit_for_bit(po):
    """FS_MIXED (float64 state, float32 controllers) with IDMController(noise=0.2) -- the headline precision on the
    reference's RL ring experiment AS SHIPPED -- against oracle/csim/refsim_rl.c, which draws the same Philox words through the
    same exact Box-Muller sequences: reset with warm-up steps, 1500 steps of an action tape, a second fragment (the draw
    counter runs on) -- observations, rewards, done flags, positions and speeds bit for bit; and within 1e-4 of the float64
    oracle running the same noise (the draws are float32 there too: float64 has no exact form)."""
    from oracle import cbuild
    from test_ringrl_gpu import make, rl_ring_spec, rollout, tape
    K, R = 1500, 6
    spec = rl_ring_spec(R=R, N=22, po=po, noise=0.2, warmup=30, seed=7, horizon=4000)
    spec["noise_math"] = "exact"
    acts = tape(K, R, 1, seed=11, scale=0.8)
    sim, twin = make(spec, "mixed"), cbuild.CRingRLMixed(spec)
    np.testing.assert_array_equal(sim.reset(), twin.reset())
    assert sim.last_kernel.startswith("k_ring_pair")
    for frag in range(2):
        o, r, d = rollout(sim, K, acts)
        to, tr, td = twin.rollout(K, acts)
        np.testing.assert_array_equal(o, to, err_msg="fragment %d" % frag)
        np.testing.assert_array_equal(r, tr)
        np.testing.assert_array_equal(d, td)
        np.testing.assert_array_equal(sim.pos, twin.x)
        np.testing.assert_array_equal(sim.vel, twin.v)
    assert twin.nctr.min() == 30 + 2 * K and sim.vel.max() > 1.0
    sim.close()
