"""Closed-loop use of the batched simulator: the HIP-graph fragment (VecFlowEnv.capture), the
RLlib-shaped FlowVectorEnv adapter (flow/utils/rllib.py / examples/train.py:110-212 of the reference give every
rollout worker its own SUMO; here the workers are replicas) and the on-device training example."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))


def flow_params(horizon=60):
    import train_vec
    return train_vec.ring_flow_params(horizon)


def test_captured_fragment_bit_exact_without_noise():
    import torch
    from flow_amd.controllers import ContinuousRouter, IDMController, RLController
    from flow_amd.core.params import VehicleParams
    from flow_amd.envs import VecFlowEnv
    fp = flow_params(horizon=40)
    veh = VehicleParams()
    veh.add(veh_id="human", acceleration_controller=(IDMController, {}), routing_controller=(ContinuousRouter, {}),
            num_vehicles=21)
    veh.add(veh_id="rl", acceleration_controller=(RLController, {}), routing_controller=(ContinuousRouter, {}),
            num_vehicles=1)
    fp["veh"] = veh
    R, K = 33, 16
    lin = torch.nn.Linear(3, 1).to("cuda")

    def policy(obs):
        with torch.no_grad():
            return torch.tanh(lin(obs))

    a = VecFlowEnv(fp, num_replicas=R, device=0)
    g = a.capture(K, policy=policy, reset_done=True)
    g.begin(a.reset())
    b = VecFlowEnv(fp, num_replicas=R, device=0)
    ob = b.reset().clone()
    for frag in range(4):                             # 64 steps: crosses the 40-step horizon -> in-graph resets
        obs, act, rew, done = g.replay()
        g.synchronize()
        for k in range(K):
            assert torch.equal(obs[k], ob)
            act_b = policy(ob)
            assert torch.equal(act[k], act_b)
            o2, r2, d2 = b.step(act_b)
            assert torch.equal(rew[k], r2) and torch.equal(done[k], d2)
            if bool(d2.any()):
                b.reset_done()
            ob = b._obs.clone()
            assert torch.equal(obs[k + 1], ob)
        assert int(done.sum()) == (R if frag == 2 else 0)
    a.close(), b.close()


def test_flow_vector_env_has_the_rllib_vector_env_interface():
    from flow_amd.utils.vector_env import FlowVectorEnv
    env = FlowVectorEnv(flow_params(horizon=5), num_envs=6, seed=2)
    obs = env.vector_reset()
    assert len(obs) == 6 and obs[0].shape == (3,)
    for t in range(5):
        obs, rew, done, info = env.vector_step([[0.1]] * 6)
        assert len(obs) == len(rew) == len(done) == len(info) == 6 and isinstance(rew[0], float)
    assert all(done)
    o3 = env.reset_at(3)
    assert o3.shape == (3,) and len(env.get_sub_environments()) == 1
    obs, rew, done, info = env.vector_step([[0.0]] * 6)
    assert not done[3]                                # replica 3 started a new episode, the others are past theirs
    assert env.action_space.shape == (1,) and env.observation_space.shape == (3,)
    env.close()


def test_train_vec_example_runs_on_the_device():
    import train_vec
    hist = train_vec.main(["--replicas", "64", "--fragment", "20", "--horizon", "50", "--iterations", "3",
                           "--epochs", "1"])
    assert len(hist) == 3 and all(np.isfinite(hist))


def test_train_script_has_the_reference_command_line_and_trains_reference_experiments():
    """examples/train.py EXP_CONFIG [--rl_trainer --num_steps --rollout_size ...] (the reference's train.py:34-71):
    the two single-agent experiments written with the reference's own parameter values run on the device (the ring one
    with its 750 warm-up steps inside the in-graph resets), the rllib route fails loudly without ray."""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "train.py")
    for exp in ("singleagent_figure_eight", "singleagent_ring"):
        res = subprocess.run([sys.executable, script, exp, "--num_steps", "2", "--rollout_size", "15", "--replicas", "48"],
                             capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        lines = [ln for ln in res.stdout.splitlines() if ln.startswith("iteration")]
        assert len(lines) == 2 and all(np.isfinite(float(ln.split()[5])) for ln in lines), res.stdout
        # both roll out on a fused policy + step kernel: the figure eight (AccelEnv, 28 observations) on k_loop_policy
        fused = "k_loop_policy" if exp == "singleagent_figure_eight" else "k_ring_policy"
        assert "fused policy + step kernel (%s)" % fused in res.stdout, res.stdout
    res = subprocess.run([sys.executable, script, "singleagent_ring", "--rl_trainer", "rllib"], capture_output=True,
                         text=True, timeout=600)
    try:
        import ray  # noqa: F401
    except ImportError:
        assert res.returncode != 0 and "ray" in res.stderr


def test_a_large_handle_on_the_generic_kernel_warns_once_and_names_the_field():
    """the 4-7x cliff from the rollout kernels of a closed loop to the generic k_steps is not silent
    at the replica counts where it matters; a handle on a rollout kernel says nothing."""
    import warnings
    import torch
    from flow_amd.controllers import ContinuousRouter, IDMController, RLController
    from flow_amd.core.params import VehicleParams
    from flow_amd.envs import VecFlowEnv
    fp = flow_params(horizon=20)
    veh = VehicleParams()
    veh.add(veh_id="human", acceleration_controller=(IDMController, {"fail_safe": "safe_velocity"}),
            routing_controller=(ContinuousRouter, {}), num_vehicles=21)
    veh.add(veh_id="rl", acceleration_controller=(RLController, {}), routing_controller=(ContinuousRouter, {}),
            num_vehicles=1)
    slow = VecFlowEnv(dict(fp, veh=veh), num_replicas=1024, device=0)
    slow.reset()
    act = torch.zeros((1024, 1), dtype=torch.float32, device="cuda")
    with pytest.warns(UserWarning, match=r"generic kernel k_steps.*fail_safe"):
        slow.rollout(4, actions=act)
    assert slow.sim.last_kernel in VecFlowEnv.GENERIC_KERNELS and "fail_safe" in slow.why_generic()
    with warnings.catch_warnings():
        warnings.simplefilter("error")                # once per handle
        slow.rollout(4, actions=act)
    slow.close()
    fast = VecFlowEnv(fp, num_replicas=1024, device=0)
    fast.reset()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        fast.rollout(4, actions=act)
    assert fast.sim.last_kernel not in VecFlowEnv.GENERIC_KERNELS and fast.why_generic() == []
    fast.close()


def test_train_vec_through_rccl_with_one_rank_equals_the_plain_run():
    """The N > 1 code path of examples/train_vec.py (process group over nccl = RCCL, flat gradient all-reduce, the global
    advantage sums) with ONE rank -- what a one-GPU box can run of it -- prints the history of the plain run, digit for
    digit (tests/test_train_dist_gloo.py holds the two-rank bit-for-bit statement on gloo)."""
    import subprocess
    script = os.path.join(ROOT, "examples", "train_vec.py")
    argv = [sys.executable, script, "--replicas", "64", "--fragment", "20", "--horizon", "50", "--iterations", "3",
            "--epochs", "2"]

    def rewards(env):
        res = subprocess.run(argv, capture_output=True, text=True, timeout=600, env=env)
        assert res.returncode == 0, res.stderr[-2000:]
        rows = [ln.split("rollout")[0] for ln in res.stdout.splitlines() if ln.startswith("iteration")]
        assert len(rows) == 3, res.stdout
        return rows

    plain = rewards(dict(os.environ))
    forced = rewards(dict(os.environ, TRAIN_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29561",
                          HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert plain == forced


def test_train_script_trains_the_shared_policy_of_the_multi_agent_experiments():
    """examples/train.py multiagent_ring / multiagent_figure_eight (every agent maps to the policy 'av' in the reference's
    files): one observation block and one action column per agent, ONE policy, the shared reward."""
    import subprocess
    script = os.path.join(ROOT, "examples", "train.py")
    for exp in ("multiagent_ring",):                      # (multiagent_figure_eight: the same path; RUN_SLOW covers nothing more)
        res = subprocess.run([sys.executable, script, exp, "--num_steps", "2", "--rollout_size", "12", "--replicas", "32"],
                             capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        lines = [ln for ln in res.stdout.splitlines() if ln.startswith("iteration")]
        assert len(lines) == 2 and all(np.isfinite(float(ln.split()[5])) for ln in lines), res.stdout
        assert "shared by 2 agents" in res.stdout
    res = subprocess.run([sys.executable, script, "adversarial_figure_eight", "--num_steps", "1"], capture_output=True,
                         text=True, timeout=600)
    assert res.returncode != 0 and "one shared policy" in res.stderr
