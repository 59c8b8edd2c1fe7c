#!/usr/bin/env python3
"""Runner for RL experiments on the GPU step loop -- the command line of the reference's examples/train.py:34-71.

    python examples/train.py singleagent_ring [--rl_trainer device|rllib] [--num_steps N] [--rollout_size K]
                                              [--num_cpus C] [--replicas R]

``--rl_trainer device`` (default here): PPO with the rollout on the device -- R replicas of the experiment in one
handle, a fragment of ``rollout_size`` closed-loop steps (policy, Env.step, reset of finished episodes) replayed as one
HIP graph, observations / actions / the update never leave HBM (examples/train_vec.py).  It takes the place of the
reference's RLlib rollout workers, each of which drives one SUMO process (train.py:149-212).
``--rl_trainer rllib``: the reference's own route -- register the environment (flow.utils.registry.make_create_env) and
hand it to ray.tune with the reference's PPO settings; needs ray, which this image does not ship.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import flow_amd  # noqa: E402

flow_amd.install_as_flow()


def parse_args(args):
    parser = argparse.ArgumentParser(description="Train an RL experiment from exp_configs/rl.",
                                     epilog="python train.py EXP_CONFIG")
    parser.add_argument('exp_config', type=str,
                        help='module in exp_configs/rl/singleagent or exp_configs/rl/multiagent')
    parser.add_argument('--rl_trainer', type=str, default="device", help='device (on-GPU PPO) or rllib')
    parser.add_argument('--num_cpus', type=int, default=1, help='rollout workers (rllib only)')
    parser.add_argument('--num_steps', type=int, default=20, help='training iterations')
    parser.add_argument('--rollout_size', type=int, default=100, help='env steps per rollout fragment')
    parser.add_argument('--checkpoint_path', type=str, default=None, help='rllib: checkpoint to restore')
    parser.add_argument('--replicas', type=int, default=1024, help='device: environment replicas in the handle')
    return parser.parse_known_args(args)[0]


def load_experiment(name):
    """The module of an experiment: exp_configs/rl/singleagent first, then multiagent (train.py:318-330)."""
    for pkg in ("exp_configs.rl.singleagent", "exp_configs.rl.multiagent"):
        try:
            module = __import__(pkg, fromlist=[name])
            return getattr(module, name), pkg.endswith("multiagent")
        except (ImportError, AttributeError):
            continue
    raise ValueError("Unable to find experiment config %r" % name)


def train_rllib(submodule, flags):
    """The reference's RLlib set-up (train.py:110-212) over make_create_env; FlowVectorEnv lets RLlib step all
    replicas of a worker in one call."""
    try:
        import ray
        from ray.tune import run_experiments
        from ray.tune.registry import register_env
        try:
            from ray.rllib.agents.agent import get_agent_class
        except ImportError:
            from ray.rllib.agents.registry import get_agent_class
    except ImportError as e:
        raise ImportError("--rl_trainer rllib needs ray[rllib], which is not installed here; use --rl_trainer device") from e
    from copy import deepcopy
    from flow.utils.registry import make_create_env
    from flow.utils.rllib import FlowParamsEncoder
    flow_params = submodule.flow_params
    horizon = flow_params['env'].horizon
    config = deepcopy(get_agent_class("PPO")._default_config)
    config.update(num_workers=flags.num_cpus, train_batch_size=horizon * submodule.N_ROLLOUTS, gamma=0.999,
                  use_gae=True, kl_target=0.02, num_sgd_iter=10, horizon=horizon)
    config["lambda"] = 0.97
    config["model"].update({"fcnet_hiddens": [32, 32, 32]})
    config['env_config']['flow_params'] = json.dumps(flow_params, cls=FlowParamsEncoder, sort_keys=True, indent=4)
    config['env_config']['run'] = "PPO"
    create_env, gym_name = make_create_env(params=flow_params)
    register_env(gym_name, create_env)
    ray.init(num_cpus=flags.num_cpus + 1)
    exp = {"run": "PPO", "env": gym_name, "config": config, "checkpoint_freq": 20, "checkpoint_at_end": True,
           "max_failures": 999, "stop": {"training_iteration": flags.num_steps}}
    if flags.checkpoint_path is not None:
        exp['restore'] = flags.checkpoint_path
    run_experiments({flow_params["exp_tag"]: exp})


def train_device(submodule, flags, multiagent=False):
    from train_vec import train_on_device
    fp = submodule.flow_params
    fp['sim'].render = False
    return train_on_device(fp, replicas=flags.replicas, fragment=flags.rollout_size, iterations=flags.num_steps,
                           shared_agents=multiagent)


def main(args):
    flags = parse_args(args)
    submodule, multiagent = load_experiment(flags.exp_config)
    if flags.rl_trainer.lower() == "rllib":
        return train_rllib(submodule, flags)
    if flags.rl_trainer.lower() == "device":
        # multi-agent experiments whose agents share the policy 'av' (multiagent_ring / _figure_eight / _merge: one
        # observation block and one action column per agent) train that ONE policy; adversarial_figure_eight has two
        # policies with opposite rewards: not this loop
        if multiagent and len(getattr(submodule, "POLICIES_TO_TRAIN", ["av"])) > 1:
            raise ValueError("--rl_trainer device trains one shared policy; %s trains %s"
                             % (flags.exp_config, submodule.POLICIES_TO_TRAIN))
        if multiagent and flags.exp_config.startswith("adversarial"):
            raise ValueError("--rl_trainer device trains one shared policy; adversarial experiments train two")
        return train_device(submodule, flags, multiagent)
    raise ValueError("rl_trainer should be either 'device' or 'rllib'.")


if __name__ == "__main__":
    main(sys.argv[1:])
