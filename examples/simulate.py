#!/usr/bin/env python3
"""Run a non-RL experiment on the GPU step loop (the reference's examples/simulate.py:11-77 contract):

    python examples/simulate.py ring --num_runs 1 [--gen_emission]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import flow_amd  # noqa: E402

flow_amd.install_as_flow()


def parse_args(args):
    parser = argparse.ArgumentParser(description="Run a non-RL experiment from exp_configs/non_rl.")
    parser.add_argument('exp_config', type=str, help='name of a module in exp_configs/non_rl (ring, figure_eight, merge, bottleneck)')
    parser.add_argument('--num_runs', type=int, default=1)
    parser.add_argument('--gen_emission', action='store_true', help='write the trajectory CSV under ./data')
    return parser.parse_known_args(args)[0]


if __name__ == "__main__":
    flags = parse_args(sys.argv[1:])
    module = __import__("exp_configs.non_rl", fromlist=[flags.exp_config])
    flow_params = getattr(module, flags.exp_config).flow_params
    flow_params['sim'].render = False
    if flags.gen_emission:
        flow_params['sim'].emission_path = os.path.join(os.getcwd(), "data")
    from flow.core.experiment import Experiment
    info = Experiment(flow_params).run(flags.num_runs, convert_to_csv=flags.gen_emission)
    print({k: v for k, v in info.items()})
