"""Experiment: the non-RL rollout runner of flow/core/experiment.py:13-198 on the GPU step loop."""
import datetime
import logging
import time

import numpy as np

from flow_amd.utils.registry import make_create_env


class Experiment:
    """Same contract as the reference: ``Experiment(flow_params, custom_callables).run(num_runs,
    rl_actions, convert_to_csv)`` returns ``info_dict`` with returns / velocities / outflows (+ custom)."""

    def __init__(self, flow_params, custom_callables=None):
        self.custom_callables = custom_callables or {}
        create_env, _ = make_create_env(flow_params)
        self.env = create_env()
        logging.info(" Starting experiment {} at {}".format(self.env.network.name, str(datetime.datetime.utcnow())))

    def run(self, num_runs, rl_actions=None, convert_to_csv=False):
        num_steps = self.env.env_params.horizon
        if convert_to_csv and self.env.sim_params.emission_path is None:      # experiment.py:108-117
            raise ValueError(
                'The experiment was run with convert_to_csv set to True, but no emission file will be '
                'generated. Set emission_path in the simulation parameters or convert_to_csv to False.')
        info_dict = {"returns": [], "velocities": [], "outflows": []}
        info_dict.update({key: [] for key in self.custom_callables.keys()})
        if rl_actions is None:
            def rl_actions(*_):
                return None
        t = time.time()
        times = []
        for i in range(num_runs):
            ret, vel = 0, []
            custom_vals = {key: [] for key in self.custom_callables.keys()}
            state = self.env.reset()
            for j in range(int(num_steps)):
                t0 = time.time()
                state, reward, done, _ = self.env.step(rl_actions(state))
                t1 = time.time()
                times.append(1 / (t1 - t0))
                veh_ids = self.env.k.vehicle.get_ids()
                vel.append(np.mean(self.env.k.vehicle.get_speed(veh_ids)))
                ret += reward
                for (key, lambda_func) in self.custom_callables.items():
                    custom_vals[key].append(lambda_func(self.env))
                if done:
                    break
            info_dict["returns"].append(ret)
            info_dict["velocities"].append(np.mean(vel))
            info_dict["outflows"].append(self.env.k.vehicle.get_outflow_rate(int(500)))
            for key in custom_vals.keys():
                info_dict[key].append(np.mean(custom_vals[key]))
            print("Round {0}, return: {1}".format(i, ret))
        for key in info_dict.keys():
            print("Average, std {}: {}, {}".format(key, np.mean(info_dict[key]), np.std(info_dict[key])))
        print("Total time:", time.time() - t)
        print("steps/second:", np.mean(times))
        emission = self.env.write_emission() if convert_to_csv else None
        self.env.terminate()
        if emission is not None:
            info_dict["emission_csv"] = emission
        return info_dict
