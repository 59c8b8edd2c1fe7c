"""Experiment: the non-RL rollout runner (interface of flow/core/experiment.py:13-198) over the GPU step loop.

Two ways to run ``num_runs`` episodes of ``env_params.horizon`` steps:

* **batched** (default whenever it is possible): the runs are the replicas of ONE ``VecFlowEnv`` and an episode is
  ONE rollout launch; returns, mean speeds and episode lengths are reduced on the device from the [K, R] reward /
  done planes and the [K, R, obs] observation plane.  Possible when nothing has to look inside the environment
  between two steps: no ``custom_callables``, no emission CSV, ``rl_actions`` absent (or a callable on the batched
  observation tensor, in which case the episode is stepped launch by launch, still on the device), and an
  environment with an in-kernel head whose observation carries the speeds (AccelEnv / WaveAttenuationEnv layout).
* **stepwise**: one scalar environment, one launch per step, host accessors (``env.k.vehicle``) available to the
  callables and to the trajectory recorder -- the reference's execution model.

Both return the reference's ``info_dict`` keys (``returns``, ``velocities``, ``outflows`` + one list per custom
callable; ``emission_csv`` when a trajectory file was written) and print its summary lines.
"""
import datetime
import logging
import time

import numpy as np

from flow_amd.utils.registry import make_create_env


class _RunLog(object):
    """Per-run scalars of one experiment and the summary the reference prints (experiment.py:172-179)."""

    def __init__(self, extra_keys):
        self.table = {"returns": [], "velocities": [], "outflows": []}
        for key in extra_keys:
            self.table[key] = []
        self.step_rates = []

    def close_run(self, index, ret, mean_speed, outflow, extras):
        self.table["returns"].append(ret)
        self.table["velocities"].append(mean_speed)
        self.table["outflows"].append(outflow)
        for key, values in extras.items():
            self.table[key].append(float(np.mean(values)) if len(values) else float("nan"))
        print("Round {0}, return: {1}".format(index, ret))

    def summary(self, wall_seconds):
        for key, values in self.table.items():
            print("Average, std {}: {}, {}".format(key, np.mean(values), np.std(values)))
        print("Total time:", wall_seconds)
        print("steps/second:", np.mean(self.step_rates) if self.step_rates else float("nan"))
        return self.table


class Experiment:
    """``Experiment(flow_params, custom_callables).run(num_runs, rl_actions=None, convert_to_csv=False)``."""

    def __init__(self, flow_params, custom_callables=None):
        self.flow_params = flow_params
        self.custom_callables = custom_callables or {}
        create_env, _ = make_create_env(flow_params)
        self.env = create_env()
        logging.info(" Starting experiment {} at {}".format(self.env.network.name, str(datetime.datetime.utcnow())))

    # ------------------------------------------------------------------------------------------ dispatch
    def run(self, num_runs, rl_actions=None, convert_to_csv=False, batched=None):
        if convert_to_csv and self.env.sim_params.emission_path is None:      # experiment.py:108-117
            raise ValueError(
                'The experiment was run with convert_to_csv set to True, but no emission file will be '
                'generated. Set emission_path in the simulation parameters or convert_to_csv to False.')
        can_batch = (not self.custom_callables and not convert_to_csv and self._speeds_in_observation()
                     and np.isfinite(self.env.env_params.horizon))
        if batched is None:
            batched = can_batch and rl_actions is None
        if batched and not can_batch:
            raise ValueError("batched=True needs an in-kernel AccelEnv-layout head, a finite horizon, no callables "
                             "and no emission file")
        return self._run_batched(num_runs, rl_actions) if batched else self._run_stepwise(
            num_runs, rl_actions, convert_to_csv)

    def _speeds_in_observation(self):
        from flow_amd import _lib as L
        env = self.env
        return (getattr(env, "FS_ENV", None) in (L.FS_ENV_ACCEL, L.FS_ENV_WAVE_ATTENUATION)
                and not getattr(env, "HOST_HEADS", False) and env.sim.obs_dim == 2 * env.sim.N)

    # ------------------------------------------------------------------------------------------ batched
    def _run_batched(self, num_runs, policy):
        import torch
        from flow_amd.envs import VecFlowEnv
        horizon = int(self.env.env_params.horizon)
        n_veh, v_max = self.env.sim.N, float(self.env.k.network.max_speed())
        log = _RunLog(())
        began = time.time()
        vec = VecFlowEnv(self.flow_params, num_replicas=int(num_runs), device=self.env.sim.device)
        obs0 = vec.reset()
        if policy is None:
            obs, rew, done = vec.rollout(horizon, None)               # [K, R, obs], [K, R], [K, R]
        else:
            planes, cur = [], obs0
            for _ in range(horizon):
                cur, r, d = vec.step(policy(cur))
                planes.append((cur.clone(), r.clone(), d.clone()))
            obs, rew, done = (torch.stack(p) for p in zip(*planes))
        # a run stops counting after its first done step (experiment.py:160-161 `if done: break`)
        finished_before = torch.cumsum(done.to(torch.int32), dim=0) - done.to(torch.int32) > 0
        live = (~finished_before).to(torch.float32)
        steps_run = live.sum(dim=0)
        returns = (rew * live).sum(dim=0)
        speeds = obs[:, :, :n_veh].to(torch.float64).mean(dim=2) * v_max    # mean over vehicles, per step and run
        mean_speed = (speeds * live.to(torch.float64)).sum(dim=0) / steps_run.to(torch.float64)
        torch.cuda.synchronize(vec.device)
        elapsed = time.time() - began
        returns, mean_speed = returns.cpu().numpy(), mean_speed.cpu().numpy()
        for run in range(int(num_runs)):
            log.close_run(run, float(returns[run]), float(mean_speed[run]), 0.0, {})      # closed networks: no outflow
        log.step_rates = [float(steps_run.sum().item()) / elapsed]
        vec.close()
        self.env.terminate()
        info = log.summary(elapsed)
        info["steps_run"] = steps_run.cpu().numpy().astype(int).tolist()
        return info

    # ------------------------------------------------------------------------------------------ stepwise
    def _run_stepwise(self, num_runs, policy, convert_to_csv):
        env, horizon = self.env, self.env.env_params.horizon
        log = _RunLog(self.custom_callables.keys())
        began = time.time()
        for run in range(int(num_runs)):
            obs = env.reset()
            episode_return, speed_samples = 0, []
            probes = {key: [] for key in self.custom_callables}
            step = 0
            while step < horizon:
                action = None if policy is None else policy(obs)
                tick = time.time()
                obs, reward, done, _ = env.step(action)
                log.step_rates.append(1.0 / max(time.time() - tick, 1e-9))
                speed_samples.append(np.mean(env.k.vehicle.get_speed(env.k.vehicle.get_ids())))
                episode_return += reward
                for key, probe in self.custom_callables.items():
                    probes[key].append(probe(env))
                step += 1
                if done:
                    break
            log.close_run(run, episode_return, np.mean(speed_samples), env.k.vehicle.get_outflow_rate(int(500)),
                          probes)
        info = log.summary(time.time() - began)
        emission = env.write_emission() if convert_to_csv else None
        env.terminate()
        if emission is not None:
            info["emission_csv"] = emission
        return info
