"""Multi-lane ring leg (k_steps_ml): 4096 replicas x 21 IDM vehicles on a 3-lane ring, LaneChangeAccelEnv head with
2 RL vehicles and random [acc, dir] actions; prints one JSON object.

    python scripts/bench_ml.py [replicas] [steps]
"""
import json
import os
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))

if __name__ == "__main__":
    import numpy as np
    import torch
    from helpers import multilane_spec
    from flow_amd.sim import FlowSim
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    spec = multilane_spec(R=R, N=21, lanes=3, horizon=10 ** 9, n_rl=2, seed=1, lane_change_duration=5)
    sim = FlowSim(spec, "f32")
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(0)
    acts = torch.rand((K, R, 4), device=dev, generator=gen) * 2 - 1
    acts[:, :, 1::2] = torch.round(acts[:, :, 1::2])
    obs = torch.empty((K, R, sim.obs_dim), dtype=torch.float32, device=dev)
    rew = torch.empty((K, R), dtype=torch.float32, device=dev)
    done = torch.empty((K, R), dtype=torch.uint8, device=dev)
    sim.reset()
    sim.rollout_dev(50, obs[:50], rew[:50], done[:50], actions=acts[:50])
    sim.sync()
    t0 = time.perf_counter()
    sim.rollout_dev(K, obs, rew, done, actions=acts)
    sim.sync()
    dt = time.perf_counter() - t0
    print(json.dumps({"value": R * K / dt, "unit": "env-steps/s", "replicas": R, "steps": K,
                      "workload": "3-lane ring, 21 IDM vehicles (2 RL with lane-change commands), LaneChangeAccelEnv; k_steps_ml"}))
