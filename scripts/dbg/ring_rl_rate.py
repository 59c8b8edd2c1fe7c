"""Open-loop rollout rate of the reference's RL ring experiment (21 noisy IDM + 1 RL, WaveAttenuationPOEnv) with an action tape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "examples"))
import torch
import train_vec
from flow_amd.envs import VecFlowEnv
R, K = 4096, 1500
vec = VecFlowEnv(train_vec.ring_flow_params(1500), num_replicas=R, device=0)
dev = torch.device("cuda", 0)
tape = (torch.rand((K, R, 1), device=dev) * 2 - 1)
out = (torch.empty((K, R, vec.obs_dim), device=dev), torch.empty((K, R), device=dev), torch.empty((K, R), dtype=torch.uint8, device=dev))
vec.reset(); vec.rollout(K, tape, out=out); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    vec.reset(); vec.rollout(K, tape, out=out)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("ring 21 IDM(noise) + 1 RL, PO head: %.2f G env-steps/s, kernel %s" % (3 * R * K / dt / 1e9, vec.sim.last_kernel))
