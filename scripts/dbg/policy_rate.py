"""Closed-loop rate of the fused policy + step kernel on the reference's RL ring (21 noisy IDM + 1 RL, WaveAttenuationPOEnv,
ring length per replica, fcnet_hiddens [32, 32, 32]): K = 500-step fragments, 4096 replicas; next to the HIP-graph form
(VecFlowEnv.capture with the same network as a torch module)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import train_vec
from flow_amd.envs import VecFlowEnv
from flow_amd.utils.device_policy import DevicePolicy

R, K = 4096, 500
dev = torch.device("cuda", 0)
for precision, noise in (("f32", 0.2), ("mixed", 0.0)):
    fp = train_vec.ring_flow_params(3000)
    fp["sim"].precision = precision
    fp["env"].additional_params["ring_length"] = [220, 270]
    if not noise:
        for t in fp["veh"].type_parameters.values():
            if "noise" in t["acceleration_controller"][1]:
                t["acceleration_controller"][1]["noise"] = 0.0
    vec = VecFlowEnv(fp, num_replicas=R, device=0)
    hidden = [torch.nn.Linear(3, 32), torch.nn.Linear(32, 32), torch.nn.Linear(32, 32)]
    head = torch.nn.Linear(32, 2)
    for l in hidden + [head]:
        l.to(dev)
    pol = DevicePolicy(hidden, head, seed=1)
    vec.reset()
    out = vec.policy_rollout(pol, K, reset_done=True)
    torch.cuda.synchronize()
    ms = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); vec.policy_rollout(pol, K, reset_done=True, out=out); e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    t = sum(ms) / len(ms) * 1e-3
    print("%-6s fused policy + step: %.2f G env-steps/s (%.3f ms per %d-step fragment), kernel %s, mean reward %.3f" %
          (precision, R * K / t / 1e9, t * 1e3, K, vec.sim.last_kernel, float(out[3].mean())))
    vec.close()
