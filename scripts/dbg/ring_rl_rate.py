"""Open-loop rollout rate of the reference's RL ring experiment (21 noisy IDM + 1 RL, WaveAttenuationPOEnv, ring length
per replica) with an action tape: kernel time of the 1500-step launch (HIP events), f32 with noise / f32 quiet / mixed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "examples"))
import torch
import train_vec
from flow_amd.envs import VecFlowEnv

R, K = 4096, 1500
dev = torch.device("cuda", 0)
CASES = (("f32 noise 0.2", "f32", 0.2, "hw"), ("f32 noise 0.2 exact", "f32", 0.2, "exact"), ("f32 quiet", "f32", 0.0, "hw"),
         ("mixed noise 0.2", "mixed", 0.2, "hw"), ("mixed quiet", "mixed", 0.0, "hw"))
if len(sys.argv) > 1:                                  # e.g. "exact": only the cases whose label holds the word
    CASES = tuple(c for c in CASES if sys.argv[1] in c[0] + " ")
for label, precision, noise, math in CASES:
    fp = train_vec.ring_flow_params(1500)
    fp["sim"].precision = precision
    fp["sim"].noise_math = math
    fp["env"].additional_params["ring_length"] = [220, 270]        # singleagent_ring.py:58-62: a length per episode
    if not noise:
        for t in fp["veh"].type_parameters.values():
            ac = t["acceleration_controller"]
            if "noise" in ac[1]:
                ac[1]["noise"] = 0.0
    vec = VecFlowEnv(fp, num_replicas=R, device=0)
    tape = (torch.rand((K, R, 1), device=dev) * 2 - 1)
    out = (torch.empty((K, R, vec.obs_dim), device=dev), torch.empty((K, R), device=dev),
           torch.empty((K, R), dtype=torch.uint8, device=dev))
    vec.reset(); vec.rollout(K, tape, out=out); torch.cuda.synchronize()
    ms = []
    for _ in range(4):
        vec.reset()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); vec.rollout(K, tape, out=out); e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    t = sum(ms) / len(ms) * 1e-3
    print("%-20s| %.2f G env-steps/s  (%.3f ms per %d-step launch), kernel %s" % (label, R * K / t / 1e9, t * 1e3, K, vec.sim.last_kernel))
    vec.close()
