import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from bench import c2_spec
from flow_amd.sim import FlowSim
for R, K in ((70, 64), (1024, 64), (4096, 64), (4096, 1500)):
    spec = c2_spec(R, seed=1000)
    sim = FlowSim(spec, "f32")
    dev = torch.device("cuda", 0)
    obs = torch.full((K, R, 44), float("nan"), device=dev)
    rew = torch.full((K, R), float("nan"), device=dev)
    done = torch.full((K, R), 7, dtype=torch.uint8, device=dev)
    sim.reset()
    sim.rollout_dev(K, obs, rew, done)
    sim.sync()
    n = rew.isnan().float().mean(dim=1).cpu().numpy()
    print("R", R, "K", K, "nan frac per step (first 20):", np.round(n[:20], 3), "last:", np.round(n[-14:], 3),
          "done7:", float((done == 7).float().mean()), "obs nan:", float(obs.isnan().float().mean()))
    bad = np.nonzero(n)[0]
    print("   steps with nan rewards:", bad[:40], len(bad))
    sim.close()
