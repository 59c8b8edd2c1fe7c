"""C3 (figure eight, 13 noisy IDM + 1 RL) rollout rate, both heads: bench.py's own leg, three runs each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda", 0)
for po in (False, True):
    best = max(bench.c3_leg(dev, po=po)["value"] for _ in range(3))
    print("C3 %s: %.3f G env-steps/s" % ("WaveAttenuationPOEnv" if po else "AccelEnv", best / 1e9))
res = max((bench.c3_leg(dev, precision="mixed") for _ in range(3)), key=lambda d: d["value"])
print("C3 AccelEnv, FS_MIXED: %.3f G env-steps/s on %s" % (res["value"] / 1e9, res["kernel"]))
