"""C4 legs of bench.py side by side (best of three): queue kernel, lane changing on (k_steps_wide), float64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda", 0)
for name, kw in (("f32", {}), ("f32 lane_change_mode 1621", {"lane_change_mode": 1621}), ("f64", {"precision": "f64"}),
                 ("f32 1024 replicas", {"R": 1024})):
    r = max((bench.c4_leg(dev, **kw) for _ in range(3)), key=lambda d: d["value"])
    print("C4 %-28s %.2f M env-steps/s  (%s)" % (name, r["value"] / 1e6, r.get("kernel")))
