"""C5 (merge, k_steps_open) and C4 (lane drop, k_steps_wide) rates: bench.py's own legs, best of three."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda", 0)
which = sys.argv[1:] or ["c5", "c4"]
if "c5" in which:
    r = max((bench.c5_leg(dev) for _ in range(3)), key=lambda d: d["value"])
    print("C5: %.4f G sub-steps/s" % (r["value"] / 1e9))
if "c4" in which:
    r = max((bench.c4_leg(dev) for _ in range(3)), key=lambda d: d["value"])
    print("C4: %.2f M env-steps/s" % (r["value"] / 1e6))
