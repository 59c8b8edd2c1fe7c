"""C4 (lane drop, k_steps_wide) in float32 and float64: bench.py's leg, best of two."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device("cuda", 0)
for prec in ("f32", "f64"):
    r = max((bench.c4_leg(dev, precision=prec) for _ in range(2)), key=lambda d: d["value"])
    print("C4 %s: %.2f M env-steps/s" % (prec, r["value"] / 1e6))
