import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device("cuda", 0)
for prec in ("f32", "mixed", "f64"):
    r = max((bench.c5_leg(dev, precision=prec) for _ in range(2)), key=lambda d: d["value"])
    print("C5 %s: %.4f G sub-steps/s" % (prec, r["value"] / 1e9))
