"""C3 legs only (figure eight, both heads): python scripts/bench_c3.py"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
out = {"c3_figure_eight": bench.c3_leg(dev), "c3_figure_eight_po": bench.c3_leg(dev, po=True)}
print(json.dumps(out))
