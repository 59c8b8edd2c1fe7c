"""The C5 merge / C4 bottleneck leg of bench.py on its own (for profiling k_steps_open): prints one JSON object.

    python scripts/bench_c5.py [replicas] [c5|c4]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    import torch
    import bench
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    leg = sys.argv[2] if len(sys.argv) > 2 else "c5"
    fn = bench.c4_leg if leg == "c4" else bench.c5_leg
    print(json.dumps(fn(torch.device("cuda", 0), R=R)))
