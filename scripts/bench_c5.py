"""The C5 merge leg of bench.py on its own (for profiling k_steps_open): prints one JSON object."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    import torch
    import bench
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    print(json.dumps(bench.c5_leg(torch.device("cuda", 0), R=R)))
