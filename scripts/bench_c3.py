"""The C3 figure-eight leg of bench.py on its own (for profiling k_steps): prints one JSON object.

    python scripts/bench_c3.py [po]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    import torch
    import bench
    print(json.dumps(bench.c3_leg(torch.device("cuda", 0), po=(len(sys.argv) > 1 and sys.argv[1] == "po"))))
