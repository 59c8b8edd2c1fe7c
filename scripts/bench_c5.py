"""The C5 merge / C4 bottleneck leg of bench.py on its own (for profiling k_steps_open): prints one JSON object.

    python scripts/bench_c5.py [replicas] [c5|c4|c4lc] [slots (c4 only, default 256)]

c4lc: the lane-drop leg with lane changing on (flow/benchmarks/bottleneck1's mode 1621 -- BASELINE's wording of C4): k_steps_wide.
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
    if leg in ("c4", "c4lc"):
        slots = int(sys.argv[3]) if len(sys.argv) > 3 else 256
        print(json.dumps(bench.c4_leg(torch.device("cuda", 0), R=R, slots=slots, lane_change_mode=1621 if leg == "c4lc" else 0)))
    else:
        print(json.dumps(bench.c5_leg(torch.device("cuda", 0), R=R)))
