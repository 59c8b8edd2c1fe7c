"""bench.py's c3_closed_loop leg on its own: prints one JSON object."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if __name__ == "__main__":
    import torch
    import bench
    print(json.dumps(bench.c3_closed_loop_leg(torch.device("cuda", 0))))
