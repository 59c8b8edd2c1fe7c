import sys, json
sys.path.insert(0, '/root/repo')
import torch, bench
print(json.dumps(bench.ring_defaults_leg(torch.device("cuda", 0)), indent=1))
