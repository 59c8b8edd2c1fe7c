#!/bin/bash
# block-size sweep of k_rollout_idm (how many independent waves share a CU)
for b in 64 128 256 512 768 1024; do
  FLOWSIM_ROLLOUT_BLOCK=$b timeout -k 10 200 python bench.py --no-extras --steps 15000 --warmup 3000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($b, d['value'], d['roofline']['avg_launch_ms'])"
done
