#!/bin/bash
# replicas-per-GPU sweep of the C2 rollout (how far the kernel is from the HBM roof once the chip is full)
for cfg in "4096 1500 30000" "16384 375 7500" "65536 100 3000" "262144 25 750"; do
  set -- $cfg
  timeout -k 10 280 python bench.py --no-extras --replicas $1 --fragment $2 --steps $3 --warmup $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print($1, '%.3g env-steps/s' % d['value'], '%.0f GB/s' % r['achieved'], 'frac %.3f' % r['frac'], '%.3f ms/launch' % r['avg_launch_ms'], r['steps_per_launch'])"
done
