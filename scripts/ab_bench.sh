#!/bin/bash
# A/B timings of the rollout kernels in ONE gpurun call (boxes differ by up to ~25 %: never compare across calls).
#   scripts/ab_bench.sh "VAR=val python bench.py args" ...   each argument is one command line
for cmd in "$@"; do
  out=$(bash -c "$cmd" 2>/dev/null | tail -1)
  python3 - "$cmd" "$out" <<'PY'
import json, sys
cmd, out = sys.argv[1], sys.argv[2]
try:
    j = json.loads(out)
    print("%-90s %7.3f G  %.4f ms/launch  frac %.3f" % (cmd[-90:], j["value"] / 1e9, j["roofline"]["avg_launch_ms"], j["roofline"]["frac"]))
except Exception as e:
    print(cmd, "FAILED", out[:200])
PY
done
