#!/bin/bash
# VALU instructions per wave-step of k_ring_pair<PO> with the hardware and with the exact Box-Muller (run through gpurun)
set -e
OUT=gpurun_out/prof_exact
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
for m in "exact" "f32 noise 0.2 "; do
  tag=$(echo $m | tr ' .' '__')
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/$tag -- python3 scripts/dbg/ring_rl_rate.py "$m" > $OUT/$tag.txt 2> $OUT/$tag.err
  cat $OUT/$tag.txt
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/prof_exact/*/")):
    f = glob.glob(d + "*/*counter_collection.csv")
    if not f: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        if "k_ring_pair" in r["Kernel_Name"]:
            acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    best = max(acc.values(), key=lambda c: c.get("SQ_INSTS_VALU", 0))
    waves = best["SQ_WAVES"]
    print(d, {k: round(v / waves / 1500, 1) for k, v in best.items() if k != "SQ_WAVES"}, "waves", waves)
PY
