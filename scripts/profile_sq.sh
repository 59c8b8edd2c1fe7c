#!/bin/bash
# SQ + GRBM counters of the headline rollout kernel (one --pmc pass each; run through gpurun).
#   scripts/profile_sq.sh TAG [bench args]
set -e
TAG=${1:-r02}
shift || true
OUT=gpurun_out/prof_$TAG
ARGS="--steps 6000 --warmup 1500 --no-extras $*"
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS > $OUT/bench_sq.json 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_grbm -- python3 bench.py $ARGS > $OUT/bench_grbm.json 2> $OUT/grbm.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py $ARGS > $OUT/bench_sq2.json 2> $OUT/sq2.err || true
python3 - $OUT <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
for d in ("pmc_sq", "pmc_grbm", "pmc_sq2"):
    fs = sorted(glob.glob(os.path.join(out, d, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    if not fs: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[-1])):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "rollout" not in k: continue
        print(d, k, {c: (len(x), sum(x) / len(x)) for c, x in v.items()})
    ks = sorted(glob.glob(os.path.join(out, d, "*", "*_kernel_trace.csv")), key=os.path.getmtime)
    if ks:
        durs = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(ks[-1])) if "rollout" in r["Kernel_Name"]]
        if durs: print(d, "kernel-trace avg ns", sum(durs) / len(durs), len(durs))
PY
