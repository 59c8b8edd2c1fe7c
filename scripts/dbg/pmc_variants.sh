#!/bin/bash
# cycles vs wall time of the pair-kernel harness variants (is a slower variant more cycles, or a lower clock?)
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
OUT=gpurun_out/pmc_variants
rm -rf $OUT; mkdir -p $OUT
for v in "$@"; do
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/$v -- scripts/ubench/$v > $OUT/$v.log 2> $OUT/$v.err
done
python3 - "$OUT" "$@" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for v in sys.argv[2:]:
    cc = glob.glob("%s/%s/*/*_counter_collection.csv" % (out, v))
    kt = glob.glob("%s/%s/*/*_kernel_trace.csv" % (out, v))
    if not cc or not kt:
        print(v, "no output"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc[0])):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(kt[0])):
        dur[r["Kernel_Name"][:40]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k in agg:
        a = {c: sum(x) / len(x) for c, x in agg[k].items()}
        d = sum(dur[k]) / len(dur[k])
        print("%-12s %-40s %.1f us  clock %.2f GHz  per wave-step: cyc %.0f valu_act %.0f wait %.0f insts %.1f" % (
            v, k, d / 1e3, a["GRBM_GUI_ACTIVE"] / d, a["SQ_WAVE_CYCLES"] * 4 / a.get("SQ_WAVES", 1024) / 1500 if False else a["SQ_WAVE_CYCLES"] * 4 / 1024 / 1500,
            a["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / 1500, a["SQ_WAIT_ANY"] * 4 / 1024 / 1500, a["SQ_INSTS_VALU"] / 1024 / 1500))
PY
