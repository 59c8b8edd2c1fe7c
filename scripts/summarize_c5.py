#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs written by scripts/profile_c5.sh into profiles/<tag>_c5_*:

    python scripts/summarize_c5.py gpurun_out/prof_c5_r01 r01 [c5|c4]

<tag>_c5_kernel_stats.csv (the --kernel-trace --stats table), <tag>_c5_pmc_summary.json (SQ counters of the longest
k_steps_open launch, per wave and per simulation sub-step) and <tag>_c5_bench.json (the leg's JSON)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
leg = sys.argv[3] if len(sys.argv) > 3 else "c5"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")


def newest(pattern):
    return sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)[-1]


shutil.copy(newest("trace/*/*_kernel_stats.csv"), os.path.join(out, "%s_%s_kernel_stats.csv" % (tag, leg)))
bench = json.loads(open(os.path.join(src, "bench_trace.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(out, "%s_%s_bench.json" % (tag, leg)), "w"), indent=1)
if leg == "c3" and "c3_figure_eight" in bench:   # scripts/bench_c3.py prints both heads: the counters are the AccelEnv launch's
    bench = dict(bench["c3_figure_eight"], po_head=bench.get("c3_figure_eight_po"))
slots = bench.get("slots", 64)                 # 64 slots per replica: one wave each; k_steps_wide: 2 or 4 waves
waves = bench["replicas"] * (1 if slots <= 64 else (2 if slots <= 128 else 4))
if "k_drop_queue" in str(bench.get("kernel", "")):
    waves = bench["replicas"] * 4                 # one wave per entry lane, whatever the slot count
substeps = bench.get("env_steps", bench.get("steps", 0)) * bench.get("sims_per_step", 1)
pattern = "k_steps_wide" if slots > 64 else "k_steps_open"
_names = [r["Name"] for r in csv.DictReader(open(newest("trace/*/*_kernel_stats.csv")))]
for _q in ("k_merge_queue", "k_drop_queue"):          # the queue-order kernels (flowsim_queue.h), when the leg ran on them
    if any(_q in n for n in _names):
        pattern = _q
if leg == "c3":                                # 14 vehicles -> 16 lanes per replica, 4 replicas per wave
    waves, pattern = bench["replicas"] // 4, ("fs::k_rollout_loop" if any("k_rollout_loop" in r["Name"] for r in csv.DictReader(open(newest("trace/*/*_kernel_stats.csv")))) else "fs::k_steps<")
    substeps = bench.get("steps_per_launch", 1500)
counters = {}
for d in ("pmc_sq", "pmc_sq2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(newest(d + "/*/*_counter_collection.csv"))):
        if pattern in r["Kernel_Name"]:
            agg[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for name, by_dispatch in agg.items():
        total = max(by_dispatch.values())      # the timed 600-step launch
        counters[name] = {"per_launch": total, "per_wave_per_substep": total / waves / substeps}
rows = list(csv.DictReader(open(newest("trace/*/*_kernel_stats.csv"))))
k = max(rows, key=lambda r: float(r["TotalDurationNs"]))
json.dump({"kernel": k["Name"], "waves": waves, "substeps_per_launch": substeps,
           "longest_launch_ns_kernel_trace": float(k["MaxNs"]), "counters": counters,
           "note": "SQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count in units of 4 clock cycles"},
          open(os.path.join(out, "%s_%s_pmc_summary.json" % (tag, leg)), "w"), indent=1)
print(json.dumps(counters, indent=1))
