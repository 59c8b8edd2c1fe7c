#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs of scripts/profile_rl.sh into profiles/<tag>_rl_*: the kernel-trace --stats tables of the open-loop
(k_ring_pair) and the closed-loop (k_ring_policy) runs and the SQ counters of their longest launches per wave (4 replicas)
and step.     python scripts/summarize_rl.py gpurun_out/prof_rl_r03 r03"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")


def newest(pattern):
    return sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)[-1]


summary = {"note": "SQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count in units of 4 clock cycles; 4096 replicas = 1024 waves"}
for leg, trace, pmc, pattern, steps, txt in (("open_loop", "trace", "pmc_sq", "k_ring_pair", 1500, "rate_trace.txt"),
                                             ("closed_loop", "trace_pol", "pmc_sq_pol", "k_ring_policy", 500, "policy_trace.txt")):
    stats = newest(trace + "/*/*_kernel_stats.csv")
    shutil.copy(stats, os.path.join(out, "%s_rl_%s_kernel_stats.csv" % (tag, leg)))
    rows = [r for r in csv.DictReader(open(stats)) if pattern in r["Name"]]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    names = {}
    for r in csv.DictReader(open(newest(pmc + "/*/*_counter_collection.csv"))):
        if pattern in r["Kernel_Name"]:
            agg[r["Kernel_Name"]][(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    kernels = {}
    for kname, vals in agg.items():
        by_counter = collections.defaultdict(list)
        for (cname, _), v in vals.items():
            by_counter[cname].append(v)
        waves = 1024
        kernels[kname] = {c: max(v) / waves / steps for c, v in by_counter.items()}     # the full-length launches
    summary[leg] = {"kernel_trace": [{"kernel": r["Name"], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                      "max_ns": float(r["MaxNs"])} for r in rows],
                    "steps_per_launch": steps, "counters_per_wave_per_step": kernels,
                    "printed": open(os.path.join(src, txt)).read().strip().splitlines()[-3:]}
json.dump(summary, open(os.path.join(out, "%s_rl_pmc_summary.json" % tag), "w"), indent=1)
print(json.dumps(summary, indent=1)[:3000])
