#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs written by scripts/profile_gpu.sh into the summaries kept under profiles/.

    python scripts/summarize_profile.py gpurun_out/prof_r02_mixed r02 mixed

Writes profiles/<tag>_kernel_stats_<prec>.csv (the --kernel-trace --stats table), profiles/<tag>_pmc_summary_<prec>.json
(per-launch counters of the dominant kernel, HBM bytes corrected as MI355X_MICROARCH.md prescribes:
FETCH_SIZE x2 on gfx950 for wide coalesced reads, WRITE_SIZE as is, both reported in KiB; SQ counters per wave and
step; effective clock from GRBM_GUI_ACTIVE) and profiles/<tag>_bench_under_profiler_<prec>.json (the bench line
printed during the trace pass)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
prec = sys.argv[3] if len(sys.argv) > 3 else "f32"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)

def newest(*parts):
    """gpurun merges every call's files into the same directory: take the latest pass"""
    return sorted(glob.glob(os.path.join(src, *parts)), key=os.path.getmtime)[-1]


stats = newest("trace", "*", "*_kernel_stats.csv")
shutil.copy(stats, os.path.join(out, "%s_kernel_stats_%s.csv" % (tag, prec)))
rows = list(csv.DictReader(open(stats)))
dom = max(rows, key=lambda r: float(r["TotalDurationNs"]))
kernel = dom["Name"]

bench = json.loads(open(os.path.join(src, "bench_trace.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(out, "%s_bench_under_profiler_%s.json" % (tag, prec)), "w"), indent=1)


def per_launch(pass_dir):
    f = newest(pass_dir, "*", "*_counter_collection.csv")
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"] == kernel:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for k, v in agg.items()}


pmc = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_grbm"):
    try:
        pmc.update(per_launch(d))
    except IndexError:
        pass
# only the full 1500-step launches count (the profiled command also makes shorter ones)
full_ns = [float(r["AverageNs"]) for r in rows if r["Name"] == kernel]
steps = bench["roofline"]["steps_per_launch"]
R = bench["config"]["replicas_per_gpu"]
fetch_kib = pmc["FETCH_SIZE"]["mean"]
write_kib = pmc["WRITE_SIZE"]["mean"]
hbm_bytes = (2.0 * fetch_kib + write_kib) * 1024.0
summary = {
    "kernel": kernel, "replicas": R, "steps_per_launch": steps,
    "avg_launch_ns_kernel_trace": float(dom["AverageNs"]), "launches_kernel_trace": int(dom["Calls"]),
    "avg_launch_ms_hip_events_same_run": bench["roofline"]["avg_launch_ms"],
    "counters_per_launch": pmc,
    "fetch_bytes_per_launch_corrected": 2.0 * fetch_kib * 1024.0, "write_bytes_per_launch": write_kib * 1024.0,
    "hbm_bytes_per_launch": hbm_bytes, "algorithmic_bytes_per_launch": bench["roofline"]["bytes_per_launch"],
    "note": "FETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads); "
            "4-byte-per-lane stores are an uncalibrated width in the guide, WRITE_SIZE taken at face value",
}
waves = pmc.get("SQ_WAVES", {}).get("mean")
if waves:
    per = waves * steps
    summary["per_wave_step"] = {k: pmc[k]["mean"] / per for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU") if k in pmc}
    for k in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
        if k in pmc:
            summary["per_wave_step"][k + "_x4_cycles"] = 4.0 * pmc[k]["mean"] / per
if "GRBM_GUI_ACTIVE" in pmc:
    summary["effective_clock_GHz"] = pmc["GRBM_GUI_ACTIVE"]["mean"] / 8.0 / float(dom["AverageNs"])
json.dump(summary, open(os.path.join(out, "%s_pmc_summary_%s.json" % (tag, prec)), "w"), indent=1)
print(json.dumps(summary, indent=1))
