"""Summarise a tools/prof.sh output directory: per-kernel time and PMC means (per launch)."""
import csv
import glob
import collections
import sys
import json

import os

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "sweep"
out = {}


def newest(pattern):
    """One file per pass: gpurun merges a run's files into whatever an earlier run of the same tag left here (rocprofv3 names its files
    after the process id), so only the newest file of each pass directory belongs to the last run."""
    by_dir = {}
    for f in glob.glob(pattern):
        k = os.path.dirname(os.path.dirname(f))
        if k not in by_dir or os.path.getmtime(f) > os.path.getmtime(by_dir[k]):
            by_dir[k] = f
    return sorted(by_dir.values())


for f in newest(f"{d}/trace/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if pat in r["Name"]:
            out["kernel"] = r["Name"]
            out["calls"] = int(r["Calls"])
            out["avg_us"] = float(r["AverageNs"]) / 1e3
# the same kernel over the launches bench.py's own launch_us comes from: the run of the dominant kernel that closes the
# bench (run it with --no-standing-start so that nothing follows); the spin-up and warm-up launches before it are slower
# while the device clocks ramp up, so the all-launch average above sits a few percent higher
try:
    line = [l for l in open(f"{d}/trace.log") if l.startswith("{")][-1]
    roof = json.loads(line)["roofline"]
    # the launches bench.py's launch_us is averaged over: its closing run of the dominant kernel
    timed = int(roof.get("launches") or max(6, min(60, int(json.loads(line)["steps"]) // max(1, round(roof["steps_per_launch"])))))
    for f in newest(f"{d}/trace/*/*_kernel_trace.csv"):
        durs = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                for r in csv.DictReader(open(f)) if r["Kernel_Name"] == out.get("kernel")]
        durs.sort()
        # single steps that finish a fused loop are other kernels, so count this kernel's share of the timed launches
        last = [x[1] for x in durs[-timed:]]
        if last:
            out["timed_launches"] = len(last)
            out["timed_avg_us"] = sum(last) / len(last) / 1e3
except Exception as e:  # older runs without the line
    out["timed_avg_us"] = None
for f in newest(f"{d}/pmc_*/*/*_counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            out["vgpr"] = int(r["VGPR_Count"]); out["sgpr"] = int(r["SGPR_Count"]); out["lds"] = int(r["LDS_Block_Size"])
            out["scratch"] = int(r["Scratch_Size"])
    for k, v in agg.items():
        out[k] = sum(v) / len(v)
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    # FETCH_SIZE/WRITE_SIZE are in KiB-like units of 1024 B; on gfx950 FETCH_SIZE tallies the 128-B
    # requests of wide (16 B/lane) streaming reads at 64 B, so the read side is doubled.
    out["read_bytes_corrected"] = out["FETCH_SIZE"] * 1024 * 2
    out["write_bytes"] = out["WRITE_SIZE"] * 1024
    out["hbm_bytes_per_launch"] = out["read_bytes_corrected"] + out["write_bytes"]
    if "avg_us" in out:
        out["fabric_TBps"] = out["hbm_bytes_per_launch"] / (out.get("timed_avg_us") or out["avg_us"]) * 1e-6
print(json.dumps(out, indent=1))
