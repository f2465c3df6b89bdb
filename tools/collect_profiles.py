"""Copies the rocprofv3 summaries of tools/prof.sh runs from gpurun_out/ into profiles/ (tracked)
and (re)builds profiles/traffic.json, which bench.py reads for roofline.traffic.

usage: python tools/collect_profiles.py r01 jacobi:prof_jacobi_tb:tb2 fdtd1:prof_fdtd1b:star_sweep ...
       (variant : directory under gpurun_out : kernel-name substring)
"""
import json
import os
import shutil
import subprocess
import sys
import glob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
traffic_path = os.path.join(ROOT, "profiles", "traffic.json")
traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
for spec in sys.argv[2:]:
    variant, d, pat = spec.split(":")
    src = os.path.join(ROOT, "gpurun_out", d)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_summary.py"), src, pat],
                         capture_output=True, text=True, check=True).stdout
    summ = json.loads(out)
    base = os.path.join(ROOT, "profiles", f"{rnd}_{variant}")
    json.dump(summ, open(base + "_summary.json", "w"), indent=1)
    for f in sorted(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime):  # the newest one wins
        shutil.copy(f, base + "_kernel_stats.csv")
    try:  # the shape bench.py reported for the profiled run: bench.py drops the figure when the kernel's shape has changed since
        line = [l for l in open(os.path.join(src, "trace.log")) if l.startswith("{")][-1]
        launch_info = json.loads(line)["config"]["kernel"].rsplit(" launches=", 1)[0]
    except Exception:
        launch_info = None
    traffic[f"{variant}_n512"] = {
        "kernel": summ.get("kernel"), "launch_info": launch_info, "avg_us": summ.get("avg_us"), "timed_avg_us": summ.get("timed_avg_us"),
        "hbm_bytes_per_launch": summ.get("hbm_bytes_per_launch"),
        "read_bytes_corrected": summ.get("read_bytes_corrected"), "write_bytes": summ.get("write_bytes"),
        "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, units of 1024 B; FETCH_SIZE doubled "
                "(gfx950 tallies the 128-B requests of 16-B-per-lane streaming reads at 64 B)",
        "round": rnd,
    }
    print(variant, "->", base + "_summary.json")
json.dump(traffic, open(traffic_path, "w"), indent=1)
