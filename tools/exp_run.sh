#!/usr/bin/env bash
# tools/exp_run.sh <tag> <lib or ""> <tune.py args...>: time a tune.py run and collect FETCH_SIZE / WRITE_SIZE for it
# (separate rocprofv3 passes), with an alternative build of the library if one is named.  GPU box only.
set -uo pipefail
TAG="$1"; LIB="$2"; shift 2
OUT="gpurun_out/exp_${TAG}"
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
[ -n "$LIB" ] && export FDTDX_LIB="$PWD/$LIB"
timeout -k 10 200 python3 tools/tune.py "$@" > "$OUT/time.log" 2>&1 || { echo "timing failed"; tail -5 "$OUT/time.log"; exit 1; }
cat "$OUT/time.log"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d "$OUT/$c" -- python3 tools/tune.py "$@" --steps 6 > "$OUT/$c.log" 2>&1 || { echo "pass $c failed"; tail -5 "$OUT/$c.log"; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
for c, mul in (("FETCH_SIZE", 2048), ("WRITE_SIZE", 1024)):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{d}/{c}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        tail = v[len(v) // 2:]
        print(f"{c} x{mul}: {sum(tail) / len(tail) * mul / 1e6:9.1f} MB per launch over {len(tail)} launches  {k[:90]}")
PY
