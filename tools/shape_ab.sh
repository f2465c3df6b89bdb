#!/usr/bin/env bash
# tools/shape_ab.sh <variant> <shape index> [bench args]: bench.py with the library's own tile shape and with shape <index> (tools/tune.py
# lists them), alternating on one box (GPU box; tuning aid)
V="$1"; SH="$2"; shift 2
for rep in 1 2 3; do for sh in -1 "$SH"; do echo -n "$V $* shape $sh: "; timeout -k 10 200 python3 bench.py --variant "$V" --shape $sh --no-cpu --no-standing-start "$@" 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["launch_us"])'; done; done
