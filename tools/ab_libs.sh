#!/usr/bin/env bash
# tools/ab_libs.sh <variant> <steps> <dir> ...: bench.py's figure for one variant with each of build/<dir>/libfdtdx.so in turn, on one
# box, the tree's own library first and last (GPU box; experiments with kernel build options)
set -uo pipefail
V="$1"; S="$2"; shift 2
for D in TREE "$@" TREE; do
  if [ "$D" = "TREE" ]; then LIB="$PWD/fdtd_amd/libfdtdx.so"; else LIB="$PWD/build/$D/libfdtdx.so"; fi
  echo -n "$D: "
  FDTDX_LIB="$LIB" timeout -k 10 200 python3 bench.py --variant "$V" --steps "$S" --no-cpu --no-standing-start ${AB_EXTRA:-} 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["launch_us"], d["ms_per_step"])' || echo failed
done
