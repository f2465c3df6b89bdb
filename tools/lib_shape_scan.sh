#!/usr/bin/env bash
# tools/lib_shape_scan.sh <variant> <steps> <dir> <shape index> ...: bench.py's figure for each named tile shape with the tree's library
# and with build/<dir>/libfdtdx.so, alternating, three rounds on one box (GPU box; kernel experiments on the drivers' own fields)
V="$1"; S="$2"; D="$3"; shift 3
for rep in 1 2 3; do for sh in "$@"; do for L in TREE "$D"; do
  if [ "$L" = "TREE" ]; then LIB="$PWD/fdtd_amd/libfdtdx.so"; else LIB="$PWD/build/$L/libfdtdx.so"; fi
  echo -n "$V steps=$S $L shape $sh: "
  FDTDX_LIB="$LIB" timeout -k 10 200 python3 bench.py --variant "$V" --steps "$S" --shape $sh --no-cpu --no-standing-start 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["launch_us"], d["config"]["kernel"][:80])' || echo failed
done; done; done
