#!/usr/bin/env bash
# tools/golden_scan.sh <steps> <radius> ...: the host-order sweeps (bench.py --order golden) of the named radii, shape 0 (the library's
# choice) and shape 2 (32 x 32 tile on 512 lanes, two rows per lane) with the tree's library and with build/gold_adj_m<r>,
# build/gold_adjlate_m<r> (adjacent rows; + late publish).  GPU box; experiment.
S="$1"; shift
for m in "$@"; do for rep in 1 2; do for L in TREE gold_adj_m$m gold_adjlate_m$m; do for sh in 0 2; do
  if [ "$L" = "TREE" ]; then LIB="$PWD/fdtd_amd/libfdtdx.so"; else LIB="$PWD/build/$L/libfdtdx.so"; fi
  echo -n "fdtd$m golden $L shape $sh: "
  FDTDX_LIB="$LIB" timeout -k 10 200 python3 bench.py --variant fdtd$m --order golden --steps "$S" --shape $sh --no-cpu --no-standing-start 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["launch_us"], d["config"]["kernel"][:80])' || echo failed
done; done; done; done
