#!/usr/bin/env bash
# tools/run_ablations_bounded.sh <radius> <scale> <mask> ...: per-step time of the ablated libraries on fields that stay finite
# (coefficients scaled by <scale> = 1 / their sum), 400 steps in blocks of 20: the last blocks are the settled speed (GPU box)
set -uo pipefail
R="$1"; SC="$2"; shift 2
for MASK in "$@"; do
  if [ "$MASK" = "0" ]; then LIB="$PWD/fdtd_amd/libfdtdx.so"; else LIB="$PWD/build/abl_m${R}_${MASK}/libfdtdx.so"; fi
  echo -n "mask ${MASK}: "
  FDTDX_LIB="$LIB" timeout -k 10 120 python3 tools/steptrace.py --radius "$R" --total 400 --every 20 --scale "$SC" 2>/dev/null | tail -1 || echo failed
done
