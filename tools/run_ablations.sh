#!/usr/bin/env bash
# tools/run_ablations.sh <radius> <variant index> <mask> ...: time the libraries tools/build_ablations.sh built (GPU box)
set -uo pipefail
R="$1"; V="$2"; shift 2
for MASK in "$@"; do
  if [ "$MASK" = "0" ]; then LIB="$PWD/fdtd_amd/libfdtdx.so"; else LIB="$PWD/build/abl_m${R}_${MASK}/libfdtdx.so"; fi
  echo -n "mask ${MASK}: "
  FDTDX_LIB="$LIB" timeout -k 10 120 python3 tools/tune.py --radii "$R" --variants "$V" --steps 40 2>&1 | grep "us/step" | sed 's/\[kind.*//' || echo failed
done
