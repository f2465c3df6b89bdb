#!/usr/bin/env bash
# tools/tune_libs.sh <radius> <steps> <dir> ...: tools/tune.py (finite fields, every compiled shape of the radius) with the tree's library
# and with each of build/<dir>/libfdtdx.so, the tree's first and last (GPU box; kernel experiments built by tools/build_variant.sh)
set -uo pipefail
M="$1"; S="$2"; shift 2
for D in TREE "$@" TREE; do
  if [ "$D" = "TREE" ]; then LIB="$PWD/fdtd_amd/libfdtdx.so"; else LIB="$PWD/build/$D/libfdtdx.so"; fi
  echo "== radius $M $D"
  FDTDX_LIB="$LIB" timeout -k 10 200 python3 tools/tune.py --radii "$M" --steps "$S" 2>&1 | cut -c1-160 || exit 1
done
