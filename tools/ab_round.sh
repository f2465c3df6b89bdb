#!/usr/bin/env bash
# tools/ab_round.sh <suffix> <radius> ...: parity of the small cases, then bench.py A/B (tools/ab_libs.sh), for build/m<radius>_<suffix> (GPU box)
set -uo pipefail
SFX="$1"; shift
for M in "$@"; do
  L="m${M}_${SFX}"
  FDTDX_LIB=$PWD/build/$L/libfdtdx.so timeout -k 10 200 python -u -m pytest tests/test_gpu_parity_small.py -x -q -m gpu --timeout 120 --timeout-method thread 2>&1 | tail -1 | sed "s/^/$L parity: /"
  echo "== fdtd$M"; bash tools/ab_libs.sh fdtd$M 100 $L TREE $L
done
