#!/usr/bin/env bash
# tools/build_ablations.sh <radius> <mask> [<mask> ...]: copies of the library whose radius-<radius> star kernels are built with
# FDTDX_ABLATE=<mask> (star_kernel.h: timing experiments, results wrong), as build/abl_m<radius>_<mask>/libfdtdx.so.  The other
# objects are the tree's own (run make first).
set -euo pipefail
cd "$(dirname "$0")/../fdtd_amd/csrc"
R="$1"; shift
EXTRA_SCHED=""
case "$R" in 6|7|8) EXTRA_SCHED="-mllvm -amdgpu-sched-strategy=iterative-ilp";; esac
for MASK in "$@"; do
  (
    D="../../build/abl_m${R}_${MASK}"; mkdir -p "$D"
    hipcc -std=c++17 -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wall -Wno-unused-function -Wno-uninitialized -pthread $EXTRA_SCHED ${ABL_EXTRA:-} -DFDTDX_ABLATE=${MASK} -c star_m${R}.hip -o "$D/star_m${R}.o"
    OBJS=$(ls *.o | grep -v "^star_m${R}.o$")
    hipcc --offload-arch=gfx950 -shared -fPIC -pthread $OBJS "$D/star_m${R}.o" -ldl -o "$D/libfdtdx.so"
    echo "built $D/libfdtdx.so"
  ) &
  while [ "$(jobs -r | wc -l)" -ge 4 ]; do sleep 1; done
done
wait
