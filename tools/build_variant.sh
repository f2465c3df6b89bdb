#!/usr/bin/env bash
# tools/build_variant.sh <name> <unit> <flags...>: a copy of the library whose translation unit fdtd_amd/csrc/<unit>.hip is built with
# extra compiler flags (kernel experiments), as build/<name>/libfdtdx.so; the other objects are the tree's own (run make first).
# tools/ab_libs.sh times such copies against the tree's library on one box.
set -euo pipefail
cd "$(dirname "$0")/../fdtd_amd/csrc"
NAME="$1"; UNIT="$2"; shift 2
D="../../build/$NAME"; mkdir -p "$D"
SCHED=""
# SCHED_STRATEGY=<name> overrides the unit's usual scheduler (none: the compiler's default)
case "${SCHED_STRATEGY:-}$UNIT" in tb_r1_star|tb_r2_star|tb_r2_golden|tb_r3_star|spc13|star_m6|star_m7|star_m8) SCHED="-mllvm -amdgpu-sched-strategy=iterative-ilp";; esac
if [ -n "${SCHED_STRATEGY:-}" ] && [ "$SCHED_STRATEGY" != "none" ]; then SCHED="-mllvm -amdgpu-sched-strategy=$SCHED_STRATEGY"; fi
hipcc -std=c++17 -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wall -Wno-unused-function -Wno-uninitialized -pthread -I../../include $SCHED "$@" -c "$UNIT.hip" -o "$D/$UNIT.o" -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "VGPRs Spill: [1-9]" | sort | uniq -c || true
hipcc --offload-arch=gfx950 -shared -fPIC -pthread $(ls *.o | grep -v "^$UNIT.o$") "$D/$UNIT.o" -ldl -o "$D/libfdtdx.so"
echo "built build/$NAME/libfdtdx.so"
