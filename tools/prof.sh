#!/usr/bin/env bash
# tools/prof.sh <tag> <bench args...>: rocprofv3 kernel trace + PMC passes of one bench.py run.
# Writes CSVs under gpurun_out/prof_<tag>/; run on the GPU box (gpurun).  Each PMC pass is its
# own run (counters never share a run with a trace).
set -uo pipefail
TAG="$1"; shift
OUT="gpurun_out/prof_${TAG}"
rm -rf "$OUT"  # never mix passes of different builds
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS=("$@")
run() {  # name, rocprof options...
    local name="$1"; shift
    timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py "${ARGS[@]}" --no-cpu > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
}
run trace --kernel-trace --stats || exit 1
run pmc_fetch --pmc FETCH_SIZE || exit 1
run pmc_write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 1
run pmc_sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU || exit 1
run pmc_sq2 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU GRBM_GUI_ACTIVE || exit 1
find "$OUT" -name "*.csv" | head -30
