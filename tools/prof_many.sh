#!/usr/bin/env bash
# tools/prof_many.sh <variant>...: tools/prof.sh for each bench variant named (output under gpurun_out/prof_r_<variant>);
# "config3" runs the default bench command under the kernel trace only.  GPU box only; stops at the first failure.
set -uo pipefail
for v in "$@"; do
  if [ "$v" = config3 ]; then
    export TMPDIR=/tmp
    rm -rf gpurun_out/prof_r_config3; mkdir -p gpurun_out/prof_r_config3
    timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r_config3/trace -- python3 bench.py --no-cpu > gpurun_out/prof_r_config3/trace.log 2>&1 || { echo "config3 trace failed"; tail -5 gpurun_out/prof_r_config3/trace.log; exit 1; }
    tail -1 gpurun_out/prof_r_config3/trace.log | cut -c1-300
  else
    tools/prof.sh "r_$v" --variant "$v" --no-standing-start > "gpurun_out/prof_r_$v.log" 2>&1 || { echo "$v failed"; tail -5 "gpurun_out/prof_r_$v.log"; exit 1; }
    echo "$v done"
  fi
done
