#!/usr/bin/env bash
# tools/clocks.sh <bench args...>: samples clocks and power (rocm-smi) while bench.py loops, to tell a
# kernel held back by the power cap (shader clock drops under load) from one that is not.
set -uo pipefail
python3 bench.py "$@" --no-cpu > gpurun_out/clocks_bench.json 2> gpurun_out/clocks_bench.err &
PID=$!
sleep 12   # import + warm-up
for i in 1 2 3 4 5 6; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr -s ' ' | tr '\n' ';'
    echo
    sleep 1
done
wait $PID
cat gpurun_out/clocks_bench.json | cut -c1-120
