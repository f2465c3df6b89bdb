#!/usr/bin/env bash
# tools/rehearse_shared_gpu.sh: the N > 1 paths of bench.py with two ranks on the box's one GPU (a rehearsal of the code paths, not a
# measurement): the IPC and local-slab tests, one variant over the copy transport, the probe's fallback chain forced by
# FDTDX_BENCH_FAIL_PROBE, and the whole default bench.  Results under gpurun_out/ (GPU box).
set -o pipefail
export MASTER_ADDR=127.0.0.1
R="python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541"
timeout -k 10 300 python -u -m pytest tests/test_gpu_ipc_slabs.py tests/test_gpu_native_slabs.py -x -q -m gpu --timeout 200 --timeout-method thread > gpurun_out/r03_ipc2_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03_ipc2_tests.log
timeout -k 10 200 $R bench.py --gpus 2 --rehearse-shared-gpu --variant fdtd4 --steps 50 --no-cpu > gpurun_out/r03_reh_ipc_fdtd4.json 2> gpurun_out/r03_reh_ipc_fdtd4.err; echo "fdtd4 rc=$?"; tail -c 700 gpurun_out/r03_reh_ipc_fdtd4.json; tail -3 gpurun_out/r03_reh_ipc_fdtd4.err
FDTDX_BENCH_FAIL_PROBE=ipc timeout -k 10 200 $R bench.py --gpus 2 --rehearse-shared-gpu --variant fdtd2 --steps 20 --no-cpu > gpurun_out/r03_reh_fallback.json 2> gpurun_out/r03_reh_fallback.err; echo "fallback rc=$?"; tail -c 500 gpurun_out/r03_reh_fallback.json; grep -v "amdgpu.ids\|Warning\|warn" gpurun_out/r03_reh_fallback.err | tail -5
timeout -k 10 500 $R bench.py --gpus 2 --rehearse-shared-gpu --steps 20 --warmup 5 > gpurun_out/r03_reh_config3.json 2> gpurun_out/r03_reh_config3.err; echo "config3 rc=$?"; python - <<PY
import json
d=json.loads(open("gpurun_out/r03_reh_config3.json").read().strip().splitlines()[-1])
print(d["value"], d["config"].get("exchange"), {k:v["gcells"] for k,v in d["variants"].items()})
PY
# (three or four ranks on the one GPU: export GPU_MAX_HW_QUEUES=2 first -- with four hardware queues per process the driver starts
# time-slicing the GPU between the processes and every exchange waits for a scheduling quantum, 2.7 / 5.7 ms)
