"""Host cost of one ghost-plane exchange as SlabRunner issues it (torch.distributed batch_isend_irecv on the
"nccl" = RCCL backend), measured on ONE GPU by sending to oneself: the transfers are local copies, the host
path (P2POp objects, group start/end, work handles, stream hand-over) is the real one."""
import os
import sys
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
planes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = planes * 512 * 544
src = [torch.zeros(n, device="cuda") for _ in range(2)]
dst = [torch.zeros(n, device="cuda") for _ in range(2)]


CACHED = []


def exchange():
    ops = CACHED
    if not ops or "--fresh" in sys.argv:
        ops = []
        for s, d in zip(src, dst):
            ops.append(dist.P2POp(dist.isend, s, 0))
            ops.append(dist.P2POp(dist.irecv, d, 0))
        CACHED[:] = ops
    reqs = dist.batch_isend_irecv(ops)
    for r in reqs:
        r.wait()


for _ in range(20):
    exchange()
torch.cuda.synchronize()
for label, sync_each in (("host enqueue only", False), ("with device sync each call", True)):
    t0 = time.perf_counter()
    for _ in range(200):
        exchange()
        if sync_each:
            torch.cuda.synchronize()
    if not sync_each:
        host = (time.perf_counter() - t0) / 200
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / 200
    print(f"{label}: {total * 1e6:.1f} us per exchange" + ("" if sync_each else f" (host returns after {host * 1e6:.1f} us)"))
dist.destroy_process_group()
