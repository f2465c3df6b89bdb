"""Host cost of the multi-rank step sequence on one GPU (tuning aid, not a measurement of scaling).

Runs the loop of a middle rank of three (ghost planes on both sides, boundary-first launches) with
the exchange switched off, and prints per step: host time to enqueue the launches vs GPU time to
run them.  The multi-GPU loop keeps up with the GPU only while the first stays below the second.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fdtd_amd import make_program
from fdtd_amd.slab import HipSlabBackend, SlabRunner

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = 200
for name, kind, radius in (("jacobi", "jacobi7", 1), ("fdtd1", "star", 1), ("fdtd2", "star", 2), ("fdtd4", "star", 4),
                           ("fdtd8", "star", 8), ("fdtdspc", "spc13", 1)):
    ghost = 3 if radius == 1 and kind != "spc13" else radius
    be = HipSlabBackend(3 * n, n, n, ghost, 1, 3, 0)
    be.alloc("p0")
    be.alloc("p1")
    prog = make_program(kind, radius=radius)
    run = SlabRunner(be, ghost, prog, 1, 3, exchange="none", side_stream=os.environ.get("SIDE", "") == "1")
    run.run("p0", "p1", 10)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    run.run("p0", "p1", steps)
    e1.record()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{name}: host enqueue {th / steps * 1e6:7.1f} us/step, GPU {e0.elapsed_time(e1) / steps * 1e3:7.1f} us/step "
          f"[{be.st.lastLaunchInfo().rsplit('launches=', 1)[1]} launches]", flush=True)
    be.st.close()
    del be, run
    torch.cuda.empty_cache()
