"""What the library's own multi-slab loop (fdtdx_run: boundary planes first, ghost-plane exchange beside the interior sweep)
costs against the undecomposed loop, on ONE GPU: the same 512^3 grid as one slab and cut into n z-slabs that exchange their
ghost planes by device copies.  The slabs share the GPU, so this measures the launch sequence and the exchange machinery,
not xGMI.  usage: python tools/native_slabs_cost.py [n_slabs=2] [steps=300]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fdtd_amd import Stencil, make_program

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
N = 512
for kind, R, ghost in (("star", 1, 3), ("star", 2, 2), ("star", 4, 4), ("star", 8, 8), ("spc13", 1, 2)):
    prog = make_program(kind, radius=R)
    res = {}
    for parts in (1, n):
        slabs = [Stencil(N, N, N, ghost, rank=r, nprocs=parts, device=0) for r in range(parts)]
        for s in slabs:
            s.mallocCube("p0"), s.mallocCube("p1")
            s.initMt("p0", "p1")
        if parts > 1:
            Stencil.commInitLocal(slabs)
        Stencil.run(slabs, "p0", "p1", prog, steps)  # clocks
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        Stencil.run(slabs, "p0", "p1", prog, steps)
        e1.record()
        torch.cuda.synchronize()
        res[parts] = e0.elapsed_time(e1) * 1e3 / steps
        info = slabs[0].lastLaunchInfo().split("|")[-1].split(" grid=")[0].strip()
        for s in slabs:
            s.close()
    cells = float(N - 2 * R) ** 3
    print(f"{kind} r={R} ghost={ghost}: 1 slab {res[1]:7.1f} us/step ({cells / res[1] * 1e-3:6.1f} Gcell/s), {n} slabs {res[n]:7.1f} us/step "
          f"({cells / res[n] * 1e-3:6.1f} Gcell/s), ratio {res[1] / res[n]:.3f}  [{info}]", flush=True)
