"""Duration of the boundary-plane launches of a multi-rank step (the G planes next to a neighbour) against the number of
z chunks they are cut into.  usage: python tools/boundary_cost.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fdtd_amd import Stencil, make_program

N = 512
cs = torch.cuda.current_stream().cuda_stream


def timed(f, cnt=200):
    f(), f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(cnt):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / cnt


for kind, R, ghost, fused in (("star", 1, 3, 3), ("star", 2, 2, 0), ("star", 4, 4, 0), ("star", 8, 8, 0), ("spc13", 1, 2, 2)):
    with Stencil(3 * N, N, N, ghost, rank=1, nprocs=3, device=0) as st:
        st.mallocCube("p0"), st.mallocCube("p1")
        st.initMt("p0", "p1")
        prog = make_program(kind, radius=R)
        st.sweep if False else None
        lo = int(st.geometry.ghost_lo)
        for _ in range(300):
            st.propagatePlanes("p0", "p1", prog, lo, lo + N, cs)
        out = []
        for zc in (1, 2, 3, 4, 8):
            if zc > ghost:
                continue
            st.setOption("zchunks", zc), st.setOption("tb_zchunks", zc)
            if fused:
                t = timed(lambda: st.propagateFusedShells("p0", "p1", "p1", "p0", prog, fused, lo, lo + ghost, cs))
            else:
                t = timed(lambda: st.propagatePlanes("p0", "p1", prog, lo, lo + ghost, cs))
            out.append(f"zchunks={zc}: {t:6.1f} us")
        print(f"{kind} r={R} planes={ghost} fused={fused}: " + ", ".join(out), flush=True)
