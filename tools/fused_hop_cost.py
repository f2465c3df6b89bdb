"""Duration of one fused two-step launch of the 13-point program for the ways its shell sources can be wired: into a third
cube or in place, with the level-0 shell read from `in` itself or from the other cube.  usage: python tools/fused_hop_cost.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fdtd_amd import Stencil, make_program

n = 512
stream = torch.cuda.current_stream().cuda_stream
with Stencil(n, n, n, 1) as st:
    for c in ("p0", "p1", "c"):
        st.mallocCube(c)
    st.initMt("p0", "p1")
    st.initMt("c", "p1")
    prog = make_program("spc13")
    for _ in range(6):
        st.sweep("p0", "p1", prog, 60, stream)

    def t(label, f, cnt=40):
        f(), f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(cnt):
            f()
        e1.record()
        torch.cuda.synchronize()
        print(f"{label}: {e0.elapsed_time(e1) * 1e3 / cnt:.1f} us per launch", flush=True)
    lo, hi = 1, n - 1
    t("third cube, shell of level 0 in `in`   (out=c  in=p1 even=p1 odd=p0)", lambda: st.propagateFusedShells("c", "p1", "p1", "p0", prog, 2, lo, hi, stream))
    t("in place,   shell of level 0 in `in`   (out=p0 in=p1 even=p1 odd=p0)", lambda: st.propagateFusedShells("p0", "p1", "p1", "p0", prog, 2, lo, hi, stream))
    t("in place,   shell of level 0 elsewhere (out=p1 in=p0 even=p1 odd=p0)", lambda: st.propagateFusedShells("p1", "p0", "p1", "p0", prog, 2, lo, hi, stream))
