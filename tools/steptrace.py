"""Time per step along one run (events every `--every` steps): shows whether a sweep's speed depends on how
long the device has been busy or on the values in the cubes (the star drivers' fields grow by the coefficient
sum each step and overflow to inf after ~100-700 steps depending on the radius)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fdtd_amd import Stencil, make_program

ap = argparse.ArgumentParser()
ap.add_argument("--kind", default="star")
ap.add_argument("--radius", type=int, default=8)
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--total", type=int, default=400)
ap.add_argument("--every", type=int, default=10)
ap.add_argument("--temporal", type=int, default=0)
ap.add_argument("--scale", type=float, default=1.0, help="multiplies every coefficient (1/sum keeps values bounded)")
a = ap.parse_args()
n, M = a.n, a.radius
stream = torch.cuda.current_stream().cuda_stream
with Stencil(n, n, n, M) as st:
    st.mallocCube("p0"), st.mallocCube("p1")
    if a.kind == "jacobi7":
        st.initJacobi("p0"), st.initJacobi("p1")
    else:
        st.initMt("p0", "p1")
    st.setOption("temporal", a.temporal)
    prog = make_program(a.kind, radius=M)
    for i in range(13):
        prog.coef[i] *= a.scale
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.total // a.every + 1)]
    ev[0].record()
    for k in range(a.total // a.every):
        st.sweep("p0", "p1", prog, a.every, stream)
        ev[k + 1].record()
    torch.cuda.synchronize()
    us = [ev[k].elapsed_time(ev[k + 1]) * 1e3 / a.every for k in range(len(ev) - 1)]
    print(" ".join(f"{u:.0f}" for u in us))
