"""What cutting a step into the boundary-plane launch and the interior launch costs by itself, without any exchange: a middle slab
(512 owned planes, ghost planes on both sides) swept as one launch per step against the two launches of the multi-rank step
(the 2 x M planes next to the neighbours first, then the rest), same stream.  GPU box; usage: python tools/split_step_cost.py [steps=200]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fdtd_amd import Stencil, make_program

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N = 512
cs = torch.cuda.current_stream().cuda_stream


def timed(f):
    for _ in range(3):
        f(30)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    f(steps)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / steps


for R in (3, 4, 5, 6, 7, 8):
    prog = make_program("star", radius=R)
    for i in range(13):
        prog.coef[i] /= (0.01 + 0.27 * R)
    with Stencil(3 * N, N, N, R, rank=1, nprocs=3, device=0) as st:
        st.mallocCube("p0"), st.mallocCube("p1")
        st.initMt("p0", "p1")
        g = st.geometry
        lo, hi = int(g.ghost_lo), int(g.ghost_lo) + N  # owned planes

        def whole(k):
            a, b = "p0", "p1"
            for _ in range(k):
                st.propagatePlanes(a, b, prog, lo, hi, cs)
                a, b = b, a

        def split(k):
            a, b = "p0", "p1"
            for _ in range(k):
                st.propagateTwoRanges(a, b, prog, lo, lo + R, hi - R, hi, cs)
                info_b = st.lastLaunchInfo()
                st.propagatePlanes(a, b, prog, lo + R, hi - R, cs)
                a, b = b, a
            return info_b

        tw = timed(whole)
        ts = timed(split)
        st.propagateTwoRanges("p0", "p1", prog, lo, lo + R, hi - R, hi, cs)
        ib = st.lastLaunchInfo().split("tile=")[1].split(" launches")[0]
        st.propagatePlanes("p0", "p1", prog, lo + R, hi - R, cs)
        ii = st.lastLaunchInfo().split("tile=")[1].split(" launches")[0]
        tb = timed(lambda k: [st.propagateTwoRanges("p0", "p1", prog, lo, lo + R, hi - R, hi, cs) for _ in range(k)])
        print(f"radius {R}: one launch {tw:6.1f} us, boundary + interior {ts:6.1f} us (ratio {tw / ts:.3f}); boundary launch alone {tb:5.1f} us [{ib}] interior [{ii}]", flush=True)
