"""Time per step of a single-step star sweep against the number of z chunks on a grid whose tiles do not fill the CUs in one round
(GPU box; tuning aid).  usage: python tools/zchunk_scan.py Z Y X radius [chunk counts ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fdtd_amd import Stencil, make_program

Z, Y, X, R = (int(v) for v in sys.argv[1:5])
counts = [int(v) for v in sys.argv[5:]] or [0, 1, 2, 3, 4, 5, 6, 8, 10, 12, 16]
cs = torch.cuda.current_stream().cuda_stream
with Stencil(Z, Y, X, R) as st:
    st.mallocCube("p0"), st.mallocCube("p1")
    st.initMt("p0", "p1")
    st.setOption("temporal", 0)
    prog = make_program("star", radius=R)
    for _ in range(3):
        st.sweep("p0", "p1", prog, 20, cs)
    for zc in counts:
        st.setOption("zchunks", zc)
        st.sweep("p0", "p1", prog, 4, cs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        st.sweep("p0", "p1", prog, 20, cs)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        cells = float(Z - 2 * R) * (Y - 2 * R) * (X - 2 * R)
        print(f"{Z}x{Y}x{X} r={R} zchunks={zc:2d}: {us:8.1f} us/step {cells / us * 1e-3:7.1f} Gcell/s  [{st.lastLaunchInfo().split('tile=')[1].split(' launches')[0]}]", flush=True)
