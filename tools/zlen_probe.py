"""Time per plane of the single-step sweep against the length of its z chunks (the planes a chunk reads only to fill its
queue are a fixed 2M per chunk).  usage: python tools/zlen_probe.py [radius=8]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fdtd_amd import Stencil, make_program

M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cs = torch.cuda.current_stream().cuda_stream
for Z in (272, 512, 768, 1024, 2048):
    with Stencil(Z, 512, 512, M) as st:
        st.mallocCube("p0"), st.mallocCube("p1")
        prog = make_program("star", radius=M)
        st.setOption("temporal", 0)
        st.sweep("p0", "p1", prog, 200, cs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        st.sweep("p0", "p1", prog, 200, cs)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 200
        print(f"M={M} Z={Z}: {us:7.1f} us/step, {us / (Z - 2 * M) * 1e3:6.1f} ns per plane  [{st.lastLaunchInfo().split(' grid=')[1][:26]}]", flush=True)
