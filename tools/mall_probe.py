"""How fast do sweeps run when both cubes fit in the 256 MB Infinity Cache?  (exploration aid)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fdtd_amd import Stencil, make_program

stream = torch.cuda.current_stream().cuda_stream
for M, kind in ((1, "star"), (4, "star"), (8, "star"), (1, "spc13")):
    for Z in (24, 32, 48, 64, 96, 128, 192, 256, 512):
        if Z <= 2 * M + 2:
            continue
        with Stencil(Z, 512, 512, M) as st:
            st.mallocCube("p0"); st.mallocCube("p1"); st.initMt("p0", "p1")
            prog = make_program(kind, radius=M)
            st.sweep("p0", "p1", prog, 4, stream); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            steps = 40
            e0.record(); st.sweep("p0", "p1", prog, steps, stream); e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / steps
            cells = (Z - 2 * M) * (512 - 2 * M) ** 2
            ws = 2 * Z * 544 * 512 * 4 / 1e6
            print(f"{kind} M={M} Z={Z:4d} working set {ws:6.0f} MB: {ms*1e3:7.1f} us/step {cells/ms*1e-6:7.1f} Gcell/s  {st.lastLaunchInfo()[-40:]}", flush=True)
