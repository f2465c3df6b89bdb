"""Times fused kernel shapes and z-chunk counts at 512^3 (tuning aid).
usage: python tools/tune_tb.py [n] [plan-json {variant: [zchunks..]}] [kind star|spc13|jacobi7] [radius]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fdtd_amd import Stencil, make_program

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = 100
stream = torch.cuda.current_stream().cuda_stream
import json
plan = json.loads(sys.argv[2]) if len(sys.argv) > 2 else {str(v): [0, 2, 4, 8, 16] for v in range(5)}
KIND = sys.argv[3] if len(sys.argv) > 3 else "star"
RADIUS = int(sys.argv[4]) if len(sys.argv) > 4 else 1
for kind in (KIND,):
    with Stencil(n, n, n, RADIUS) as st:
        st.mallocCube("p0"); st.mallocCube("p1")
        if kind != "jacobi7":
            st.initMt("p0", "p1")
        else:
            st.initJacobi("p0"); st.initJacobi("p1")
        prog = make_program(kind, radius=RADIUS)
        for _ in range(6):
            st.sweep("p0", "p1", prog, 60, stream)  # let the device clock settle
        for v, zcs in sorted((int(k), z) for k, z in plan.items()):
            for zc in zcs:
                st.setOption("tb_variant", v); st.setOption("tb_zchunks", zc)
                st.sweep("p0", "p1", prog, 6, stream); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); st.sweep("p0", "p1", prog, steps, stream); e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / steps
                print(f"{kind} v={v} zc={zc}: {ms*1e3:7.1f} us/step {(n-2*RADIUS)**3/ms*1e-6:7.1f} Gcell/s [{st.lastLaunchInfo().split('|')[-1].strip()}]", flush=True)
