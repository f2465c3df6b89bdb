"""Times the one-launch Yee step at 512^3 for several z-chunk counts (tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fdtd_amd import Stencil
n = 512
names = ("ex", "ey", "ez", "hx", "hy", "hz")
nb = tuple(x + "_b" for x in names)
cs = torch.cuda.current_stream().cuda_stream
with Stencil(n, n, n, 1) as st:
    bufs = {}
    for i, name in enumerate(names + nb):
        bufs[name] = torch.rand(int(st.geometry.cube_elems), dtype=torch.float32, device="cuda") * 1e-3
        st.attachCube(name, bufs[name].data_ptr(), bufs[name].numel() * 4)
    sets = (names, nb)
    def loop(k):
        for t in range(k):
            st.yee3dStepFused(sets[t % 2], sets[(t + 1) % 2], 0.31, 0.27, stream=cs)
    loop(80)
    for zc in [int(v) for v in sys.argv[1:]] or [0]:
        st.setOption("zchunks", zc)
        loop(10)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); loop(40); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 40
        print(f"zchunks={zc}: {us:.0f} us per step, {(n - 2) ** 3 / us * 1e-3:.1f} Gcell/s", flush=True)
