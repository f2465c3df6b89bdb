"""Times every compiled tile shape of the composite step of main.cu at full size (tuning aid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from fdtd_amd import Stencil, make_composite, FdtdxError

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
with Stencil(n, n, n, 4) as st:
    for name in ("p0", "p1", "vel", "sum"):
        st.mallocCube(name)
    st.initMt("p0", "p1")
    st.initMt("vel", "sum")
    addr = np.random.default_rng(0).integers(0, n ** 3, 10000).astype(np.uint64)
    st.mallocAddr("addr", len(addr))
    st.transferAddrToGPU("addr", addr)
    comp = make_composite()
    for fused in (True, False):
        for zc in (0, 2, 4, 8):
            v = 0
            while True:
                st.setOption("variant", v)
                st.setOption("zchunks", zc)

                def step(s0, s1):
                    if fused:
                        st.compositeStep(s0, s1, "vel", "sum", "addr", comp, 1.0, n // 2, n // 3, n // 4, 0.9)
                    else:
                        st.propagateComposite(s0, s1, "vel", comp)
                try:
                    step("p0", "p1")
                except FdtdxError:
                    break
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for t in range(10):
                    step("p0", "p1") if t % 2 == 0 else step("p1", "p0")
                e1.record()
                torch.cuda.synchronize()
                print(f"fused={fused} v={v} zc={zc}: {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us/step  [{st.lastLaunchInfo()[:90]}]", flush=True)
                v += 1
