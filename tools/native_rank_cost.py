"""What one rank of the library's multi-GPU loop (fdtdx_run over RCCL) costs against the undecomposed loop, measured on ONE GPU:
a middle slab of a 3-rank chain (512 owned planes, ghost planes on both sides) whose two neighbours are itself, so every step
runs the real sequence -- boundary-plane sweeps, ncclSend/ncclRecv of the ghost planes on the exchange stream, the interior
sweep beside them -- with the transfers going through local memory instead of an xGMI link.  The fields are wrong (a slab is
not its own neighbour); the time is that of a rank whose links are never the bottleneck.
usage: python tools/native_rank_cost.py [steps=300] [reserve_cus=0] [boundary_stream=1]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fdtd_amd import Stencil, make_program

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
reserve = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bstream = int(sys.argv[3]) if len(sys.argv) > 3 else 1
N = 512
cs = torch.cuda.current_stream().cuda_stream
for kind, R, ghost in (("star", 1, 3), ("jacobi7", 1, 3), ("star", 2, 2), ("star", 4, 4), ("star", 7, 7), ("star", 8, 8), ("spc13", 1, 2)):
    prog = make_program(kind, radius=R)
    res = {}
    for chain in (1, 3):
        st = Stencil(N * chain, N, N, ghost, rank=chain // 2, nprocs=chain, device=0)
        st.mallocCube("p0"), st.mallocCube("p1")
        if kind == "jacobi7":
            st.initJacobi("p0"), st.initJacobi("p1")
        else:
            st.initMt("p0", "p1")
        if chain > 1:
            st.commInit(Stencil.commUniqueId(), comm_rank=0, comm_size=1, peer_lo=0, peer_hi=0)
            st.setOption("reserve_cus", reserve)
            st.setOption("boundary_stream", bstream)
        Stencil.run([st], "p0", "p1", prog, steps, [cs])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        Stencil.run([st], "p0", "p1", prog, steps, [cs])
        e1.record()
        torch.cuda.synchronize()
        res[chain] = e0.elapsed_time(e1) * 1e3 / steps
        st.close()
    print(f"{kind} r={R} ghost={ghost}: undecomposed {res[1]:7.1f} us/step, middle rank over RCCL {res[3]:7.1f} us/step, "
          f"ratio {res[1] / res[3]:.3f}", flush=True)
