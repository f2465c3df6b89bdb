"""One configuration of tools/native_rank_cost.py (middle rank over RCCL loopback) for a kernel trace.
usage: python tools/native_rank_trace.py kind radius ghost [steps=60] [boundary_stream=1] [reserve_cus=0]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fdtd_amd import Stencil, make_program

kind, R, ghost = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 60
N = 512
cs = torch.cuda.current_stream().cuda_stream
st = Stencil(N * 3, N, N, ghost, rank=1, nprocs=3, device=0)
st.mallocCube("p0"), st.mallocCube("p1")
st.initMt("p0", "p1")
st.commInit(Stencil.commUniqueId(), comm_rank=0, comm_size=1, peer_lo=0, peer_hi=0)
prog = make_program(kind, radius=R)
st.setOption("boundary_stream", int(sys.argv[5]) if len(sys.argv) > 5 else 1)
st.setOption("reserve_cus", int(sys.argv[6]) if len(sys.argv) > 6 else 0)
Stencil.run([st], "p0", "p1", prog, 300, [cs])
torch.cuda.synchronize()
Stencil.run([st], "p0", "p1", prog, steps, [cs])
torch.cuda.synchronize()
st.close()
