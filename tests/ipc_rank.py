"""One rank of a multi-process run of the library's loop over the IPC transport (helper of test_gpu_ipc_slabs.py; every rank
uses cuda:0 on a one-GPU box, which is enough for processes to map one another's cubes and copy into them).

usage: ipc_rank.py <rank> <nprocs> <dir> <kind> <radius> <ghost> <steps>
reads <dir>/a.npy, b.npy (the whole initial grid), writes <dir>/out_<rank>_p0.npy, _p1.npy (the planes this rank owns)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from fdtd_amd import Stencil, make_program


def main():
    rank, nprocs, d, kind, radius, ghost, steps = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
    a, b = np.load(os.path.join(d, "a.npy")), np.load(os.path.join(d, "b.npy"))
    Z, Y, X = a.shape
    st = Stencil(Z, Y, X, ghost, rank=rank, nprocs=nprocs, device=0)
    g = st.geometry
    lo, cnt = int(g.first_global), int(g.local_planes)
    names = ["p0", "p1"] + (["work"] if (kind == "star" and radius == 2) else [])
    for name, arr in zip(names, (a, b, a)):
        st.mallocCube(name)
        st.uploadLocal(name, np.ascontiguousarray(arr[lo:lo + cnt]))
    desc = st.commIpcExport(names)
    tmp = os.path.join(d, f"desc_{rank}.tmp")
    with open(tmp, "wb") as f:
        f.write(desc)
    os.rename(tmp, os.path.join(d, f"desc_{rank}.bin"))

    def peer(r):
        if r < 0 or r >= nprocs:
            return None
        path = os.path.join(d, f"desc_{r}.bin")
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > 120:
                raise SystemExit(f"rank {rank}: no descriptor from rank {r}")
            time.sleep(0.01)
        return open(path, "rb").read()

    st.commIpcConnect(peer(rank - 1), peer(rank + 1))
    st.barrier()
    Stencil.run([st], "p0", "p1", make_program(kind, radius=radius), steps, None, work="work" if "work" in names else None)
    st.sync()
    info = st.lastLaunchInfo()
    np.save(os.path.join(d, f"out_{rank}_p0.npy"), st.downloadOwned("p0"))
    np.save(os.path.join(d, f"out_{rank}_p1.npy"), st.downloadOwned("p1"))
    with open(os.path.join(d, f"info_{rank}.txt"), "w") as f:
        f.write(info)
    st.barrier()  # nobody unmaps a cube a neighbour may still be copying into
    st.close()


if __name__ == "__main__":
    main()
