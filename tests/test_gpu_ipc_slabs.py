"""The IPC transport of the multi-slab loop (fdtdx_comm_ipc_export / _connect): one PROCESS per slab, every process mapping its
neighbours' cubes and pushing its boundary planes into their ghost planes with device copies, the hosts pacing the exchange
through shared memory (the reference runs one MPI rank per slab and blocks its hosts in Sendrecv while the kernels run,
fdtd1.cu:70-78, stencil.h:649-656).  On the one-GPU box every process uses cuda:0; the processes, the mapped memory, the copies
and the pacing are real.  The gathered planes must equal the oracle's undecomposed loop bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("kind,R,ghost,n,steps,fused", [("star", 1, 1, 2, 5, None), ("star", 4, 4, 3, 4, None), ("star", 8, 8, 2, 3, None),
                                                        ("star", 1, 3, 2, 11, "fused3"), ("spc13", 1, 2, 3, 10, "fused2"),
                                                        ("star", 2, 4, 2, 9, "fused2")])
def test_ipc_ranks_equal_undecomposed_loop(oracle_cpu, tmp_path, kind, R, ghost, n, steps, fused):
    _run_ranks(oracle_cpu, tmp_path, kind, R, ghost, n, steps, fused, (24 * n + 2 * ghost, 41, 200))


# the planes of the bench's own grid (512 x 512 cells, 1.1 MB each; ghost blocks of 3.3 and 8.9 MB per face and exchange)
@pytest.mark.parametrize("kind,R,ghost,n,steps,fused", [("star", 8, 8, 2, 3, None), ("star", 1, 3, 2, 7, "fused3")])
def test_ipc_ranks_full_size_planes(oracle_cpu, tmp_path, kind, R, ghost, n, steps, fused):
    _run_ranks(oracle_cpu, tmp_path, kind, R, ghost, n, steps, fused, (192, 512, 512))


def _run_ranks(oracle_cpu, tmp_path, kind, R, ghost, n, steps, fused, dims):
    Z, Y, X = dims
    a, b = oracle_cpu.init_mt(Z, Y, X)
    np.save(tmp_path / "a.npy", a)
    np.save(tmp_path / "b.npy", b)
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "ipc_rank.py"), str(r), str(n), str(tmp_path), kind, str(R), str(ghost), str(steps)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(n)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=240)[0])
    finally:
        for p in procs:  # the children this test started, by their own handles
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    got0 = np.concatenate([np.load(tmp_path / f"out_{r}_p0.npy") for r in range(n)], axis=0)
    got1 = np.concatenate([np.load(tmp_path / f"out_{r}_p1.npy") for r in range(n)], axis=0)
    info = open(tmp_path / "info_0.txt").read()
    oracle_cpu.run(kind, a, b, steps, M=R, order=oracle_cpu.ORDER_CHAIN)
    assert (fused in info) if fused else ("fused" not in info), info
    assert np.array_equal(got0.view(np.uint32), a.view(np.uint32)), info
    assert np.array_equal(got1.view(np.uint32), b.view(np.uint32)), info
