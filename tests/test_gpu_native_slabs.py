"""The native multi-slab path: fdtdx_run / fdtdx_step_begin / fdtdx_hop_begin with the ghost-plane exchange done inside
the library (fdtdx_comm.hip), the counterpart of the drivers' loop fdtd1.cu:68-80 with stencil.h:639-682.

  * local transport: ranks 0..n-1 of an n-way z split live in this one process on the one GPU, each with a stream of its
    own; the library moves the planes fdtdx_halo_region names with device copies on its exchange streams.  Results must
    equal the oracle's undecomposed loop bit for bit, for single steps and for fused hops (three steps per launch for
    radius-1 programs with ghost depth 3, two for the 13-point program with ghost depth 2).
  * RCCL transport: one rank that is its own neighbour on both sides (a communicator of one): the real ncclSend/ncclRecv
    group on the exchange stream, checked by where the planes end up.
"""
import numpy as np
import pytest
import torch

from fdtd_amd import Stencil, make_program

pytestmark = pytest.mark.gpu


class _CAI:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def make_slabs(Z, Y, X, ghost, n, fields):
    slabs = []
    for r in range(n):
        st = Stencil(Z, Y, X, ghost, rank=r, nprocs=n, device=0)
        g = st.geometry
        lo, cnt = int(g.first_global), int(g.local_planes)
        for name, arr in fields.items():
            st.mallocCube(name)
            st.uploadLocal(name, arr[lo:lo + cnt])
        slabs.append(st)
    return slabs


@pytest.mark.parametrize("own_streams", [False, True], ids=["default-stream", "stream-per-slab"])
@pytest.mark.parametrize("kind,R,ghost,n,steps,fused", [
    ("star", 1, 1, 2, 5, None), ("star", 4, 4, 2, 3, None), ("star", 8, 8, 3, 3, None), ("star", 2, 2, 3, 4, None),
    ("star", 1, 3, 2, 11, "fused3"), ("jacobi7", 1, 3, 3, 8, "fused3"), ("spc13", 1, 2, 2, 10, "fused2"), ("spc13", 1, 2, 3, 13, "fused2"),
    ("spc13", 1, 1, 2, 6, None)])
def test_local_slabs_equal_undecomposed_loop(oracle_cpu, kind, R, ghost, n, steps, fused, own_streams):
    Z, Y, X = 20 * n + 2 * ghost, 41, 200
    if kind == "jacobi7":
        a = oracle_cpu.init_jacobi(Z, Y, X)
        b = a.copy()
    else:
        a, b = oracle_cpu.init_mt(Z, Y, X)
    prog = make_program(kind, radius=R)
    slabs = make_slabs(Z, Y, X, ghost, n, {"p0": a, "p1": b})
    Stencil.commInitLocal(slabs)
    streams = [torch.cuda.Stream() for _ in slabs] if own_streams else None
    Stencil.run(slabs, "p0", "p1", prog, steps, [s.cuda_stream for s in streams] if streams else None)
    torch.cuda.synchronize()
    info = slabs[0].lastLaunchInfo()
    got0 = np.concatenate([s.downloadOwned("p0") for s in slabs], axis=0)
    got1 = np.concatenate([s.downloadOwned("p1") for s in slabs], axis=0)
    for s in slabs:
        s.close()
    oracle_cpu.run(kind, a, b, steps, M=R, order=oracle_cpu.ORDER_CHAIN)
    assert (fused in info) if fused else ("fused" not in info), info
    for name, got, ref in (("p0", got0, a), ("p1", got1, b)):
        bad = np.argwhere(got.view(np.uint32) != ref.view(np.uint32))
        assert len(bad) == 0, f"{name}: {len(bad)} cells differ, first {bad[:3].tolist()} ({info})"


@pytest.mark.parametrize("kind,R,ghost,steps", [("star", 4, 4, 4), ("star", 1, 3, 8), ("spc13", 1, 2, 10)])
def test_boundary_stream_option(oracle_cpu, kind, R, ghost, steps):
    """The same loop with the boundary-plane sweep on the library's own stream, beside the interior sweep."""
    n = 3
    Z, Y, X = 20 * n + 2 * ghost, 41, 200
    a, b = oracle_cpu.init_mt(Z, Y, X)
    prog = make_program(kind, radius=R)
    slabs = make_slabs(Z, Y, X, ghost, n, {"p0": a, "p1": b})
    for s in slabs:
        s.setOption("boundary_stream", 1)
    Stencil.commInitLocal(slabs)
    Stencil.run(slabs, "p0", "p1", prog, steps, None)
    torch.cuda.synchronize()
    got0 = np.concatenate([s.downloadOwned("p0") for s in slabs], axis=0)
    got1 = np.concatenate([s.downloadOwned("p1") for s in slabs], axis=0)
    for s in slabs:
        s.close()
    oracle_cpu.run(kind, a, b, steps, M=R, order=oracle_cpu.ORDER_CHAIN)
    assert np.array_equal(got0.view(np.uint32), a.view(np.uint32)) and np.array_equal(got1.view(np.uint32), b.view(np.uint32))


def test_local_step_by_step_with_marked_boundary(oracle_cpu):
    """The drivers' own order of calls (fdtd1.cu:70-78): boundary sweeps, then the whole interior queued, then the exchange --
    which still only waits for the boundary sweeps, marked with commBoundaryDone."""
    Z, Y, X, M, steps = 48, 33, 130, 2, 5
    a, b = oracle_cpu.init_mt(Z, Y, X)
    prog = make_program("star", radius=M)
    slabs = make_slabs(Z, Y, X, M, 2, {"p0": a, "p1": b})
    Stencil.commInitLocal(slabs)
    out, inp = "p0", "p1"
    for _ in range(steps):
        for s in slabs:
            g = s.geometry
            lo = int(g.ghost_lo) or M
            hi = int(g.local_planes) - (int(g.ghost_hi) or M)
            ilo, ihi = lo, hi
            if g.ghost_lo:
                ilo = lo + M
                s.propagatePlanes(out, inp, prog, lo, ilo)
            if g.ghost_hi:
                ihi = hi - M
                s.propagatePlanes(out, inp, prog, ihi, hi)
            s.commBoundaryDone()
            s.propagatePlanes(out, inp, prog, ilo, ihi)
            s.commExchangeStart(out)
        for s in slabs:
            s.commExchangeWait()
        out, inp = inp, out
    torch.cuda.synchronize()
    got0 = np.concatenate([s.downloadOwned("p0") for s in slabs], axis=0)
    for s in slabs:
        s.close()
    oracle_cpu.run("star", a, b, steps, M=M, order=oracle_cpu.ORDER_CHAIN)
    assert np.array_equal(got0.view(np.uint32), a.view(np.uint32))


def test_exchange_phases_are_checked():
    slabs = [Stencil(40, 20, 64, 2, rank=r, nprocs=2, device=0) for r in range(2)]
    for s in slabs:
        s.mallocCube("p0")
    from fdtd_amd import FdtdxError
    with pytest.raises(FdtdxError):  # no communicator yet
        slabs[0].commExchangeStart("p0")
    Stencil.commInitLocal(slabs)
    slabs[0].commExchangeStart("p0")
    with pytest.raises(FdtdxError):  # slab 1 has not started its half
        slabs[0].commExchangeWait()
    slabs[1].commExchangeStart("p0")
    for s in slabs:
        s.commExchangeWait()
    torch.cuda.synchronize()
    for s in slabs:
        s.close()


def test_rccl_send_recv_on_the_exchange_stream():
    """A middle slab (neighbours on both sides) whose two neighbours are itself: ncclSend/ncclRecv in one group on the
    exchange stream, ordered behind the caller's stream.  The planes sent down come back into the low ghost planes, the
    planes sent up into the high ones."""
    Z, Y, X, M = 60, 24, 100, 3
    with Stencil(Z, Y, X, M, rank=1, nprocs=3, device=0) as st:
        g = st.geometry
        L, P = int(g.local_planes), int(g.plane)
        st.mallocCube("p0")
        cube = torch.as_tensor(_CAI(st.getCube("p0"), L * P), device="cuda").view(L, P)
        cube.copy_(torch.arange(L, device="cuda", dtype=torch.float32)[:, None] + torch.rand(L, P, device="cuda") * 0.5)
        before = cube.clone()
        st.commInit(Stencil.commUniqueId(), comm_rank=0, comm_size=1, peer_lo=0, peer_hi=0)
        for _ in range(2):  # twice: the second exchange reuses streams and events
            st.commCubeHalo("p0", torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(cube[:M], before[M:2 * M])              # low ghosts <- first M owned planes
        assert torch.equal(cube[L - M:], before[L - 2 * M:L - M])  # high ghosts <- last M owned planes
        assert torch.equal(cube[M:L - M], before[M:L - M])         # owned planes untouched


@pytest.mark.parametrize("n", [2, 3])
def test_gather_cube_over_local_slabs(n):
    """transferCubeToCPU's gather (stencil.h:610-637) inside the library: one call on any slab of the process leaves the
    whole grid on the host."""
    Z, Y, X, M = 50, 21, 70, 2
    rng = np.random.default_rng(11)
    a = rng.random((Z, Y, X), dtype=np.float32)
    slabs = make_slabs(Z, Y, X, M, n, {"p0": a})
    Stencil.commInitLocal(slabs)
    for st in (slabs[0], slabs[-1]):
        got = np.zeros_like(a)
        st.gatherCube(got, "p0")
        assert np.array_equal(got, a)
    for s in slabs:
        s.close()


def test_gather_cube_over_rccl_communicator_of_one():
    """The RCCL form of the gather on a communicator of one rank: the slab's planes go through ncclBroadcast into the
    staging buffer and are unpadded from there (with more ranks every slab takes that route in turn)."""
    Z, Y, X, M = 40, 19, 130, 1
    rng = np.random.default_rng(12)
    a = rng.random((Z, Y, X), dtype=np.float32)
    with Stencil(Z, Y, X, M) as st:
        st.mallocCube("p0")
        st.transferCubeToGPU("p0", a)
        st.commInit(Stencil.commUniqueId())
        got = np.zeros_like(a)
        st.gatherCube(got, "p0")
    assert np.array_equal(got, a)


@pytest.mark.parametrize("n,steps,own_streams", [(2, 10, False), (3, 9, True), (2, 7, False), (3, 4, False)])
def test_radius2_fuses_on_slabs_through_a_work_cube(oracle_cpu, n, steps, own_streams):
    """fdtdx_run_work: the radius-2 star (fdtd2.cu) on slabs with ghost depth 4 runs two steps per launch, hopping p1 -> work ->
    p1 with one exchange of four planes per hop, and ends bit-equal to the oracle's step-by-step loop (every residue of the step
    count: the loop closes with one to four single steps, the first of them out of the work cube where the hops end there)."""
    ghost = 4
    Z, Y, X = 24 * n + 2 * ghost, 41, 200
    a, b = oracle_cpu.init_mt(Z, Y, X)
    prog = make_program("star", radius=2)
    slabs = make_slabs(Z, Y, X, ghost, n, {"p0": a, "p1": b, "work": np.zeros_like(a)})
    Stencil.commInitLocal(slabs)
    streams = [torch.cuda.Stream() for _ in slabs] if own_streams else None
    Stencil.run(slabs, "p0", "p1", prog, steps, [s.cuda_stream for s in streams] if streams else None, work="work")
    torch.cuda.synchronize()
    info = slabs[0].lastLaunchInfo()
    got0 = np.concatenate([s.downloadOwned("p0") for s in slabs], axis=0)
    got1 = np.concatenate([s.downloadOwned("p1") for s in slabs], axis=0)
    for s in slabs:
        s.close()
    oracle_cpu.run("star", a, b, steps, M=2, order=oracle_cpu.ORDER_CHAIN)
    assert ("fused2" in info) == (steps >= 4 and ((steps // 2 - 1) if steps % 2 == 0 else ((steps - 1) // 2) & ~1) > 0), info
    assert np.array_equal(got0.view(np.uint32), a.view(np.uint32)), info
    assert np.array_equal(got1.view(np.uint32), b.view(np.uint32)), info
