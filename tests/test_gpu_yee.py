"""EXTENSION -- no reference oracle.  The staggered-grid E/H updates (3-D six-field Yee step, 2-D
ex/ey/hz) do not exist in the reference; they are defined by oracle/fdtd_oracle.c (oracle_yee3d_half,
oracle_yee2d_step) and the HIP kernels must reproduce that definition bit for bit in both orders.
Beyond that, a property the scheme itself offers: the discrete divergence of H is left unchanged by a
step (div curl = 0 on the staggered grid) up to rounding."""
import numpy as np
import pytest
import torch

from fdtd_amd import Stencil, ORDER_CHAIN, ORDER_GOLDEN
from fdtd_amd import lib

pytestmark = pytest.mark.gpu
ORD = {ORDER_CHAIN: 1, ORDER_GOLDEN: 0}
NAMES = ("ex", "ey", "ez", "hx", "hy", "hz")


def fields(shape, seed=3):
    rng = np.random.default_rng(seed)
    return [rng.random(shape, dtype=np.float32) for _ in range(6)]


def gpu_yee3d(shape, f, steps, order, ce=0.31, ch=0.27):
    Z, Y, X = shape
    with Stencil(Z, Y, X, 1) as st:
        for n, a in zip(NAMES, f):
            st.mallocCube(n)
            st.transferCubeToGPU(n, a)
        for _ in range(steps):
            st.yee3dStep(NAMES, ce, ch, order)
        st.sync()
        out = [np.empty(shape, np.float32) for _ in range(6)]
        for n, a in zip(NAMES, out):
            st.transferCubeToCPU(a, n)
    return out


def gpu_yee3d_fused(shape, f, steps, order, ce=0.31, ch=0.27):
    """The one-launch step, ping-ponging between two sets of six cubes (both start as the initial fields,
    so both carry the shell)."""
    Z, Y, X = shape
    sets = (NAMES, tuple(n + "_b" for n in NAMES))
    with Stencil(Z, Y, X, 1) as st:
        for names in sets:
            for n, a in zip(names, f):
                st.mallocCube(n)
                st.transferCubeToGPU(n, a)
        for t in range(steps):
            st.yee3dStepFused(sets[t % 2], sets[(t + 1) % 2], ce, ch, order)
        st.sync()
        out = [np.empty(shape, np.float32) for _ in range(6)]
        for n, a in zip(sets[steps % 2], out):
            st.transferCubeToCPU(a, n)
    return out


@pytest.mark.parametrize("order", [ORDER_CHAIN, ORDER_GOLDEN])
@pytest.mark.parametrize("shape", [(3, 3, 3), (9, 14, 21), (20, 37, 260), (40, 9, 513), (70, 23, 1030)])
def test_yee3d_fused_step_equals_two_half_steps(oracle_cpu, shape, order):
    """Tiles of 7 rows x 256 cells, z chunks, ragged x: the fused launch equals the CPU definition bit for bit."""
    f = fields(shape)
    ref = [a.copy() for a in f]
    for _ in range(3):
        oracle_cpu.yee3d_step(ref, 0.31, 0.27, order=ORD[order])
    got = gpu_yee3d_fused(shape, f, 3, order)
    for n, g, r in zip(NAMES, got, ref):
        assert np.array_equal(g.view(np.uint32), r.view(np.uint32)), n


def test_yee3d_fused_fullsize(oracle_cpu):
    shape = (256, 256, 256)
    f = fields(shape, seed=5)
    ref = [a.copy() for a in f]
    for _ in range(2):
        oracle_cpu.yee3d_step(ref, 0.31, 0.27, order=1)
    got = gpu_yee3d_fused(shape, f, 2, ORDER_CHAIN)
    for n, g, r in zip(NAMES, got, ref):
        assert np.array_equal(g.view(np.uint32), r.view(np.uint32)), n


@pytest.mark.parametrize("order", [ORDER_CHAIN, ORDER_GOLDEN])
@pytest.mark.parametrize("shape", [(3, 3, 3), (9, 14, 21), (20, 37, 260), (40, 9, 513)])
def test_yee3d_equals_its_cpu_definition(oracle_cpu, shape, order):
    f = fields(shape)
    ref = [a.copy() for a in f]
    for _ in range(4):
        oracle_cpu.yee3d_step(ref, 0.31, 0.27, order=ORD[order])
    got = gpu_yee3d(shape, f, 4, order)
    for n, g, r in zip(NAMES, got, ref):
        assert np.array_equal(g.view(np.uint32), r.view(np.uint32)), n


def test_yee3d_fullsize_and_divergence(oracle_cpu):
    shape = (256, 256, 256)
    f = fields(shape, seed=5)
    ref = [a.copy() for a in f]
    for _ in range(2):
        oracle_cpu.yee3d_step(ref, 0.31, 0.27, order=1)
    got = gpu_yee3d(shape, f, 2, ORDER_CHAIN)
    for n, g, r in zip(NAMES, got, ref):
        assert np.array_equal(g.view(np.uint32), r.view(np.uint32)), n

    def div_h(hx, hy, hz):  # forward differences, like the curl of the H half step, where every cell involved is updated
        s = np.s_[2:-2, 2:-2, 2:-2]
        return ((hx[2:-2, 2:-2, 3:-1] - hx[s]) + (hy[2:-2, 3:-1, 2:-2] - hy[s]) + (hz[3:-1, 2:-2, 2:-2] - hz[s])).astype(np.float64)
    before, after = div_h(*f[3:]), div_h(*got[3:])
    assert np.abs(after - before).max() < 2e-5  # rounding only: the update adds a discrete curl


@pytest.mark.parametrize("order", [ORDER_CHAIN, ORDER_GOLDEN])
@pytest.mark.parametrize("shape", [(2, 2), (17, 33), (200, 301), (2048, 2048)])
def test_yee2d_equals_its_cpu_definition(oracle_cpu, shape, order):
    nx, ny = shape
    steps = 6 if nx < 2048 else 3
    ex, ey, hz = oracle_cpu.yee2d_init(nx, ny)
    dev = [torch.from_numpy(a.copy()).cuda() for a in (ex, ey, hz)]
    lib.yee2d_steps(dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), nx, ny, 0, steps, 0.5, 0.7, order,
                    stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for t in range(steps):
        oracle_cpu.yee2d_step(ex, ey, hz, t, 0.5, 0.7, order=ORD[order])
    for name, d, r in zip(("ex", "ey", "hz"), dev, (ex, ey, hz)):
        assert np.array_equal(d.cpu().numpy().view(np.uint32), r.view(np.uint32)), name


@pytest.mark.parametrize("nslabs,shape,steps,order", [(2, (20, 37, 260), 4, ORDER_CHAIN), (3, (23, 14, 70), 5, ORDER_GOLDEN),
                                                      (4, (256, 256, 256), 3, ORDER_CHAIN)])
def test_yee3d_on_z_slabs_equals_undecomposed(nslabs, shape, steps, order):
    """The six-field step cut into z-slabs (fdtd_amd/slab.py:YeeRunner): the ranks of an nslabs-way split run in
    this one process on the one GPU and hand their boundary planes to each other by device copies (after the H half
    Hx, Hy go up, after the E half Ex, Ey go down).  The gathered fields equal the undecomposed run bit for bit."""
    from fdtd_amd.slab import HipSlabBackend, YeeRunner
    Z, Y, X = shape
    f = fields(shape, seed=8)
    ref = gpu_yee3d(shape, f, steps, order)
    bes = [HipSlabBackend(Z, Y, X, 1, r, nslabs, 0) for r in range(nslabs)]
    for be in bes:
        lo, n = int(be.g.first_global), int(be.g.local_planes)
        for name, a in zip(NAMES, f):
            be.alloc(name)
            be.st.uploadLocal(name, np.ascontiguousarray(a[lo:lo + n]))
    peers = dict(enumerate(bes))
    runs = [YeeRunner(be, NAMES, r, nslabs, exchange="local", order=order, peers=peers) for r, be in enumerate(bes)]
    for _ in range(steps):
        for which, coef in ((0, 0.27), (1, 0.31)):
            for run in runs:
                run.half(which, coef)
    torch.cuda.synchronize()
    for name, r in zip(NAMES, ref):
        got = np.concatenate([be.st.downloadOwned(name) for be in bes], axis=0)
        assert np.array_equal(got.view(np.uint32), r.view(np.uint32)), name
    for be in bes:
        be.st.close()


@pytest.mark.parametrize("own_streams", [False, True], ids=["default-stream", "stream-per-slab"])
@pytest.mark.parametrize("nslabs,shape,steps,order", [(2, (20, 37, 260), 4, ORDER_CHAIN), (3, (23, 14, 70), 5, ORDER_GOLDEN),
                                                      (4, (256, 256, 256), 3, ORDER_CHAIN)])
def test_yee3d_native_slab_half_steps_equal_undecomposed(nslabs, shape, steps, order, own_streams):
    """fdtdx_yee3d_slab_half_begin / fdtdx_step_end: the library itself updates the plane a neighbour needs first, moves
    it (slabs of this process: device copies on its exchange streams) and updates the rest beside that.  Equal to the
    undecomposed run bit for bit, on the default stream and with a stream per slab."""
    Z, Y, X = shape
    f = fields(shape, seed=9)
    ref = gpu_yee3d(shape, f, steps, order)
    slabs = []
    for r in range(nslabs):
        st = Stencil(Z, Y, X, 1, rank=r, nprocs=nslabs, device=0)
        lo, n = int(st.geometry.first_global), int(st.geometry.local_planes)
        for name, a in zip(NAMES, f):
            st.mallocCube(name)
            st.uploadLocal(name, np.ascontiguousarray(a[lo:lo + n]))
        slabs.append(st)
    Stencil.commInitLocal(slabs)
    streams = [torch.cuda.Stream() for _ in slabs] if own_streams else None
    cs = [s.cuda_stream for s in streams] if streams else [None] * nslabs
    for _ in range(steps):
        for which, coef in ((0, 0.27), (1, 0.31)):
            for st, s in zip(slabs, cs):
                st.yee3dSlabHalfBegin(NAMES, which, coef, order, s)
            for st, s in zip(slabs, cs):
                st.stepEnd(s)
    torch.cuda.synchronize()
    for name, r in zip(NAMES, ref):
        got = np.concatenate([st.downloadOwned(name) for st in slabs], axis=0)
        assert np.array_equal(got.view(np.uint32), r.view(np.uint32)), name
    for st in slabs:
        st.close()


def test_yee3d_slab_half_step_over_rccl_loop_back():
    """A middle slab whose two neighbours are itself (a communicator of one): the real ncclSend/ncclRecv group of
    fdtdx_yee3d_slab_half_begin on the exchange stream.  The plane sent up comes back into the low ghost plane, the plane
    sent down into the high one; the update itself equals fdtdx_yee3d_half on the same planes; Ez / Hz ghosts stay."""
    Z, Y, X = 40, 21, 130
    f = fields((Z, Y, X), seed=12)
    cs = torch.cuda.current_stream().cuda_stream
    out = {}
    for native in (False, True):
        with Stencil(Z, Y, X, 1, rank=1, nprocs=3, device=0) as st:
            lo, L = int(st.geometry.first_global), int(st.geometry.local_planes)
            for name, a in zip(NAMES, f):
                st.mallocCube(name)
                st.uploadLocal(name, np.ascontiguousarray(a[lo:lo + L]))
            if native:
                st.commInit(Stencil.commUniqueId(), comm_rank=0, comm_size=1, peer_lo=0, peer_hi=0)
            res = []
            for which, coef in ((0, 0.27), (1, 0.31)):
                if native:
                    st.yee3dSlabHalfBegin(NAMES, which, coef, ORDER_CHAIN, cs)
                    st.stepEnd(cs)
                else:
                    st.yee3dHalf(NAMES, which, coef, ORDER_CHAIN, 1, L - 1, cs)
                st.sync()
                P = int(st.geometry.plane)
                snap = [torch.as_tensor(_CAI(st.getCube(n), L * P), device="cuda").view(L, P).clone() for n in NAMES]
                res.append(snap)
            out[native] = res
    for h, (plain, nat) in enumerate(zip(out[False], out[True])):
        for i, (p, q) in enumerate(zip(plain, nat)):
            # owned planes: the same update (the E half's first owned plane differences the low ghost of Hx, Hy, which
            # only the exchanging run has filled with the new values)
            assert torch.equal(p[1 + h:-1], q[1 + h:-1]), (h, NAMES[i])
    L = out[True][0][0].shape[0]
    for i in (3, 4):                                                        # Hx, Hy: last owned plane -> low ghost, after the H half
        assert torch.equal(out[True][0][i][0], out[True][0][i][L - 2])
        assert torch.equal(out[True][0][i][L - 1], out[False][0][i][L - 1])
    for i in (0, 1):                                                        # Ex, Ey: first owned plane -> high ghost, after the E half
        assert torch.equal(out[True][1][i][L - 1], out[True][1][i][1])
    for i in (2, 5):                                                        # Ez, Hz ghosts untouched
        assert torch.equal(out[True][1][i][0], out[False][1][i][0]) and torch.equal(out[True][1][i][L - 1], out[False][1][i][L - 1])


class _CAI:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def test_yee_runner_native_exchange_on_local_slabs():
    """fdtd_amd/slab.py:YeeRunner(exchange="native") is the library's own split (half_begin on every slab, then half_end)."""
    from fdtd_amd.slab import HipSlabBackend, YeeRunner
    shape, nslabs, steps = (26, 19, 140), 3, 4
    Z, Y, X = shape
    f = fields(shape, seed=10)
    ref = gpu_yee3d(shape, f, steps, ORDER_CHAIN)
    bes = [HipSlabBackend(Z, Y, X, 1, r, nslabs, 0) for r in range(nslabs)]
    for be in bes:
        lo, n = int(be.g.first_global), int(be.g.local_planes)
        for name, a in zip(NAMES, f):
            be.alloc(name)
            be.st.uploadLocal(name, np.ascontiguousarray(a[lo:lo + n]))
    Stencil.commInitLocal([be.st for be in bes])
    runs = [YeeRunner(be, NAMES, r, nslabs, exchange="native") for r, be in enumerate(bes)]
    for _ in range(steps):
        for which, coef in ((0, 0.27), (1, 0.31)):
            for run in runs:
                run.half_begin(which, coef)
            for run in runs:
                run.half_end()
    torch.cuda.synchronize()
    for name, r in zip(NAMES, ref):
        got = np.concatenate([be.st.downloadOwned(name) for be in bes], axis=0)
        assert np.array_equal(got.view(np.uint32), r.view(np.uint32)), name
    for be in bes:
        be.st.close()
