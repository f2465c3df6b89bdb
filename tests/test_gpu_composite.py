"""The composite pipeline of main.cu (sweep + inject + filt + mul, main.cu:149-171) through the C ABI:
bit-exact against the CPU oracle in both orders, against the committed fields of the reference's own
compiled host loop (host order), for every compiled tile shape, on ragged grids, at 512^3, and split
into two z-slabs (plane-range forms of all four operations, ghost planes traded by device copies)."""
import os

import numpy as np
import pytest
import torch

from fdtd_amd import FdtdxError, Stencil, make_composite, ORDER_CHAIN, ORDER_GOLDEN
from fdtd_amd import lib

pytestmark = pytest.mark.gpu
ORD = {ORDER_CHAIN: 1, ORDER_GOLDEN: 0}
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_main_v1.npz")


def gpu_pipeline(shape, p0, p1, vel, s, addr, steps, order, variant=-1, zchunks=0, fused=False):
    Z, Y, X = shape
    with Stencil(Z, Y, X, 4) as st:
        for n in ("p0", "p1", "vel", "sum"):
            st.mallocCube(n)
        st.transferCubeToGPU("p0", p0)
        st.transferCubeToGPU("p1", p1)
        st.transferCubeToGPU("vel", vel)
        st.transferCubeToGPU("sum", s)
        st.mallocAddr("addr", len(addr))
        st.transferAddrToGPU("addr", addr)
        assert st.getAddrSize("addr") == len(addr)
        st.setOption("variant", variant)
        st.setOption("zchunks", zchunks)
        comp = make_composite(order=order)
        s0, s1 = "p0", "p1"
        for _ in range(steps):
            if fused:
                st.compositeStep(s0, s1, "vel", "sum", "addr", comp, 1.0, Z // 2, Y // 3, X // 4, 0.9)
            else:
                st.propagateComposite(s0, s1, "vel", comp)
                st.inject(s0, "addr", 1.0)
                st.filt(s0, Z // 2, Y // 3, X // 4, 0.9, order)
                st.mulAcc("sum", s0, s1, 4, order)
            s0, s1 = s1, s0
        st.sync()
        out = [np.empty(shape, np.float32) for _ in range(3)]
        st.transferCubeToCPU(out[0], s0)
        st.transferCubeToCPU(out[1], s1)
        st.transferCubeToCPU(out[2], "sum")
        info = st.lastLaunchInfo()
    return out, info


def oracle_pipeline(cpu, p0, p1, vel, s, addr, steps, order):
    p0, p1, s = p0.copy(), p1.copy(), s.copy()
    for _ in range(steps):
        cpu.main_step(p0, p1, vel, s, addr, order=ORD[order])
        p0, p1 = p1, p0
    return [p0, p1, s]


def inputs(cpu, shape, na, seed=1):
    nz, ny, nx = shape
    p0, p1, vel = cpu.init_mt(nz, ny, nx, nfields=3)
    addr = np.random.default_rng(seed).integers(0, nz * ny * nx, na).astype(np.uint64)
    if na > 8:
        addr[5] = addr[2]  # a repeated address adds twice
    return p0, p1, vel, np.zeros(shape, np.float32), addr


def same(got, ref, info=""):
    for name, g, r in zip(("s0", "s1", "sum"), got, ref):
        bad = np.argwhere(g.view(np.uint32) != r.view(np.uint32))
        assert len(bad) == 0, f"{name}: {len(bad)} cells differ, first {bad[:3].tolist()} {info}"


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("order", [ORDER_CHAIN, ORDER_GOLDEN])
def test_every_composite_shape(oracle_cpu, order, fused):
    shape = (17, 75, 2 * 4 * 64 + 27)
    p0, p1, vel, s, addr = inputs(oracle_cpu, shape, 2000)
    # injection addresses on the planes filt scales, on the shell and next to it: every case of the fused step
    Z, Y, X = shape
    addr[:6] = [(Z // 2) * Y * X + 9 * X + 11, 5 * Y * X + (Y // 3) * X + 40, 6 * Y * X + 20 * X + X // 4, 3, 4 * Y * X + 4 * X + 4,
                (Z // 2) * Y * X + (Y // 3) * X + X // 4]
    ref = oracle_pipeline(oracle_cpu, p0, p1, vel, s, addr, 3, order)
    v = 0
    while True:
        try:
            got, info = gpu_pipeline(shape, p0, p1, vel, s, addr, 3, order, variant=v, zchunks=2, fused=fused)
        except FdtdxError as e:
            assert e.code == lib.ERR_UNSUPPORTED
            break
        same(got, ref, info)
        v += 1
    assert v >= 3


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("shape", [(9, 9, 9), (12, 11, 40), (23, 40, 130)])
def test_composite_small_grids(oracle_cpu, shape, fused):
    p0, p1, vel, s, addr = inputs(oracle_cpu, shape, 50)
    for order in (ORDER_CHAIN, ORDER_GOLDEN):
        same(gpu_pipeline(shape, p0, p1, vel, s, addr, 5, order, fused=fused)[0],
             oracle_pipeline(oracle_cpu, p0, p1, vel, s, addr, 5, order))


def test_host_order_equals_reference_fixture(oracle_cpu):
    z = np.load(GOLD, allow_pickle=False)
    shape, steps, addr = tuple(int(v) for v in z["shape"]), int(z["steps"]), z["addr"]
    p0, p1, vel = oracle_cpu.init_mt(*shape, nfields=3)
    got, _ = gpu_pipeline(shape, p0, p1, vel, np.zeros(shape, np.float32), addr, steps, ORDER_GOLDEN)
    same(got, [z["p0"], z["p1"], z["sum"]])


@pytest.mark.parametrize("order,fused", [(ORDER_CHAIN, False), (ORDER_GOLDEN, False), (ORDER_CHAIN, True)])
def test_composite_fullsize(oracle_cpu, order, fused):
    shape = (512, 512, 512)
    p0, p1, vel, s, addr = inputs(oracle_cpu, shape, 10000)  # main.cu:11
    same(gpu_pipeline(shape, p0, p1, vel, s, addr, 2, order, fused=fused)[0], oracle_pipeline(oracle_cpu, p0, p1, vel, s, addr, 2, order))


def test_two_slabs_equal_one(oracle_cpu):
    """The plane-range forms: boundary planes first, ghost planes of the new field traded (by device
    copies here; main.cu:164-165), then the interior; equal to the undecomposed pipeline."""
    shape, steps, order, M = (40, 33, 70), 4, ORDER_CHAIN, 4
    Z, Y, X = shape
    p0, p1, vel, s, addr = inputs(oracle_cpu, shape, 3000)
    ref = oracle_pipeline(oracle_cpu, p0, p1, vel, s, addr, steps, order)
    sts = [Stencil(Z, Y, X, M, rank=r, nprocs=2, device=0) for r in range(2)]
    comp = make_composite(order=order)
    bufs = [{}, {}]
    try:
        for st, b in zip(sts, bufs):
            g = st.geometry
            lo, n = int(g.first_global), int(g.local_planes)
            for name, arr in (("p0", p0), ("p1", p1), ("vel", vel), ("sum", s)):
                b[name] = torch.zeros(int(g.cube_elems), dtype=torch.float32, device="cuda:0")
                st.attachCube(name, b[name].data_ptr(), b[name].numel() * 4)
                st.uploadLocal(name, arr[lo:lo + n])
            st.mallocAddr("addr", len(addr))
            st.transferAddrToGPU("addr", addr)
        assert sum(st.getAddrSize("addr") for st in sts) == len(addr)
        p = int(sts[0].geometry.plane)
        l0 = int(sts[0].geometry.local_planes)

        def exchange(name):
            bufs[1][name][0:M * p].copy_(bufs[0][name][(l0 - 2 * M) * p:(l0 - M) * p])
            bufs[0][name][(l0 - M) * p:l0 * p].copy_(bufs[1][name][M * p:2 * M * p])

        s0, s1 = "p0", "p1"
        for _ in range(steps):
            for st in sts:
                g = st.geometry
                lo = int(g.ghost_lo) if g.ghost_lo else M
                hi = int(g.local_planes) - (int(g.ghost_hi) if g.ghost_hi else M)
                olo, ohi = int(g.ghost_lo), int(g.ghost_lo) + int(g.owned_planes)
                # boundary planes towards the neighbour first, then the rest (main.cu:153-163)
                cut = [(hi - M, hi), (lo, hi - M)] if g.ghost_hi else [(lo, lo + M), (lo + M, hi)]
                for a, b in cut:
                    st.propagateComposite(s0, s1, "vel", comp, a, b)
                st.inject(s0, "addr", 1.0, olo, ohi)
                st.filt(s0, Z // 2, Y // 3, X // 4, 0.9, order, olo, ohi)
                st.mulAcc("sum", s0, s1, M, order, lo, hi)
            torch.cuda.synchronize()
            exchange(s0)
            s0, s1 = s1, s0
        torch.cuda.synchronize()
        got = [np.concatenate([st.downloadOwned(n) for st in sts], axis=0) for n in (s0, s1, "sum")]
    finally:
        for st in sts:
            st.close()
    same(got, ref)
