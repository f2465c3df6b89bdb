"""Every compiled tile shape of every sweep, through the C ABI, bit-exact vs the CPU oracle.

Grids are ragged on purpose: X not a multiple of 4 (masked cells inside a quad), more than one
tile in x and y for the narrow shapes, a partial last tile, and a z range that is not a multiple
of the chunk length.
"""
import numpy as np
import pytest

from fdtd_amd import FdtdxError, Stencil, make_program, ORDER_CHAIN, ORDER_GOLDEN
from fdtd_amd import lib

pytestmark = pytest.mark.gpu
ORD = {ORDER_CHAIN: 1, ORDER_GOLDEN: 0}


def sweep_all_variants(cpu, kind, M, order, shape, steps, zchunks):
    Z, Y, X = shape
    if kind == "jacobi7":
        a = cpu.init_jacobi(Z, Y, X)
        b = a.copy()
    else:
        a, b = cpu.init_mt(Z, Y, X)
    ca, cb = a.copy(), b.copy()
    cpu.run(kind, ca, cb, steps, M=M, order=ORD[order])
    seen = 0
    with Stencil(Z, Y, X, M) as st:
        st.mallocCube("p0")
        st.mallocCube("p1")
        prog = make_program(kind, radius=M, order=order)
        v = 0
        while True:
            st.transferCubeToGPU("p0", a)
            st.transferCubeToGPU("p1", b)
            st.setOption("variant", v)
            st.setOption("zchunks", zchunks)
            try:
                st.sweep("p0", "p1", prog, steps)
            except FdtdxError as e:
                assert e.code == lib.ERR_UNSUPPORTED
                break
            st.sync()
            ga, gb = np.empty_like(a), np.empty_like(b)
            st.transferCubeToCPU(ga, "p0")
            st.transferCubeToCPU(gb, "p1")
            info = st.lastLaunchInfo()
            assert np.array_equal(ga.view(np.uint32), ca.view(np.uint32)), info
            assert np.array_equal(gb.view(np.uint32), cb.view(np.uint32)), info
            seen += 1
            v += 1
    return seen


@pytest.mark.parametrize("M", range(1, 9))
def test_star_every_variant(oracle_cpu, M):
    n = sweep_all_variants(oracle_cpu, "star", M, ORDER_CHAIN, (2 * M + 9, 75, 2 * 4 * 64 + 4 * 6 + 3), 2, 3)
    assert n >= 4


@pytest.mark.parametrize("M", [1, 3, 8])
def test_star_golden_every_variant(oracle_cpu, M):
    assert sweep_all_variants(oracle_cpu, "star", M, ORDER_GOLDEN, (2 * M + 7, 41, 273), 2, 2) >= 1


def test_jacobi_every_variant(oracle_cpu):
    assert sweep_all_variants(oracle_cpu, "jacobi7", 1, ORDER_CHAIN, (11, 75, 539), 3, 3) >= 4
    assert sweep_all_variants(oracle_cpu, "jacobi7", 1, ORDER_GOLDEN, (9, 40, 270), 2, 1) >= 1


def test_spc13_every_variant(oracle_cpu):
    assert sweep_all_variants(oracle_cpu, "spc13", 1, ORDER_CHAIN, (11, 75, 539), 3, 3) >= 4
    assert sweep_all_variants(oracle_cpu, "spc13", 1, ORDER_GOLDEN, (9, 40, 270), 2, 2) >= 1


@pytest.mark.parametrize("M", [1, 2, 5, 8])
def test_minimum_grid(oracle_cpu, M):
    """The smallest grid the reference accepts (stencil.h:387: Z >= 2M+1): one interior cell."""
    n = 2 * M + 1
    a, b = oracle_cpu.init_mt(n, n, n)
    ca, cb = a.copy(), b.copy()
    oracle_cpu.run("star", ca, cb, 3, M=M, order=1)
    with Stencil(n, n, n, M) as st:
        st.mallocCube("p0")
        st.mallocCube("p1")
        st.transferCubeToGPU("p0", a)
        st.transferCubeToGPU("p1", b)
        st.sweep("p0", "p1", make_program("star", radius=M), 3)
        st.sync()
        ga, gb = np.empty_like(a), np.empty_like(b)
        st.transferCubeToCPU(ga, "p0")
        st.transferCubeToCPU(gb, "p1")
    assert np.array_equal(ga.view(np.uint32), ca.view(np.uint32))
    assert np.array_equal(gb.view(np.uint32), cb.view(np.uint32))


def test_error_codes_instead_of_asserts():
    """API misuse returns codes (the reference asserts: stencil.h:387,443,470,555)."""
    with pytest.raises(FdtdxError) as e:
        Stencil(8, 8, 8, 9)
    assert e.value.code == lib.ERR_INVALID
    with Stencil(16, 16, 16, 2) as st:
        st.mallocCube("p0")
        with pytest.raises(FdtdxError) as e:
            st.mallocCube("p0")
        assert e.value.code == lib.ERR_EXISTS
        with pytest.raises(FdtdxError) as e:
            st.getCube("nope")
        assert e.value.code == lib.ERR_NOTFOUND
        st.mallocCube("p1")
        with pytest.raises(FdtdxError) as e:
            st.propagate("p0", "p0", make_program("star", radius=2))
        assert e.value.code == lib.ERR_INVALID
        with pytest.raises(FdtdxError) as e:
            st.propagate("p0", "p1", make_program("star", radius=3))  # deeper than the ghost width
        assert e.value.code == lib.ERR_INVALID
        st.free("p1")
        with pytest.raises(FdtdxError):
            st.free("p1")


@pytest.mark.parametrize("kind,M", [("star", 1), ("star", 3), ("star", 8), ("spc13", 1), ("jacobi7", 1)])
def test_two_plane_ranges_in_one_launch(kind, M):
    """fdtdx_propagate_two_ranges: the planes next to either neighbour of a slab swept by ONE launch equal two launches
    bit for bit; either range may be empty; a second range in front of the first is refused."""
    Z, Y, X = 40, 37, 150
    prog = make_program(kind, radius=M)
    with Stencil(Z, Y, X, M) as st:
        for name in ("in", "one", "two"):
            st.mallocCube(name)
        st.initMt("in", "one")
        st.initMt("in", "two")
        lo, hi = M, Z - M
        for (a0, a1, b0, b1) in ((lo, lo + 3, hi - 5, hi), (lo, lo, hi - 2, hi), (lo + 1, lo + 4, hi, hi), (lo, lo + 7, lo + 7, lo + 9)):
            st.propagateTwoRanges("one", "in", prog, a0, a1, b0, b1)
            st.propagatePlanes("two", "in", prog, a0, a1)
            st.propagatePlanes("two", "in", prog, b0, b1)
            st.sync()
            one, two = np.empty((Z, Y, X), np.float32), np.empty((Z, Y, X), np.float32)
            st.transferCubeToCPU(one, "one")
            st.transferCubeToCPU(two, "two")
            assert np.array_equal(one.view(np.uint32), two.view(np.uint32)), (a0, a1, b0, b1)
        with pytest.raises(Exception):
            st.propagateTwoRanges("one", "in", prog, lo + 5, lo + 8, lo, lo + 2)

