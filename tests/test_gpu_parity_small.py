"""GPU parity, small grids: HIP sweeps through the C ABI vs the CPU oracle, bit for bit.

Follows the reference drivers' own self-check (fdtd1.cu:86-99: run both, compare every cell)
with the tolerance tightened to exact equality:
  * device order (FMA chain)   == oracle ORDER_CHAIN, bit-exact
  * golden order               == oracle ORDER_GOLDEN == the reference's compiled CPU loop, bit-exact
"""
import numpy as np
import pytest

from fdtd_amd import Stencil, make_program, ORDER_CHAIN, ORDER_GOLDEN

pytestmark = pytest.mark.gpu

ORD = {ORDER_CHAIN: 1, ORDER_GOLDEN: 0}  # fdtdx order -> oracle order


def run_gpu(kind, M, a, b, steps, order, variant=-1, zchunks=0):
    Z, Y, X = a.shape
    with Stencil(Z, Y, X, M) as st:
        st.mallocCube("p0", True)
        st.mallocCube("p1", True)
        st.transferCubeToGPU("p0", a)
        st.transferCubeToGPU("p1", b)
        st.setOption("variant", variant)
        st.setOption("zchunks", zchunks)
        prog = make_program(kind, radius=M, order=order)
        st.sweep("p0", "p1", prog, steps)
        st.sync()
        ga, gb = np.zeros_like(a), np.zeros_like(b)
        st.transferCubeToCPU(ga, "p0")
        st.transferCubeToCPU(gb, "p1")
        info = st.lastLaunchInfo()
    return ga, gb, info


def same_bits(x, y):
    return np.array_equal(x.view(np.uint32), y.view(np.uint32))


@pytest.mark.parametrize("M", range(1, 9))
@pytest.mark.parametrize("order", [ORDER_CHAIN, ORDER_GOLDEN])
def test_star_small_odd(oracle_cpu, M, order):
    Z, Y, X = 2 * M + 11, 2 * M + 21, 2 * M + 37  # odd, ragged: partial tiles and masked quads
    a, b = oracle_cpu.init_mt(Z, Y, X)
    ca, cb = a.copy(), b.copy()
    oracle_cpu.run("star", ca, cb, 3, M=M, order=ORD[order])
    ga, gb, info = run_gpu("star", M, a, b, 3, order)
    assert same_bits(ga, ca), info
    assert same_bits(gb, cb), info


@pytest.mark.parametrize("order", [ORDER_CHAIN, ORDER_GOLDEN])
def test_spc13_small(oracle_cpu, order):
    Z, Y, X = 13, 29, 75
    a, b = oracle_cpu.init_mt(Z, Y, X)
    ca, cb = a.copy(), b.copy()
    oracle_cpu.run("spc13", ca, cb, 4, order=ORD[order])
    ga, gb, info = run_gpu("spc13", 1, a, b, 4, order)
    assert same_bits(ga, ca), info
    assert same_bits(gb, cb), info


@pytest.mark.parametrize("order", [ORDER_CHAIN, ORDER_GOLDEN])
def test_jacobi7_small(oracle_cpu, order):
    Z, Y, X = 17, 23, 70
    a = oracle_cpu.init_jacobi(Z, Y, X)
    b = a.copy()
    ca, cb = a.copy(), b.copy()
    oracle_cpu.run("jacobi7", ca, cb, 5, order=ORD[order])
    ga, gb, info = run_gpu("jacobi7", 1, a, b, 5, order)
    assert same_bits(ga, ca), info
    assert same_bits(gb, cb), info
