"""The HIP sweeps against the fields the reference's own compiled CPU loops produced.

tests/golden/golden_v1.npz holds, for twelve cases, the two cubes after `steps` steps of the golden
loops of fdtd1..8.cu:40-73, fdtdspc.cu:40-55 and jacobi.cu:474-490 (built from the reference's
sources by oracle/build_ref.sh, -O2 -ffp-contract=off), plus CRC32s of the initial fields those runs
started from.  Here every case runs on the GPU through fdtdx_sweep in FDTDX_ORDER_GOLDEN (the host
loop's association, separate IEEE multiplies and adds) and must reproduce the stored bits: the chain
GPU == reference closes on this one box, without passing through the oracle's stepping code.  Only
the initial fields come from the oracle's generator, and their CRC32s are checked against the
fixture first.

The second half bounds the device order (FDTDX_ORDER_CHAIN: every "+ c*v" one FMA, what nvcc makes
of the lambdas) against the golden order at the step counts where values are still normal numbers:
the two are different roundings of the same sum, and the reference's own tolerance is far wider
(fdtd1.cu:93: 1e-2 relative and 1e-3 absolute).
"""
import zlib

import numpy as np
import pytest

from fdtd_amd import Stencil, make_program, ORDER_CHAIN, ORDER_GOLDEN

import golden_util

pytestmark = pytest.mark.gpu

CASES = list(golden_util.cases())


def gpu_sweep(kind, M, a, b, steps, order, temporal=1):
    Z, Y, X = a.shape
    with Stencil(Z, Y, X, M) as st:
        st.mallocCube("p0")
        st.mallocCube("p1")
        st.transferCubeToGPU("p0", a)
        st.transferCubeToGPU("p1", b)
        st.setOption("temporal", temporal)
        st.sweep("p0", "p1", make_program(kind, radius=M, order=order), steps)
        st.sync()
        ga, gb = np.empty_like(a), np.empty_like(b)
        st.transferCubeToCPU(ga, "p0")
        st.transferCubeToCPU(gb, "p1")
        info = st.lastLaunchInfo()
    return ga, gb, info


@pytest.mark.parametrize("temporal", [0, 1], ids=["single-step", "fused-where-possible"])
@pytest.mark.parametrize("case", CASES, ids=[f"{c['kind']}-M{c['M']}-{c['steps']}steps" for c in CASES])
def test_gpu_reproduces_reference_fields(oracle_cpu, case, temporal):
    Z, Y, X = case["shape"]
    if case["kind"] == "jacobi7":
        a = oracle_cpu.init_jacobi(Z, Y, X)
        b = a.copy()
    else:
        a, b = oracle_cpu.init_mt(Z, Y, X)
    # the inputs are the reference's initial fields (fdtd1.cu:21-31, jacobi.cu:412-417)
    assert (zlib.crc32(a.tobytes()), zlib.crc32(b.tobytes())) == case["crc"]
    ga, gb, info = gpu_sweep(case["kind"], case["M"], a, b, case["steps"], ORDER_GOLDEN, temporal)
    assert np.array_equal(ga.view(np.uint32), case["a"].view(np.uint32)), info
    assert np.array_equal(gb.view(np.uint32), case["b"].view(np.uint32)), info


@pytest.mark.parametrize("M", range(1, 9))
def test_device_order_within_tolerance_of_golden_at_50_steps(oracle_cpu, M):
    """50 steps (per-step gain 0.01+0.27*M: radius 1 is down to 1e-28, radius 8 up to 1e17, all normal
    numbers): the device order on the GPU, fused kernels included, stays within 1e-5 relative of the
    reference-pinned golden order computed on the CPU, cell by cell."""
    Z, Y, X = 2 * M + 34, 2 * M + 45, 2 * M + 120
    a, b = oracle_cpu.init_mt(Z, Y, X)
    ca, cb = a.copy(), b.copy()
    oracle_cpu.run("star", ca, cb, 50, M=M, order=oracle_cpu.ORDER_GOLDEN)
    ga, gb, info = gpu_sweep("star", M, a, b, 50, ORDER_CHAIN)
    tiny = np.finfo(np.float32).tiny
    for g, c in ((ga, ca), (gb, cb)):
        assert np.all(np.isfinite(c)) and np.all(np.abs(c[c != 0]) >= tiny), "left the normal range: shorten the run"
        rel = np.abs(g.astype(np.float64) - c.astype(np.float64)) / np.maximum(np.abs(c.astype(np.float64)), tiny)
        assert rel.max() <= 1e-5, (rel.max(), info)


@pytest.mark.parametrize("kind", ["spc13", "jacobi7"])
def test_device_order_within_tolerance_other_programs(oracle_cpu, kind):
    Z, Y, X = 36, 47, 122
    if kind == "jacobi7":
        a = oracle_cpu.init_jacobi(Z, Y, X)
        b = a.copy()
    else:
        a, b = oracle_cpu.init_mt(Z, Y, X)
    ca, cb = a.copy(), b.copy()
    oracle_cpu.run(kind, ca, cb, 50, order=oracle_cpu.ORDER_GOLDEN)
    ga, gb, info = gpu_sweep(kind, 1, a, b, 50, ORDER_CHAIN)
    for g, c in ((ga, ca), (gb, cb)):
        # jacobi's field i - j + 1/(k+1) changes sign across the grid: measure against the field's scale
        scale = np.maximum(np.abs(c.astype(np.float64)), 1e-6 * np.abs(c).max())
        rel = np.abs(g.astype(np.float64) - c.astype(np.float64)) / scale
        assert rel.max() <= 1e-5, (rel.max(), info)
