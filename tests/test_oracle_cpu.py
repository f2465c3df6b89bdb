"""CPU-only checks of the oracle (oracle/fdtd_oracle.c):
  * against the committed golden fields, which the reference's own compiled loops produced;
  * against those loops directly where oracle/_ref is built (this container);
  * device-order (FMA chain) vs golden order stay within the 1e-5 relative bound the
    north-star states, on values in the normal range.
"""
import zlib

import numpy as np
import pytest

from golden_util import cases

CASES = list(cases())


def init(cpu, c):
    nz, ny, nx = c["shape"]
    if c["kind"] == "jacobi7":
        a = cpu.init_jacobi(nz, ny, nx)
        return a, a.copy()
    return cpu.init_mt(nz, ny, nx)


@pytest.mark.parametrize("c", CASES, ids=[f"{c['kind']}{c['M']}-{c['idx']}" for c in CASES])
def test_oracle_reproduces_reference_golden(oracle_cpu, c):
    a, b = init(oracle_cpu, c)
    assert (zlib.crc32(a.tobytes()), zlib.crc32(b.tobytes())) == c["crc"], "initial fields differ from the reference's"
    oracle_cpu.run(c["kind"], a, b, c["steps"], M=c["M"], order=oracle_cpu.ORDER_GOLDEN)
    assert np.array_equal(a.view(np.uint32), c["a"].view(np.uint32))
    assert np.array_equal(b.view(np.uint32), c["b"].view(np.uint32))


@pytest.mark.parametrize("c", CASES, ids=[f"{c['kind']}{c['M']}-{c['idx']}" for c in CASES])
def test_chain_order_within_tolerance_of_golden(oracle_cpu, c):
    a, b = init(oracle_cpu, c)
    oracle_cpu.run(c["kind"], a, b, c["steps"], M=c["M"], order=oracle_cpu.ORDER_CHAIN)
    for got, ref in ((a, c["a"]), (b, c["b"])):
        nrm = np.abs(ref) > 1e-30
        rel = np.abs(got[nrm] - ref[nrm]) / np.abs(ref[nrm])
        assert rel.max() <= 1e-5  # fp32 max relative error bound of the north-star


def test_oracle_matches_compiled_reference_loops(oracle_cpu):
    if not oracle_cpu.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    for kind, M, shape, steps in (("star", 3, (15, 17, 33), 9), ("star", 8, (19, 23, 41), 3), ("spc13", 1, (11, 12, 29), 8),
                                  ("jacobi7", 1, (9, 16, 31), 10)):
        nz, ny, nx = shape
        if kind == "jacobi7":
            a = oracle_cpu.init_jacobi(nz, ny, nx)
            b = a.copy()
            ra, rb = oracle_cpu.ref_init_jacobi(nz, ny, nx)
        else:
            a, b = oracle_cpu.init_mt(nz, ny, nx)
            ra, rb = oracle_cpu.ref_init_mt(nz, ny, nx)
        assert np.array_equal(a, ra) and np.array_equal(b, rb)
        oracle_cpu.run(kind, a, b, steps, M=M, order=oracle_cpu.ORDER_GOLDEN)
        oracle_cpu.ref_run(kind, ra, rb, steps, M=M)
        assert np.array_equal(a.view(np.uint32), ra.view(np.uint32))
        assert np.array_equal(b.view(np.uint32), rb.view(np.uint32))


def test_slab_init_matches_whole_grid_init(oracle_cpu):
    a, b = oracle_cpu.init_mt(12, 9, 10)
    sa, sb = oracle_cpu.init_mt(5, 9, 10, z_begin=4)
    assert np.array_equal(sa, a[4:9]) and np.array_equal(sb, b[4:9])
