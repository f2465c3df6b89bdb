"""Full-size checks (512^3, the grid of the reference drivers, fdtd1.cu:8-10) of the HIP sweeps.

  * direct bit-exact comparison with the CPU oracle for a few steps (the oracle finishes a
    512^3 step in well under a second per radius);
  * size-independent properties: exact linearity under power-of-two scaling, and
    decomposition invariance (one slab vs two z-slabs exchanging ghost planes on the same GPU);
  * the 1000-step value classes: with U[0,1) inputs the interiors of radius 1..3 underflow to
    zero/denormal and radius 5..8 overflow to inf long before step 1000 (gain per step is
    0.01+0.27*M); only the classes are compared there.
"""
import numpy as np
import pytest
import torch

from fdtd_amd import Stencil, make_program
from fdtd_amd.slab import HipSlabBackend, SlabRunner

pytestmark = pytest.mark.gpu
N = 512


def gpu_run(kind, M, a, b, steps):
    Z, Y, X = a.shape
    with Stencil(Z, Y, X, M) as st:
        st.mallocCube("p0")
        st.mallocCube("p1")
        st.transferCubeToGPU("p0", a)
        st.transferCubeToGPU("p1", b)
        st.sweep("p0", "p1", make_program(kind, radius=M), steps)
        st.sync()
        ga, gb = np.empty_like(a), np.empty_like(b)
        st.transferCubeToCPU(ga, "p0")
        st.transferCubeToCPU(gb, "p1")
    return ga, gb


@pytest.mark.parametrize("kind,M,steps", [("star", 1, 3), ("star", 4, 2), ("star", 8, 2), ("spc13", 1, 2), ("jacobi7", 1, 3),
                                           # the remaining radii at the drivers' size: their default tile shapes and z chunking
                                           ("star", 3, 2), ("star", 5, 2), ("star", 6, 2), ("star", 7, 2),
                                           # four steps and more run through the fused kernels (three steps per launch at
                                           # radius 1, two at radius 2) and finish with single steps
                                           ("star", 1, 7), ("jacobi7", 1, 8), ("star", 2, 6), ("spc13", 1, 9)])
def test_fullsize_vs_oracle(oracle_cpu, kind, M, steps):
    if kind == "jacobi7":
        a = oracle_cpu.init_jacobi(N, N, N)
        b = a.copy()
    else:
        a, b = oracle_cpu.init_mt(N, N, N)
    ga, gb = gpu_run(kind, M, a, b, steps)
    oracle_cpu.run(kind, a, b, steps, M=M, order=oracle_cpu.ORDER_CHAIN)
    assert np.array_equal(ga.view(np.uint32), a.view(np.uint32))
    assert np.array_equal(gb.view(np.uint32), b.view(np.uint32))


def test_device_init_matches_reference_init(oracle_cpu):
    """fdtdx_init_mt reproduces fdtd1.cu:21-31 (checked through the oracle's pinned generator)."""
    Z, Y, X = 40, 64, 100
    a, b = oracle_cpu.init_mt(Z, Y, X)
    with Stencil(Z, Y, X, 2) as st:
        st.mallocCube("p0")
        st.mallocCube("p1")
        st.initMt("p0", "p1")
        ga, gb = np.empty_like(a), np.empty_like(b)
        st.transferCubeToCPU(ga, "p0")
        st.transferCubeToCPU(gb, "p1")
    assert np.array_equal(ga, a) and np.array_equal(gb, b)


@pytest.mark.parametrize("M", [2, 4])
def test_linearity_under_power_of_two_scaling(M):
    """f(4*u) == 4*f(u) bit for bit: the update is linear and scaling by 4 is exact in fp32."""
    outs = []
    for scale in (1.0, 4.0):
        with Stencil(N, N, N, M) as st:
            st.mallocCube("p0")
            st.mallocCube("p1")
            st.initMt("p0", "p1")
            if scale != 1.0:
                for name in ("p0", "p1"):
                    # scale in place on the device through a torch view of the cube
                    torch_view(st.getCube(name), int(st.geometry.cube_elems)).mul_(scale)
            st.sweep("p0", "p1", make_program("star", radius=M), 6)
            st.sync()
            outs.append((torch_view(st.getCube("p0"), int(st.geometry.cube_elems)).clone(),
                         torch_view(st.getCube("p1"), int(st.geometry.cube_elems)).clone()))
    for u, v in zip(outs[0], outs[1]):
        assert torch.equal(u * 4.0, v)


class _CAI:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def torch_view(ptr, n):
    """Zero-copy torch tensor over device memory owned by the library."""
    return torch.as_tensor(_CAI(ptr, n), device="cuda")


@pytest.mark.parametrize("kind,M", [("star", 1), ("star", 4), ("star", 8), ("spc13", 1)])
def test_two_slabs_equal_one(kind, M):
    """Decomposition invariance at full size: ranks 0 and 1 of a 2-way z split run in this one
    process on the same GPU; their ghost planes are exchanged by device copies between the
    regions fdtdx_halo_region names.  The gathered result equals the single-slab run bit for bit."""
    Z, Y, X, steps = 256, N, N, 5
    prog = make_program(kind, radius=M)
    with Stencil(Z, Y, X, M) as st:
        st.mallocCube("p0")
        st.mallocCube("p1")
        st.initMt("p0", "p1")
        st.sweep("p0", "p1", prog, steps)
        st.sync()
        ref0, ref1 = np.empty((Z, Y, X), np.float32), np.empty((Z, Y, X), np.float32)
        st.transferCubeToCPU(ref0, "p0")
        st.transferCubeToCPU(ref1, "p1")
    bes = [HipSlabBackend(Z, Y, X, M, r, 2, 0) for r in range(2)]
    for be in bes:
        be.alloc("p0")
        be.alloc("p1")
        be.st.initMt("p0", "p1")
    runs = [SlabRunner(be, M, prog, r, 1) for r, be in enumerate(bes)]  # nprocs=1: no torch.distributed ops
    for r, run in enumerate(runs):  # ...but keep the neighbour flags of the real 2-way split
        run.has_lo, run.has_hi = r == 1, r == 0
    out, inp = "p0", "p1"
    for _ in range(steps):
        for run in runs:
            run.step(out, inp)
        # rank 0's last M owned planes -> rank 1's low ghosts, rank 1's first M owned -> rank 0's high ghosts
        l0, l1 = runs[0].local, runs[1].local
        bes[1].planes(out, 0, M).copy_(bes[0].planes(out, l0 - 2 * M, M))
        bes[0].planes(out, l0 - M, M).copy_(bes[1].planes(out, M, M))
        out, inp = inp, out
    torch.cuda.synchronize()
    got0 = np.concatenate([be.st.downloadOwned("p0") for be in bes], axis=0)
    got1 = np.concatenate([be.st.downloadOwned("p1") for be in bes], axis=0)
    assert np.array_equal(got0.view(np.uint32), ref0.view(np.uint32))
    assert np.array_equal(got1.view(np.uint32), ref1.view(np.uint32))


@pytest.mark.parametrize("kind,R,ghost,steps,n", [("star", 4, 4, 3, 2), ("star", 1, 3, 8, 4), ("spc13", 1, 2, 6, 2), ("star", 8, 8, 2, 8)])
def test_config4_grid_on_slabs_equals_undecomposed(kind, R, ghost, steps, n):
    """The 2048 x 1024 x 1024 grid of the multi-GPU configuration (2^31 cells, cubes of 2.2e9 elements) cut into n z-slabs
    that run the library's own multi-slab loop in this one process on the one GPU (ghost planes exchanged by device copies,
    fused hops where the ghost depth allows): every owned plane equals the undecomposed run bit for bit."""
    Z, Y, X = 2048, 1024, 1024
    prog = make_program(kind, radius=R)
    gen = torch.Generator(device="cuda").manual_seed(19950918)
    full = HipSlabBackend(Z, Y, X, ghost, 0, 1, 0)
    init = {}
    for name in ("p0", "p1"):
        full.alloc(name).uniform_(generator=gen)
        init[name] = full.bufs[name].clone()
    slabs = [HipSlabBackend(Z, Y, X, ghost, r, n, 0) for r in range(n)]
    P = int(full.g.plane)
    for be in slabs:
        lo, cnt = int(be.g.first_global), int(be.g.local_planes)
        for name in ("p0", "p1"):
            be.alloc(name).copy_(init[name][lo * P:(lo + cnt) * P])
    del init
    Stencil.run([full.st], "p0", "p1", prog, steps)
    Stencil.commInitLocal([be.st for be in slabs])
    Stencil.run([be.st for be in slabs], "p0", "p1", prog, steps)
    torch.cuda.synchronize()
    info = slabs[0].st.lastLaunchInfo()
    for be in slabs:
        first, glo, own = int(be.g.first_global), int(be.g.ghost_lo), int(be.g.owned_planes)
        for name in ("p0", "p1"):
            got = be.bufs[name][glo * P:(glo + own) * P].view(torch.int32)
            ref = full.bufs[name][(first + glo) * P:(first + glo + own) * P].view(torch.int32)
            assert torch.equal(got, ref), (name, first, info)
    for be in slabs + [full]:
        be.st.close()


@pytest.mark.parametrize("M", [1, 4, 8])
def test_thousand_step_value_classes(oracle_cpu, M):
    """The drivers' step count (1000, fdtd1.cu:68) on a reduced grid the oracle can follow: identical zero / denormal /
    normal / inf / nan classes everywhere, and identical bits (same arithmetic, same order)."""
    Z, Y, X = 40, 48, 72
    a, b = oracle_cpu.init_mt(Z, Y, X)
    ga, gb = gpu_run("star", M, a, b, 1000)
    with np.errstate(all="ignore"):
        oracle_cpu.run("star", a, b, 1000, M=M, order=oracle_cpu.ORDER_CHAIN)

    def classes(x):
        ax = np.abs(x)
        tiny = np.finfo(np.float32).tiny
        return np.select([np.isnan(x), np.isinf(x), ax == 0, ax < tiny], [4, 3, 0, 1], 2)

    assert np.array_equal(classes(ga), classes(a)) and np.array_equal(classes(gb), classes(b))
    fin = np.isfinite(a)
    assert np.array_equal(ga[fin].view(np.uint32), a[fin].view(np.uint32))


@pytest.mark.parametrize("kind,M", [("star", 2), ("spc13", 1), ("jacobi7", 1)])
def test_offsets_beyond_32_bits(oracle_cpu, kind, M):
    """A cube of more than 2^31 elements (the reference's uint32_t indices stop at 2^32,
    fdtd1.cu:13-14): a block of planes at the far end of the cube, whose element offsets need
    more than 31 bits and whose byte offsets more than 33, must come out as the oracle computes
    it for the same block taken alone.  Rows are 1000 cells wide: two tiles in x for every shape."""
    Z, Y, X, K = 2100, 1024, 1000, 12
    rng = np.random.default_rng(7)
    hin = rng.random((K, Y, X), dtype=np.float32)
    hout = rng.random((K, Y, X), dtype=np.float32)
    exp = hout.copy()
    oracle_cpu.step(kind, exp, hin, M=M, order=1)
    dev = torch.device("cuda", 0)
    with Stencil(Z, Y, X, M) as st:
        g = st.geometry
        n, px = int(g.cube_elems), int(g.pitch_x)
        assert n > 2 ** 31 and n * 4 > 2 ** 33
        a = torch.zeros(n, dtype=torch.float32, device=dev)
        b = torch.zeros(n, dtype=torch.float32, device=dev)
        st.attachCube("p0", a.data_ptr(), n * 4)
        st.attachCube("p1", b.data_ptr(), n * 4)
        z0 = Z - K
        a.view(Z, Y, px)[z0:, :, :X].copy_(torch.from_numpy(hout))
        b.view(Z, Y, px)[z0:, :, :X].copy_(torch.from_numpy(hin))
        st.propagatePlanes("p0", "p1", make_program(kind, radius=M), z0 + M, Z - M, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = a.view(Z, Y, px)[z0:, :, :X].cpu().numpy()
        below = a.view(Z, Y, px)[z0 - 1].abs().max().item()
    del a, b
    torch.cuda.empty_cache()
    assert below == 0.0  # nothing written under the requested range
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
