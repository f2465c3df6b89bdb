"""Multi-rank z-slab time loop on CPU: world_size 2 and 3 over gloo.

The decomposition (fdtdx_decompose, stencil.h:398-412) and the per-step choreography
(fdtd_amd/slab.py: boundary planes, exchange, interior) are the product code under test; the
per-plane arithmetic is supplied by the CPU oracle through a test-only backend, because there is
no GPU here.  The result must be bit-identical to the undecomposed update
(decomposition invariance), for every rank count.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fdtd_amd import lib, make_program
from fdtd_amd.slab import SlabRunner


class OracleSlabBackend:
    """Padded CPU cubes laid out exactly like the device cubes; sweeps done by the oracle."""

    def __init__(self, cpu, Z, Y, X, M, rank, nprocs):
        self.cpu, self.Y, self.X, self.M = cpu, Y, X, M
        self.Zg, self.fuse = Z, 0  # fuse: deepest fusion this backend offers
        self.g = lib.decompose(Z, Y, X, M, rank, nprocs)
        self.bufs = {}

    def alloc(self, name):
        t = torch.zeros(int(self.g.cube_elems), dtype=torch.float32)
        self.bufs[name] = t
        return t

    def view3(self, name):
        return self.bufs[name].numpy().reshape(int(self.g.local_planes), self.Y, int(self.g.pitch_x))

    def upload_local(self, name, arr):
        self.view3(name)[:, :, :self.X] = arr

    def planes(self, name, first, count):
        p = int(self.g.plane)
        return self.bufs[name][first * p:(first + count) * p]

    def sweep(self, out, inp, program, z_lo, z_hi):
        if z_hi <= z_lo:
            return
        kind = {lib.STAR: "star", lib.SPC13: "spc13", lib.JACOBI7: "jacobi7"}[program.kind]
        src = np.ascontiguousarray(self.view3(inp)[:, :, :self.X])
        dst = np.ascontiguousarray(self.view3(out)[:, :, :self.X])
        self.cpu.step(kind, dst, src, M=program.radius, order=self.cpu.ORDER_CHAIN, z_lo=z_lo, z_hi=z_hi)
        self.view3(out)[z_lo:z_hi, :, :self.X] = dst[z_lo:z_hi]

    def sweep_fused(self, out, inp, mid, program, k, z_lo, z_hi):
        """k steps of radius R the slow way.  Level l is computed on [z_lo-(k-l)R, z_hi+(k-l)R) clipped to the
        interior planes of the GLOBAL grid, into a copy of the cube whose shell that level sees:
        `mid` for odd levels, `inp` for even ones, `out` itself for the last."""
        if z_hi <= z_lo:
            return
        kind = {lib.STAR: "star", lib.JACOBI7: "jacobi7"}[program.kind]
        g = self.g
        gz0, Zg = int(g.first_global), self.Zg
        R = program.radius if program.kind == lib.STAR else 1
        cur = np.ascontiguousarray(self.view3(inp)[:, :, :self.X])
        for l in range(1, k + 1):
            shell = out if l == k else (mid if l % 2 == 1 else inp)
            nxt = np.ascontiguousarray(self.view3(shell)[:, :, :self.X])
            lo = max(z_lo - (k - l) * R, R - gz0)
            hi = min(z_hi + (k - l) * R, Zg - R - gz0)
            self.cpu.step(kind, nxt, cur, M=R, order=self.cpu.ORDER_CHAIN, z_lo=lo, z_hi=hi)
            cur = nxt
        self.view3(out)[z_lo:z_hi, :, :self.X] = cur[z_lo:z_hi]

    def can_fuse(self, program):
        return self.fuse

    # -- composite pipeline (main.cu) on plane ranges, by the oracle -----------------------------------
    def _get(self, name):
        return np.ascontiguousarray(self.view3(name)[:, :, :self.X])

    def _put(self, name, arr, z_lo, z_hi):
        self.view3(name)[z_lo:z_hi, :, :self.X] = arr[z_lo:z_hi]

    def comp_sweep(self, out, inp, vel, comp, z_lo, z_hi):
        if z_hi <= z_lo:
            return
        dst = self._get(out)
        k = np.array([comp.cz, comp.cy, comp.cx, comp.scal_num, comp.scal_den, comp.wx, comp.wy, comp.wz])
        self.cpu.composite_sweep(dst, self._get(inp), self._get(vel), M=comp.radius, coef=k, order=1 - comp.order,
                                 z_lo=z_lo, z_hi=z_hi, zg0=int(self.g.first_global))
        self._put(out, dst, z_lo, z_hi)

    def set_addresses(self, global_addr):
        g = self.g
        z = global_addr // (self.Y * self.X)
        keep = (z >= int(g.owned_begin)) & (z < int(g.owned_begin) + int(g.owned_planes))
        a = global_addr[keep]
        self.addr = (a - np.uint64(int(g.first_global) * self.Y * self.X)).astype(np.uint64)  # local, unpadded

    def inject(self, cube, add, z_lo, z_hi):
        dst = self._get(cube)
        z = self.addr // np.uint64(self.Y * self.X)
        self.cpu.inject(dst, np.ascontiguousarray(self.addr[(z >= z_lo) & (z < z_hi)]), add)
        self._put(cube, dst, z_lo, z_hi)

    def filt(self, cube, sel, scale, order, z_lo, z_hi):
        if z_hi <= z_lo:
            return
        dst = self._get(cube)
        self.cpu.filt(dst, sel[0], sel[1], sel[2], scale, order=1 - order, z_lo=z_lo, z_hi=z_hi, zg0=int(self.g.first_global))
        self._put(cube, dst, z_lo, z_hi)

    def mul_acc(self, s, a, b, radius, order, z_lo, z_hi):
        if z_hi <= z_lo:
            return
        dst = self._get(s)
        self.cpu.mul_acc(dst, self._get(a), self._get(b), M=radius, order=1 - order, z_lo=z_lo, z_hi=z_hi)
        self._put(s, dst, z_lo, z_hi)

    def yee_half(self, fields, which, coef, order, z_lo, z_hi):
        if z_hi <= z_lo:
            return
        f = [self._get(n) for n in fields]
        self.cpu.yee3d_half(f, which, coef, order=1 - order, z_lo=z_lo, z_hi=z_hi)
        for n, a in zip(fields[3:] if which == 0 else fields[:3], f[3:] if which == 0 else f[:3]):
            self._put(n, a, z_lo, z_hi)

    def copy_cube(self, dst, src):
        self.bufs[dst].copy_(self.bufs[src])

    def synchronize(self):
        pass


def _worker(rank, world, port, kind, M, shape, steps, q, ghost=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import cpu
    Z, Y, X = shape
    if kind == "jacobi7":
        a = cpu.init_jacobi(Z, Y, X)
        b = a.copy()
    else:
        a, b = cpu.init_mt(Z, Y, X)
    be = OracleSlabBackend(cpu, Z, Y, X, ghost or M, rank, world)
    be.fuse = (3 if M == 1 else 2) if ghost is not None else 0
    g = be.g
    lo, n = int(g.first_global), int(g.local_planes)
    be.alloc("p0")
    be.alloc("p1")
    be.upload_local("p0", a[lo:lo + n])
    be.upload_local("p1", b[lo:lo + n])
    run = SlabRunner(be, ghost or M, make_program(kind, radius=M), rank, world)
    run.run("p0", "p1", steps)
    o0, o1 = int(g.ghost_lo), int(g.ghost_lo) + int(g.owned_planes)
    q.put((rank, int(g.owned_begin), be.view3("p0")[o0:o1, :, :X].copy(), be.view3("p1")[o0:o1, :, :X].copy()))
    dist.barrier()
    dist.destroy_process_group()


def _composite_worker(rank, world, port, shape, steps, order, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import cpu
    from fdtd_amd.slab import CompositeRunner
    Z, Y, X = shape
    M = 4
    p0, p1, vel = cpu.init_mt(Z, Y, X, nfields=3)
    addr = np.random.default_rng(11).integers(0, Z * Y * X, 400).astype(np.uint64)
    be = OracleSlabBackend(cpu, Z, Y, X, M, rank, world)
    g = be.g
    lo, n = int(g.first_global), int(g.local_planes)
    for name, arr in (("p0", p0), ("p1", p1), ("vel", vel)):
        be.alloc(name)
        be.upload_local(name, arr[lo:lo + n])
    be.alloc("sum")
    be.set_addresses(addr)
    run = CompositeRunner(be, M, lib.make_composite(order=order), (Z // 2, Y // 3, X // 4), 0.9, rank, world)
    s0, s1 = run.run("p0", "p1", "vel", "sum", steps)
    o0, o1 = int(g.ghost_lo), int(g.ghost_lo) + int(g.owned_planes)
    q.put((rank, int(g.owned_begin), [be.view3(nm)[o0:o1, :, :X].copy() for nm in (s0, s1, "sum")]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,steps", [(2, (26, 11, 13), 4), (3, (40, 10, 12), 3)])
def test_composite_pipeline_decomposition_invariance(oracle_cpu, world, shape, steps):
    """main.cu's pipeline on z-slabs (fdtd_amd/slab.py:CompositeRunner, address lists split by owner,
    filt planes in global coordinates) equals the undecomposed pipeline bit for bit."""
    order = lib.ORDER_GOLDEN
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_composite_worker, args=(r, world, port, shape, steps, order, q)) for r in range(world)]
    for p in procs:
        p.start()
    parts = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    Z, Y, X = shape
    got = [np.zeros(shape, np.float32) for _ in range(3)]
    for rank, begin, fields in parts:
        for gfield, f in zip(got, fields):
            gfield[begin:begin + f.shape[0]] = f
    p0, p1, vel = oracle_cpu.init_mt(Z, Y, X, nfields=3)
    s = np.zeros(shape, np.float32)
    addr = np.random.default_rng(11).integers(0, Z * Y * X, 400).astype(np.uint64)
    for _ in range(steps):
        oracle_cpu.main_step(p0, p1, vel, s, addr, order=oracle_cpu.ORDER_GOLDEN)
        p0, p1 = p1, p0
    for gfield, ref in zip(got, (p0, p1, s)):
        assert np.array_equal(gfield.view(np.uint32), ref.view(np.uint32))


YEE = ("ex", "ey", "ez", "hx", "hy", "hz")


def _yee_fields(shape):
    rng = np.random.default_rng(5)
    return [rng.random(shape, dtype=np.float32) for _ in range(6)]


def _yee_worker(rank, world, port, shape, steps, order, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import cpu
    from fdtd_amd.slab import YeeRunner
    Z, Y, X = shape
    be = OracleSlabBackend(cpu, Z, Y, X, 1, rank, world)
    g = be.g
    lo, n = int(g.first_global), int(g.local_planes)
    for name, arr in zip(YEE, _yee_fields(shape)):
        be.alloc(name)
        be.upload_local(name, arr[lo:lo + n])
    YeeRunner(be, YEE, rank, world, order=order).run(steps, 0.31, 0.27)
    o0, o1 = int(g.ghost_lo), int(g.ghost_lo) + int(g.owned_planes)
    q.put((rank, int(g.owned_begin), [be.view3(nm)[o0:o1, :, :X].copy() for nm in YEE]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,steps,order", [(2, (12, 7, 9), 4, lib.ORDER_CHAIN), (3, (17, 6, 8), 3, lib.ORDER_GOLDEN)])
def test_staggered_grid_step_decomposition_invariance(oracle_cpu, world, shape, steps, order):
    """EXTENSION, no reference oracle: the six-field E/H step on z-slabs (fdtd_amd/slab.py:YeeRunner; after each
    half step two fields move one plane in one direction) equals the undecomposed step of its CPU definition."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_yee_worker, args=(r, world, port, shape, steps, order, q)) for r in range(world)]
    for p in procs:
        p.start()
    parts = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got = [np.zeros(shape, np.float32) for _ in range(6)]
    for rank, begin, fields in parts:
        for gfield, f in zip(got, fields):
            gfield[begin:begin + f.shape[0]] = f
    ref = _yee_fields(shape)
    for _ in range(steps):
        oracle_cpu.yee3d_step(ref, 0.31, 0.27, order=1 - order)
    for name, gfield, r in zip(YEE, got, ref):
        assert np.array_equal(gfield.view(np.uint32), r.view(np.uint32)), name


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,kind,M,shape,steps", [
    (2, "star", 1, (14, 9, 11), 5),
    (2, "star", 3, (20, 11, 13), 4),
    (3, "star", 2, (17, 9, 10), 6),   # uneven split: 6,6,5 planes
    (2, "spc13", 1, (12, 8, 9), 5),
    (2, "jacobi7", 1, (10, 9, 12), 4),
])
def test_decomposition_invariance(oracle_cpu, world, kind, M, shape, steps):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, M, shape, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    parts = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    Z, Y, X = shape
    got0, got1 = np.zeros(shape, np.float32), np.zeros(shape, np.float32)
    for rank, begin, p0, p1 in parts:
        got0[begin:begin + p0.shape[0]] = p0
        got1[begin:begin + p1.shape[0]] = p1
    if kind == "jacobi7":
        a = oracle_cpu.init_jacobi(Z, Y, X)
        b = a.copy()
    else:
        a, b = oracle_cpu.init_mt(Z, Y, X)
    oracle_cpu.run(kind, a, b, steps, M=M, order=oracle_cpu.ORDER_CHAIN)
    assert np.array_equal(got0.view(np.uint32), a.view(np.uint32))
    assert np.array_equal(got1.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("world,kind,shape,steps,ghost,M", [
    (2, "star", (16, 9, 11), 6, 2, 1),
    (2, "star", (16, 9, 11), 9, 2, 1),
    (3, "jacobi7", (23, 8, 10), 8, 2, 1),
    (2, "star", (14, 7, 9), 11, 2, 1),
    (2, "star", (16, 9, 11), 7, 3, 1),
    (2, "star", (18, 7, 9), 12, 3, 1),
    (3, "jacobi7", (26, 8, 10), 11, 3, 1),
    (2, "star", (22, 9, 11), 7, 4, 2),   # radius 2: two steps per exchange of four planes
    (3, "star", (31, 9, 10), 8, 4, 2),
])
def test_fused_hops_on_slabs(oracle_cpu, world, kind, shape, steps, ghost, M):
    """Two or three steps per exchange (ghost depth = steps x radius) equal the undecomposed step-by-step update."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, M, shape, steps, q, ghost)) for r in range(world)]
    for p in procs:
        p.start()
    parts = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    Z, Y, X = shape
    got0, got1 = np.zeros(shape, np.float32), np.zeros(shape, np.float32)
    for rank, begin, p0, p1 in parts:
        got0[begin:begin + p0.shape[0]] = p0
        got1[begin:begin + p1.shape[0]] = p1
    if kind == "jacobi7":
        a = oracle_cpu.init_jacobi(Z, Y, X)
        b = a.copy()
    else:
        a, b = oracle_cpu.init_mt(Z, Y, X)
    oracle_cpu.run(kind, a, b, steps, M=M, order=oracle_cpu.ORDER_CHAIN)
    assert np.array_equal(got0.view(np.uint32), a.view(np.uint32))
    assert np.array_equal(got1.view(np.uint32), b.view(np.uint32))


def test_decompose_matches_reference_formulas():
    # stencil.h:398-412 for Z=512, M=4 on 3 ranks: mpiSize = 171,171,170
    sizes, offs = [171, 171, 170], [0, 171, 342]
    for r in range(3):
        g = lib.decompose(512, 512, 512, 4, r, 3)
        assert g.owned_planes == sizes[r] and g.owned_begin == offs[r]
        assert g.local_planes == sizes[r] + 8 - (4 if r == 0 else 0) - (4 if r == 2 else 0)   # size2 / plane
        assert g.first_global == offs[r] - (0 if r == 0 else 4)                               # off2 / plane
        assert g.pitch_x == 544 and g.plane == 512 * 544
    with pytest.raises(lib.FdtdxError):
        lib.decompose(16, 32, 32, 4, 0, 2)   # 8 planes per slab is not > 2M (stencil.h:403)
