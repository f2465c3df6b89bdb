"""z-slab time loop with ghost-plane exchange: one process per GPU, torch.distributed P2P.

Replaces the per-step choreography of the reference drivers (fdtd1.cu:68-80) and the halo
machinery of stencil.h:639-682 / 770-808:

  reference (per step)                          here (per step)
  ------------------------------------------   ---------------------------------------------
  backupCubeHaloBackup   D2D 3M planes x2       --  (ghost planes live inside the cube)
  propagateHalo{Top,Buttom}Backup (M planes)    sweep the M boundary planes next to each face
  sync; propagate (whole slab, async)           start isend/irecv of those planes (RCCL stream)
  commCubeHaloBackup     host-blocking MPI      sweep the interior planes (compute stream, concurrently)
  sync; restoreCubeHaloBackup D2D x2; sync      wait for the exchange; swap

The boundary planes are written once (not twice as in the reference) and are received
directly into the neighbour's ghost planes of the ping-pong output cube, so there are no
staging copies.  torch.distributed's "nccl" backend is RCCL on ROCm: sends and receives run on
RCCL's own HIP stream, ordered after the boundary sweep and overlapped with the interior sweep.
The same code runs on CPU tensors over "gloo" (tests), with the sweep supplied by the caller.

Every step exchanges the cube it has just written (fdtd1.cu:70-77 does the same); since both ping-pong cubes
take that role in turn, the ghost planes of both stay current and the update does not depend on the decomposition.
"""
import ctypes

import torch
import torch.distributed as dist

from . import lib as _lib
from .stencil import Stencil


class HipSlabBackend:
    """Cubes in HBM owned by torch, sweeps by libfdtdx.so on torch's current stream."""

    def __init__(self, Z, Y, X, M, rank, nprocs, device):
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.st = Stencil(Z, Y, X, M, rank=rank, nprocs=nprocs, device=device)
        self.g = self.st.geometry
        self.bufs = {}
        # optional second stream for the boundary-plane sweeps (SlabRunner(side_stream=True))
        self.side = torch.cuda.Stream(self.device)

    def fork_side(self):
        """Work issued on the side stream from here on sees everything queued on the main stream so far."""
        self.side.wait_stream(torch.cuda.current_stream())

    def join_side(self):
        """Work issued on the main stream from here on sees everything queued on the side stream so far."""
        torch.cuda.current_stream().wait_stream(self.side)

    def side_ctx(self):
        return torch.cuda.stream(self.side)

    def _stream(self, side):
        return self.side.cuda_stream if side else torch.cuda.current_stream().cuda_stream

    def alloc(self, name):
        t = torch.zeros(int(self.g.cube_elems), dtype=torch.float32, device=self.device)
        self.bufs[name] = t
        self.st.attachCube(name, t.data_ptr(), t.numel() * 4)
        return t

    def planes(self, name, first, count):
        p = int(self.g.plane)
        return self.bufs[name][first * p:(first + count) * p]

    def sweep(self, out, inp, program, z_lo, z_hi, side=False):
        if z_hi > z_lo:
            self.st.propagatePlanes(out, inp, program, z_lo, z_hi, self._stream(side))

    def sweep_fused(self, out, inp, mid, program, k, z_lo, z_hi, side=False):
        """k (2 or 3) fused steps out = f^k(inp) on planes [z_lo, z_hi) (fdtdx_propagate_fused_planes)."""
        if z_hi > z_lo:
            self.st.propagateFusedPlanes(out, inp, mid, program, k, z_lo, z_hi, self._stream(side))

    # -- the composite pipeline of main.cu on plane ranges ---------------------------------------------
    def comp_sweep(self, out, inp, vel, comp, z_lo, z_hi):
        if z_hi > z_lo:
            self.st.propagateComposite(out, inp, vel, comp, z_lo, z_hi, torch.cuda.current_stream().cuda_stream)

    def set_addresses(self, global_addr):
        """The rank keeps the addresses whose plane it owns (stencil.h:518-544)."""
        if "addr" not in self.__dict__.setdefault("_lists", set()):
            self.st.mallocAddr("addr", len(global_addr))
            self._lists.add("addr")
        self.st.transferAddrToGPU("addr", global_addr)

    def inject(self, cube, add, z_lo, z_hi):
        if z_hi > z_lo:
            self.st.inject(cube, "addr", add, z_lo, z_hi, torch.cuda.current_stream().cuda_stream)

    def filt(self, cube, sel, scale, order, z_lo, z_hi):
        if z_hi > z_lo:
            self.st.filt(cube, sel[0], sel[1], sel[2], scale, order, z_lo, z_hi, torch.cuda.current_stream().cuda_stream)

    def mul_acc(self, s, a, b, radius, order, z_lo, z_hi):
        if z_hi > z_lo:
            self.st.mulAcc(s, a, b, radius, order, z_lo, z_hi, torch.cuda.current_stream().cuda_stream)

    def yee_half(self, fields, which, coef, order, z_lo, z_hi):
        """One half step of the staggered-grid extension on planes [z_lo, z_hi) (fdtdx_yee3d_half)."""
        if z_hi > z_lo:
            self.st.yee3dHalf(fields, which, coef, order, z_lo, z_hi, torch.cuda.current_stream().cuda_stream)

    def can_fuse(self, program):
        """Deepest fusion the library has a kernel for (0: none)."""
        # asked of the library, so that this loop steps singly exactly where the native loops do (tb_kernel.h: radius 1, two or
        # three steps; tbm_kernel.h: radius 2, two steps; tbs_kernel.h: 13-point, two steps, not on rows of exactly 65 quads)
        return self.st.fusedDepth(program)

    def copy_cube(self, dst, src):
        self.bufs[dst].copy_(self.bufs[src])

    def synchronize(self):
        torch.cuda.synchronize(self.device)


class SlabRunner:
    """The time loop of one rank.  `backend` provides alloc/planes/sweep/synchronize and `.g`."""

    def __init__(self, backend, M, program, rank, nprocs, group=None, exchange="device", side_stream=None):
        """exchange: "device" hands the cube's own planes to the process group (RCCL on GPUs, gloo on
        CPU tensors); "host" stages them through CPU tensors (rehearsals of the multi-rank loop on
        a box whose ranks share one GPU, where RCCL cannot be used); "none" skips the exchange
        (host-cost measurements of the launch sequence only: results are wrong)."""
        self.b, self.M, self.prog = backend, M, program
        self.rank, self.nprocs, self.group = rank, nprocs, group
        self.exchange = exchange
        # True: boundary launches and the exchange go to a second stream beside the interior sweep.
        # Measured slower on MI355X (the boundary launches are short, the stream hand-overs are not:
        # 114 vs 111 us per step for the fused jacobi loop at 512^3), so off unless asked for.
        self.side_stream = bool(side_stream)
        g = backend.g
        self.local = int(g.local_planes)
        self.has_lo = int(g.ghost_lo) > 0
        self.has_hi = int(g.ghost_hi) > 0
        self.radius = program.radius if program.kind == _lib.STAR else 1
        # planes this rank updates (stencil.h:690)
        self.z_lo = int(g.ghost_lo) if self.has_lo else self.radius
        self.z_hi = self.local - int(g.ghost_hi) if self.has_hi else self.local - self.radius

    def _exchange_ops(self, name):
        """P2P ops moving this cube's boundary planes into the neighbours' ghost planes
        (commCubeHalo, stencil.h:639-647).  Built once per cube: the plane views never change, and
        rebuilding them costs 10 us of the ~100 us the host spends per exchange."""
        cache = self.__dict__.setdefault("_ops", {})
        if name in cache:
            return cache[name]
        M, ops = self.M, []
        if self.has_lo:
            ops.append(dist.P2POp(dist.isend, self.b.planes(name, M, M), self.rank - 1, self.group))
            ops.append(dist.P2POp(dist.irecv, self.b.planes(name, 0, M), self.rank - 1, self.group))
        if self.has_hi:
            ops.append(dist.P2POp(dist.isend, self.b.planes(name, self.local - 2 * M, M), self.rank + 1, self.group))
            ops.append(dist.P2POp(dist.irecv, self.b.planes(name, self.local - M, M), self.rank + 1, self.group))
        cache[name] = ops
        return ops

    def _start_exchange(self, name):
        if self.nprocs <= 1 or self.exchange == "none":
            return []
        ops = self._exchange_ops(name)
        if not ops:
            return []
        if self.exchange == "device":
            return dist.batch_isend_irecv(ops)
        # host staging: device -> CPU, P2P on CPU tensors, CPU -> device on wait
        staged, cpu_ops = [], []
        for op in ops:
            t = op.tensor
            h = t.cpu() if op.op is dist.isend else torch.empty(t.shape, dtype=t.dtype)
            cpu_ops.append(dist.P2POp(op.op, h, op.peer, op.group))
            if op.op is dist.irecv:
                staged.append((t, h))
        reqs = dist.batch_isend_irecv(cpu_ops)

        class _Staged:
            def wait(self_inner):
                for r in reqs:
                    r.wait()
                for t, h in staged:
                    t.copy_(h)
        return [_Staged()]

    def _split(self, sweep):
        """One update of the owned planes as boundary-first launches: `sweep(z_lo, z_hi, side)` is
        called for the ghost-depth planes next to each neighbour, then the exchange of those planes
        starts behind them, then the interior runs beside the exchange."""
        G = self.M
        lo, hi = self.z_lo, self.z_hi
        ilo, ihi = lo, hi
        b = self.b
        two = self.side_stream and (self.has_lo or self.has_hi) and hasattr(b, "fork_side")
        if two:
            b.fork_side()
        if self.has_lo:
            ilo = min(lo + G, hi)
            sweep(lo, ilo, two)
        if self.has_hi:
            ihi = max(hi - G, ilo)
            sweep(ihi, hi, two)
        if two:
            with b.side_ctx():
                reqs = self._start_exchange(self._out)
        else:
            reqs = self._start_exchange(self._out)
        sweep(ilo, ihi, False)
        if two:
            b.join_side()
        for r in reqs:
            r.wait()

    def step(self, out, inp):
        """out = f(inp) on the owned planes, then out's ghost planes = neighbours' new boundary planes."""
        self._out = out
        self._split(lambda z0, z1, side: self.b.sweep(out, inp, self.prog, z0, z1, **({"side": True} if side else {})))

    def hop(self, out, inp, mid, k):
        """k fused steps out = f^k(inp) and one exchange of the ghost-depth boundary planes of
        `out`: 1/k of the launches and messages (same bytes) of k single steps."""
        self._out = out
        self._split(lambda z0, z1, side: self.b.sweep_fused(out, inp, mid, self.prog, k, z0, z1, **({"side": True} if side else {})))

    def fusable(self):
        """Steps per launch this slab can fuse: the deepest fused kernel the backend has for the
        program, limited towards a neighbour by the ghost depth (k steps of radius r need k*r ghost
        planes: context created with M >= 2 or 3 for a radius-1 program, 4 for radius 2); 0 if none."""
        can = getattr(self.b, "can_fuse", None)
        k = int(can(self.prog)) if can else 0
        if self.has_lo or self.has_hi:
            k = min(k, self.M // self.radius)
        return k if k >= 2 else 0

    def run(self, a, b, steps, fuse=True):
        """`steps` ping-pong steps: step 0 writes a = f(b) (fdtd1.cu:68-80); level t lives in b for
        even t, in a for odd t.  Where a fused kernel exists the loop advances three steps per
        launch straight between the two cubes (level t+3 belongs in the other cube, whose shell is
        also the one levels t+1 and t+3 see), or, with ghost depth 2, two steps per launch b -> c ->
        b (c: a third cube carrying b's shell, a lends its shell to the skipped levels).  It always
        ends with 1-3 single steps, so that a and b finish holding the two newest levels exactly as
        after the step-by-step loop."""
        done = 0
        k = self.fusable() if (fuse and steps >= 4) else 0
        if k == 3:
            for _ in range((steps - 1) // 3):
                out, inp = (a, b) if done % 2 == 0 else (b, a)
                self.hop(out, inp, out, 3)
                done += 3
        elif k == 2:
            pairs = steps // 2 - 1 if steps % 2 == 0 else ((steps - 1) // 2) & ~1
            if pairs > 0:
                c = "__fused_scratch"
                if c not in self.b.bufs:
                    self.b.alloc(c)
                self.b.copy_cube(c, b)
                cur, oth = b, c
                for _ in range(pairs):
                    self.hop(oth, cur, a, 2)
                    cur, oth = oth, cur
                done = 2 * pairs
                if cur != b:
                    self.step(a, cur)
                    done += 1
        for t in range(done, steps):
            if t % 2 == 0:
                self.step(a, b)
            else:
                self.step(b, a)
        return (a, b) if steps % 2 == 1 else (b, a)  # (newest, previous)


class CompositeRunner:
    """The time loop of main.cu:149-171 for one rank: per step
        s0 = composite(s1, vel, s0); inject(s0); filt(s0); sum += s0*s1; exchange s0; swap
    with the reference's choreography (boundary side buffers first, main.cu:151-158, then the whole
    slab, :160-163, then the exchange of both fields, :164-168) replaced by: all operations on the
    ghost-depth planes next to each neighbour, start their exchange, all operations on the remaining
    planes beside it.  Only the field just written is exchanged: the other one received its ghost
    planes when it was written a step earlier."""

    def __init__(self, backend, M, comp, select, scale, rank, nprocs, group=None, exchange="device"):
        self.b, self.M, self.comp = backend, M, comp
        self.select, self.scale = select, scale  # filt: (zsel, ysel, xsel) in global coordinates, factor
        self.x = SlabRunner(backend, M, _lib.make_program("star", radius=comp.radius), rank, nprocs, group, exchange)

    def step(self, s0, s1, vel, total):
        b, M, x = self.b, self.M, self.x
        g = b.g
        lo, hi = x.z_lo, x.z_hi                                    # planes the update writes
        olo, ohi = int(g.ghost_lo), int(g.ghost_lo) + int(g.owned_planes)  # planes inject/filt touch
        cuts = []
        if x.has_lo:
            cuts.append((olo, min(olo + M, ohi)))
        if x.has_hi:
            cuts.append((max(ohi - M, cuts[-1][1] if cuts else olo), ohi))
        rest_lo = cuts[0][1] if x.has_lo else olo
        rest_hi = cuts[-1][0] if x.has_hi else ohi
        order = self.comp.order

        def all_ops(z0, z1):
            b.comp_sweep(s0, s1, vel, self.comp, max(z0, lo), min(z1, hi))
            b.inject(s0, 1.0, z0, z1)
            b.filt(s0, self.select, self.scale, order, z0, z1)

        for z0, z1 in cuts:
            all_ops(z0, z1)
        reqs = x._start_exchange(s0)
        all_ops(rest_lo, rest_hi)
        b.mul_acc(total, s0, s1, self.comp.radius, order, lo, hi)
        for r in reqs:
            r.wait()

    def run(self, p0, p1, vel, total, steps):
        s0, s1 = p0, p1
        for _ in range(steps):
            self.step(s0, s1, vel, total)
            s0, s1 = s1, s0
        return s0, s1  # as main.cu leaves them after its swaps


class YeeRunner:
    """EXTENSION (no counterpart in the reference): the six-field staggered-grid step on z-slabs with one ghost
    plane per side.  A step is two half steps, and each half reads z differences of two fields of the other
    triple in ONE direction only:
        H half (which 0)  H -= ch curl E, forward differences:  reads Ex, Ey of plane z+1
        E half (which 1)  E += ce curl H, backward differences: reads Hx, Hy of plane z-1
    so after the H half a rank's last owned plane of Hx, Hy goes up (into the upper neighbour's low ghost plane), and
    after the E half its first owned plane of Ex, Ey goes down: two planes per half step and neighbour instead of the
    six a both-ways exchange of the whole triple would move.  Ez and Hz are never differenced along z; their ghost
    planes are not kept current.  As in SlabRunner, the plane that is sent is computed first and travels beside the
    sweep of the remaining planes.

    exchange: "device" / "host" / "none" as SlabRunner; "native": split and exchange inside the library, over its own
    communicator (half_begin / half_end; slabs of one process call begin on every slab, then end on every slab);
    "local": the neighbours are backends of this process
    (`peers[rank]`, cubes of the same names on the same GPU or on peers with access) and planes move by device copies."""

    def __init__(self, backend, names, rank, nprocs, group=None, exchange="device", order=_lib.ORDER_CHAIN, peers=None):
        if len(names) != 6:
            raise ValueError("six cube names: ex, ey, ez, hx, hy, hz")
        self.b, self.names, self.order = backend, tuple(names), order
        self.rank, self.nprocs, self.group, self.exchange, self.peers = rank, nprocs, group, exchange, peers or {}
        g = backend.g
        if (int(g.ghost_lo), int(g.ghost_hi)) not in ((0, 0), (1, 0), (0, 1), (1, 1)):
            raise ValueError("the staggered-grid step runs on slabs with one ghost plane per neighbour (context created with M = 1)")
        self.local = int(g.local_planes)
        self.has_lo, self.has_hi = int(g.ghost_lo) > 0, int(g.ghost_hi) > 0
        self.z_lo = 1                   # first plane updated: the first owned one, or plane 1 of the grid (shell untouched)
        self.z_hi = self.local - 1
        self._ops = {}

    def _moves(self, which):
        """(cube, plane sent, to rank, ghost plane received, from rank) of one half step; None where there is no neighbour."""
        if which == 0:
            cubes, send, dst, recv, src = self.names[3:5], self.local - 2, self.rank + 1, 0, self.rank - 1
            do_send, do_recv = self.has_hi, self.has_lo
        else:
            cubes, send, dst, recv, src = self.names[0:2], 1, self.rank - 1, self.local - 1, self.rank + 1
            do_send, do_recv = self.has_lo, self.has_hi
        return cubes, (send, dst) if do_send else None, (recv, src) if do_recv else None

    def _start_exchange(self, which):
        if self.exchange == "none":
            return []
        cubes, snd, rcv = self._moves(which)
        if self.exchange == "local":
            if snd:
                peer = self.peers[snd[1]]
                ghost = 0 if which == 0 else int(peer.g.local_planes) - 1
                for n in cubes:
                    peer.planes(n, ghost, 1).copy_(self.b.planes(n, snd[0], 1))
            return []
        if which not in self._ops:
            ops = []
            for n in cubes:
                if snd:
                    ops.append(dist.P2POp(dist.isend, self.b.planes(n, snd[0], 1), snd[1], self.group))
                if rcv:
                    ops.append(dist.P2POp(dist.irecv, self.b.planes(n, rcv[0], 1), rcv[1], self.group))
            self._ops[which] = ops
        ops = self._ops[which]
        if not ops:
            return []
        if self.exchange == "device":
            return dist.batch_isend_irecv(ops)
        staged, cpu_ops = [], []
        for op in ops:
            t = op.tensor
            h = t.cpu() if op.op is dist.isend else torch.empty(t.shape, dtype=t.dtype)
            cpu_ops.append(dist.P2POp(op.op, h, op.peer, op.group))
            if op.op is dist.irecv:
                staged.append((t, h))
        reqs = dist.batch_isend_irecv(cpu_ops)

        class _Staged:
            def wait(self_inner):
                for r in reqs:
                    r.wait()
                for t, h in staged:
                    t.copy_(h)
        return [_Staged()]

    def half_begin(self, which, coef):
        """exchange="native": the library's own split and exchange (fdtdx_yee3d_slab_half_begin; the context has a
        communicator: Stencil.commInit, or commInitLocal for slabs of one process)."""
        self.b.st.yee3dSlabHalfBegin(self.names, which, coef, self.order, torch.cuda.current_stream().cuda_stream)

    def half_end(self):
        self.b.st.stepEnd(torch.cuda.current_stream().cuda_stream)

    def half(self, which, coef):
        if self.exchange == "native":
            self.half_begin(which, coef)
            self.half_end()
            return
        lo, hi = self.z_lo, self.z_hi
        _, snd, _ = self._moves(which)
        if snd and hi > lo:
            z = snd[0]                  # the plane the neighbour waits for: first, so that it travels beside the rest
            self.b.yee_half(self.names, which, coef, self.order, z, z + 1)
            reqs = self._start_exchange(which)
            if z == lo:
                self.b.yee_half(self.names, which, coef, self.order, lo + 1, hi)
            else:
                self.b.yee_half(self.names, which, coef, self.order, lo, hi - 1)
        else:
            self.b.yee_half(self.names, which, coef, self.order, lo, hi)
            reqs = self._start_exchange(which)
        for r in reqs:
            r.wait()

    def step(self, ce, ch):
        self.half(0, ch)
        self.half(1, ce)

    def run(self, steps, ce, ch):
        for _ in range(steps):
            self.step(ce, ch)


def gather_cube(backend, name, Z, Y, X, rank, nprocs, group=None):
    """transferCubeToCPU's cross-rank part (stencil.h:623-636): every rank ends with the full
    unpadded [Z][Y][X] grid.  HIP backend only."""
    owned = torch.from_numpy(backend.st.downloadOwned(name))
    if nprocs == 1:
        return owned.numpy()
    parts = [None] * nprocs
    dist.all_gather_object(parts, owned.numpy(), group=group)
    import numpy as np
    return np.concatenate(parts, axis=0)
