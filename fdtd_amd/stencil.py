"""Python face of the reference's `Stencil<DataType,Z,Y,X,M,...>` host class (stencil.h:357-842).

Method names and argument meaning follow the reference so that driver code reads the same
(`mallocCube`, `transferCubeToGPU`, `propagate`, `transferCubeToCPU`, ...).  Two differences
that a C ABI forces:
  * the per-cell device lambda becomes a `Program` picked from the closed set the reference
    ships (fdtd_amd.lib.make_program);
  * API misuse raises FdtdxError instead of tripping an assert (stencil.h:443,470,555,...).
All work is done by libfdtdx.so through ctypes; nothing here computes.
"""
import ctypes

import numpy as np

from . import lib as _lib


class Stencil:
    def __init__(self, Z, Y, X, M, rank=0, nprocs=1, device=-1, pitch_x=0):
        """stencil.h:385-413.  rank/nprocs replace MPI::COMM_WORLD rank/size."""
        self._L = _lib.load()
        self.Z, self.Y, self.X, self.M = Z, Y, X, M
        cfg = _lib.Config(Z, Y, X, M, rank, nprocs, device, pitch_x, 0)
        h = ctypes.c_void_p()
        _lib.check(self._L.fdtdx_create(ctypes.byref(cfg), ctypes.byref(h)))
        self._h = h
        g = _lib.Geometry()
        _lib.check(self._L.fdtdx_get_geometry(self._h, ctypes.byref(g)))
        self.geometry = g

    # -- lifetime ------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.fdtdx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- stencil.h:420-438 ---------------------------------------------------------------
    def getRank(self):
        return self._L.fdtdx_rank(self._h)

    def getProcs(self):
        return self._L.fdtdx_nprocs(self._h)

    def sync(self):
        _lib.check(self._L.fdtdx_sync(self._h))

    # -- pool: stencil.h:553-582, 468-477 -----------------------------------------------
    def mallocCube(self, name, create_halo=False):
        """create_halo is accepted for source compatibility; ghost planes live inside the cube
        (boundary planes are received straight into them, so the side buffers of stencil.h:560-574 are not needed)."""
        _lib.check(self._L.fdtdx_malloc_cube(self._h, name.encode()))

    def attachCube(self, name, device_ptr, nbytes):
        _lib.check(self._L.fdtdx_attach_cube(self._h, name.encode(), ctypes.c_void_p(device_ptr), nbytes))

    def free(self, name):
        _lib.check(self._L.fdtdx_free(self._h, name.encode()))

    def getCube(self, name):
        p = ctypes.c_void_p()
        _lib.check(self._L.fdtdx_get_cube(self._h, name.encode(), ctypes.byref(p)))
        return p.value

    # -- transfers: stencil.h:596-637 -----------------------------------------------------
    def transferCubeToGPU(self, name, cpu):
        cpu = np.ascontiguousarray(cpu, np.float32)
        assert cpu.shape == (self.Z, self.Y, self.X), cpu.shape
        _lib.check(self._L.fdtdx_transfer_cube_to_gpu(self._h, name.encode(), cpu.ctypes.data_as(ctypes.c_void_p)))

    def transferCubeToCPU(self, cpu, name):
        assert cpu.dtype == np.float32 and cpu.flags["C_CONTIGUOUS"] and cpu.shape == (self.Z, self.Y, self.X)
        _lib.check(self._L.fdtdx_transfer_cube_to_cpu(self._h, cpu.ctypes.data_as(ctypes.c_void_p), name.encode()))

    def gatherCube(self, cpu, name):
        """transferCubeToCPU with its cross-rank gather (stencil.h:610-637): every rank ends with the whole grid."""
        assert cpu.dtype == np.float32 and cpu.flags["C_CONTIGUOUS"] and cpu.shape == (self.Z, self.Y, self.X)
        _lib.check(self._L.fdtdx_gather_cube(self._h, cpu.ctypes.data_as(ctypes.c_void_p), name.encode()))

    def getAddr(self, name):
        """Device address of an address list (get<size_t>(name), stencil.h:479-484)."""
        p = ctypes.c_void_p()
        _lib.check(self._L.fdtdx_get_addr(self._h, name.encode(), ctypes.byref(p)))
        return p.value

    def uploadLocal(self, name, local):
        local = np.ascontiguousarray(local, np.float32)
        assert local.shape == (self.geometry.local_planes, self.Y, self.X), local.shape
        _lib.check(self._L.fdtdx_upload_local(self._h, name.encode(), local.ctypes.data_as(ctypes.c_void_p)))

    def downloadOwned(self, name):
        out = np.empty((self.geometry.owned_planes, self.Y, self.X), np.float32)
        _lib.check(self._L.fdtdx_download_owned(self._h, out.ctypes.data_as(ctypes.c_void_p), name.encode()))
        return out

    def initMt(self, name_p0, name_p1):
        """fdtd1.cu:21-31 initial fields, generated for this slab's planes."""
        _lib.check(self._L.fdtdx_init_mt(self._h, name_p0.encode(), name_p1.encode()))

    def initJacobi(self, name):
        """jacobi.cu:412-417 initial field."""
        _lib.check(self._L.fdtdx_init_jacobi(self._h, name.encode()))

    # -- sweeps: stencil.h:684-768 ---------------------------------------------------------
    def propagate(self, output, input, program, stream=None, spill=True):
        """One sweep output = f(input).  `spill` (a launch-bounds hint in the reference,
        stencil.h:696-699) is accepted and ignored."""
        _lib.check(self._L.fdtdx_propagate(self._h, output.encode(), input.encode(), ctypes.byref(program),
                                           ctypes.c_void_p(stream or 0)))

    def propagatePlanes(self, output, input, program, z_lo, z_hi, stream=None):
        _lib.check(self._L.fdtdx_propagate_planes(self._h, output.encode(), input.encode(), ctypes.byref(program),
                                                  z_lo, z_hi, ctypes.c_void_p(stream or 0)))

    def propagate2Planes(self, output, input, mid, program, z_lo, z_hi, stream=None):
        """Two fused steps output = f(f(input)) on local planes [z_lo, z_hi) (jacobi.cu:446-454)."""
        _lib.check(self._L.fdtdx_propagate2_planes(self._h, output.encode(), input.encode(), mid.encode(),
                                                   ctypes.byref(program), z_lo, z_hi, ctypes.c_void_p(stream or 0)))

    def propagateFusedPlanes(self, output, input, mid, program, nsteps, z_lo, z_hi, stream=None):
        """nsteps (2 or 3) fused steps output = f^nsteps(input) on local planes [z_lo, z_hi); the
        reference's fused launch is jacobi.cu:446-454 (three steps)."""
        _lib.check(self._L.fdtdx_propagate_fused_planes(self._h, output.encode(), input.encode(), mid.encode(),
                                                        ctypes.byref(program), nsteps, z_lo, z_hi,
                                                        ctypes.c_void_p(stream or 0)))

    def propagateFusedShells(self, output, input, shell_even, shell_odd, program, nsteps, z_lo, z_hi, stream=None):
        """propagateFusedPlanes with both shell sources named: even levels (the input included) take their shell
        cells from `shell_even`, odd levels from `shell_odd`; the shell of `output` is never written."""
        _lib.check(self._L.fdtdx_propagate_fused_shells(self._h, output.encode(), input.encode(), shell_even.encode(),
                                                        shell_odd.encode(), ctypes.byref(program), nsteps, z_lo, z_hi,
                                                        ctypes.c_void_p(stream or 0)))

    def propagateTwoRanges(self, out, inp, program, z_lo, z_hi, z2_lo, z2_hi, stream=None):
        """propagateHaloTopBackup + propagateHaloButtomBackup (stencil.h:770-808) as one launch over two plane ranges."""
        _lib.check(self._L.fdtdx_propagate_two_ranges(self._h, out.encode(), inp.encode(), ctypes.byref(program), z_lo, z_hi, z2_lo, z2_hi,
                                                      ctypes.c_void_p(stream or 0)))

    def propagateFusedTwoRanges(self, out, inp, shell_even, shell_odd, program, nsteps, z_lo, z_hi, z2_lo, z2_hi, stream=None):
        _lib.check(self._L.fdtdx_propagate_fused_two_ranges(self._h, out.encode(), inp.encode(), shell_even.encode(), shell_odd.encode(),
                                                            ctypes.byref(program), nsteps, z_lo, z_hi, z2_lo, z2_hi,
                                                            ctypes.c_void_p(stream or 0)))

    def sweep(self, a, b, program, steps, stream=None, work=False):
        """The drivers' whole time loop on one rank (fdtd1.cu:68-80).  work=False: the library allocates the third cube
        the radius-2 two-step loop hops through on first use; work="name": that cube of the pool is used; work=None:
        no third cube, such programs run single steps (neither of the latter two ever allocates)."""
        if work is False:
            _lib.check(self._L.fdtdx_sweep(self._h, a.encode(), b.encode(), ctypes.byref(program), steps,
                                           ctypes.c_void_p(stream or 0)))
        else:
            _lib.check(self._L.fdtdx_sweep_work(self._h, a.encode(), b.encode(), work.encode() if work else None,
                                                ctypes.byref(program), steps, ctypes.c_void_p(stream or 0)))

    # -- the composite pipeline of main.cu -----------------------------------------------------------
    def _owned(self):
        g = self.geometry
        return int(g.ghost_lo), int(g.ghost_lo) + int(g.owned_planes)

    def propagateComposite(self, output, input, vel, comp, z_lo=None, z_hi=None, stream=None):
        """main.cu:160 (and :153-154 for a plane range): output = composite(input, vel, output)."""
        if z_lo is None:
            _lib.check(self._L.fdtdx_propagate_composite(self._h, output.encode(), input.encode(), vel.encode(),
                                                         ctypes.byref(comp), ctypes.c_void_p(stream or 0)))
        else:
            _lib.check(self._L.fdtdx_propagate_composite_planes(self._h, output.encode(), input.encode(), vel.encode(),
                                                                ctypes.byref(comp), z_lo, z_hi, ctypes.c_void_p(stream or 0)))

    def compositeStep(self, output, input, vel, sum_name, addr_name, comp, add, zsel, ysel, xsel, scale, stream=None):
        """One whole step of main.cu:160-163 (update, inject, filt, mul) with the product-accumulate
        folded into the update kernel; same results as the four separate calls."""
        _lib.check(self._L.fdtdx_composite_step(self._h, output.encode(), input.encode(), vel.encode(), sum_name.encode(),
                                                addr_name.encode(), ctypes.byref(comp), add, zsel, ysel, xsel, scale,
                                                ctypes.c_void_p(stream or 0)))

    def compositeStepStatic(self, output, input, vel, sum_name, addr_name, comp, add, zsel, ysel, xsel, scale, stream=None):
        """compositeStep without any device allocation (scratch in a static device array; at most
        lib.STATIC_ADDR_MAX listed addresses on this rank)."""
        _lib.check(self._L.fdtdx_composite_step_static(self._h, output.encode(), input.encode(), vel.encode(), sum_name.encode(),
                                                       addr_name.encode(), ctypes.byref(comp), add, zsel, ysel, xsel, scale,
                                                       ctypes.c_void_p(stream or 0)))

    def mallocAddr(self, name, capacity):
        """malloc<size_t>(name, n, true), stencil.h:446-466."""
        _lib.check(self._L.fdtdx_malloc_addr(self._h, name.encode(), capacity))

    def transferAddrToGPU(self, name, global_addr):
        """transferToGPU(name, addr, n), stencil.h:518-544: linear indices into the undecomposed [Z][Y][X] grid."""
        a = np.ascontiguousarray(global_addr, np.uint64)
        _lib.check(self._L.fdtdx_transfer_addr_to_gpu(self._h, name.encode(), a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), len(a)))

    def getAddrSize(self, name):
        n = ctypes.c_uint64()
        _lib.check(self._L.fdtdx_addr_size(self._h, name.encode(), ctypes.byref(n)))
        return n.value

    def inject(self, cube, addr_name, add=1.0, z_lo=None, z_hi=None, stream=None):
        """inject(n, inject_kernel, cube, addr, add), main.cu:161 (plane range: :155-156)."""
        lo, hi = self._owned() if z_lo is None else (z_lo, z_hi)
        _lib.check(self._L.fdtdx_inject(self._h, cube.encode(), addr_name.encode(), add, lo, hi, ctypes.c_void_p(stream or 0)))

    def filt(self, cube, zsel, ysel, xsel, scale=0.9, order=_lib.ORDER_CHAIN, z_lo=None, z_hi=None, stream=None):
        """filt(filt_kernel, cube), main.cu:162 (plane range: :157-158)."""
        lo, hi = self._owned() if z_lo is None else (z_lo, z_hi)
        _lib.check(self._L.fdtdx_filt_planes(self._h, cube.encode(), zsel, ysel, xsel, scale, order, lo, hi,
                                             ctypes.c_void_p(stream or 0)))

    def mulAcc(self, sum_name, a, b, radius, order=_lib.ORDER_CHAIN, z_lo=None, z_hi=None, stream=None):
        """filt(mul_kernel, sum, a, b), main.cu:163: sum += a*b on the interior."""
        g = self.geometry
        if z_lo is None:
            z_lo = int(g.ghost_lo) if g.ghost_lo else radius
            z_hi = int(g.local_planes) - (int(g.ghost_hi) if g.ghost_hi else radius)
        _lib.check(self._L.fdtdx_mul_acc(self._h, sum_name.encode(), a.encode(), b.encode(), radius, order, z_lo, z_hi,
                                         ctypes.c_void_p(stream or 0)))

    # -- extension: 3-D Yee E/H update (no counterpart in the reference) ---------------------------------
    @staticmethod
    def _names(fields):
        assert len(fields) == 6
        return (ctypes.c_char_p * 6)(*[f.encode() for f in fields])

    def yee3dHalf(self, fields, which, coef, order=_lib.ORDER_CHAIN, z_lo=None, z_hi=None, stream=None):
        """fields = (Ex, Ey, Ez, Hx, Hy, Hz) cube names; which 0: H -= coef*curl E, 1: E += coef*curl H."""
        if z_lo is None:
            z_lo, z_hi = 1, int(self.geometry.local_planes) - 1
        _lib.check(self._L.fdtdx_yee3d_half(self._h, self._names(fields), which, coef, order, z_lo, z_hi, ctypes.c_void_p(stream or 0)))

    def yee3dSlabHalfBegin(self, fields, which, coef, order=_lib.ORDER_CHAIN, stream=None):
        """One half step on a slab with neighbours: the plane a neighbour needs first, its exchange, the rest beside it;
        stepEnd() waits for the planes received (slabs of one process: begin on all, then end on all)."""
        _lib.check(self._L.fdtdx_yee3d_slab_half_begin(self._h, self._names(fields), which, coef, order, ctypes.c_void_p(stream or 0)))

    def yee3dStep(self, fields, ce, ch, order=_lib.ORDER_CHAIN, stream=None):
        _lib.check(self._L.fdtdx_yee3d_step(self._h, self._names(fields), ce, ch, order, ctypes.c_void_p(stream or 0)))

    def yee3dStepFused(self, before, after, ce, ch, order=_lib.ORDER_CHAIN, stream=None):
        """One launch for the whole step, from the six cubes `before` into the six cubes `after` (which must
        already hold the shell); equal to yee3dStep bit for bit."""
        _lib.check(self._L.fdtdx_yee3d_step_fused(self._h, self._names(before), self._names(after), ce, ch, order,
                                                   ctypes.c_void_p(stream or 0)))

    # -- halo regions: stencil.h:639-647 --------------------------------------------------
    def haloRegion(self, name, side, what):
        """(device pointer, float count) of the planes to send (what=0) / receive (what=1)
        on side 0 (rank-1) or 1 (rank+1); (None, 0) where there is no neighbour."""
        p = ctypes.c_void_p()
        n = ctypes.c_size_t()
        _lib.check(self._L.fdtdx_halo_region(self._h, name.encode(), side, what, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    # -- exchange with the z neighbours and the boundary-first step: stencil.h:639-682, fdtd1.cu:70-78 ------------
    @staticmethod
    def commUniqueId():
        """128 bytes naming a new RCCL communicator: made on rank 0, handed to every rank by the launcher."""
        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.load().fdtdx_comm_unique_id(buf))
        return buf.raw

    def commInit(self, comm_id, comm_rank=None, comm_size=None, peer_lo=None, peer_hi=None):
        """Join the RCCL communicator `comm_id` (one process per slab); ranks default to the slab's own."""
        buf = ctypes.create_string_buffer(bytes(comm_id), 128)
        if comm_rank is None:
            _lib.check(self._L.fdtdx_comm_init(self._h, buf))
        else:
            _lib.check(self._L.fdtdx_comm_init_ranks(self._h, buf, comm_rank, comm_size, peer_lo, peer_hi))

    @staticmethod
    def commInitLocal(stencils):
        """Link the slabs of ranks 0..n-1, all owned by this process (copy-engine transfers instead of RCCL)."""
        arr = (ctypes.c_void_p * len(stencils))(*[s._h for s in stencils])
        _lib.check(_lib.load().fdtdx_comm_init_local(arr, len(stencils)))

    def commIpcExport(self, names):
        """IPC transport, step 1 (fdtdx_comm_ipc_export): the descriptor of this rank's cubes `names`, as bytes for the launcher
        to hand to the neighbouring ranks."""
        arr = (ctypes.c_char_p * len(names))(*[n.encode() for n in names])
        buf = ctypes.create_string_buffer(_lib.IPC_DESC_BYTES)
        _lib.check(self._L.fdtdx_comm_ipc_export(self._h, arr, len(names), buf))
        return buf.raw

    def commIpcConnect(self, lo, hi):
        """IPC transport, step 2 (fdtdx_comm_ipc_connect): the descriptors (bytes) of the neighbours below / above, None where
        there is none."""
        keep = [ctypes.create_string_buffer(bytes(d), _lib.IPC_DESC_BYTES) if d is not None else None for d in (lo, hi)]
        _lib.check(self._L.fdtdx_comm_ipc_connect(self._h, keep[0], keep[1]))

    def commDestroy(self):
        """Drop this slab's communicator (fdtdx_comm_destroy); another transport can be set up afterwards."""
        _lib.check(self._L.fdtdx_comm_destroy(self._h))

    def commState(self):
        """Which exchange events have fired / which streams are idle, as text (fdtdx_comm_state; queries only)."""
        buf = ctypes.create_string_buffer(640)
        _lib.check(self._L.fdtdx_comm_state(self._h, buf, 640))
        return buf.value.decode()

    def barrier(self):
        """stencil.h:435-438."""
        _lib.check(self._L.fdtdx_comm_barrier(self._h))

    def commBoundaryDone(self, stream=None):
        _lib.check(self._L.fdtdx_comm_boundary_done(self._h, ctypes.c_void_p(stream or 0)))

    def commExchangeStart(self, name, stream=None):
        _lib.check(self._L.fdtdx_comm_exchange_start(self._h, name.encode(), ctypes.c_void_p(stream or 0)))

    def commExchangeWait(self, stream=None):
        _lib.check(self._L.fdtdx_comm_exchange_wait(self._h, ctypes.c_void_p(stream or 0)))

    def commCubeHalo(self, name, stream=None):
        """commCubeHalo (stencil.h:639-647): stream-ordered, not host-blocking."""
        _lib.check(self._L.fdtdx_comm_cube_halo(self._h, name.encode(), ctypes.c_void_p(stream or 0)))

    def stepBegin(self, output, input, program, stream=None):
        _lib.check(self._L.fdtdx_step_begin(self._h, output.encode(), input.encode(), ctypes.byref(program), ctypes.c_void_p(stream or 0)))

    def hopBegin(self, output, input, shell_even, shell_odd, program, nsteps, stream=None):
        _lib.check(self._L.fdtdx_hop_begin(self._h, output.encode(), input.encode(), shell_even.encode(), shell_odd.encode(),
                                           ctypes.byref(program), nsteps, ctypes.c_void_p(stream or 0)))

    def stepEnd(self, stream=None):
        _lib.check(self._L.fdtdx_step_end(self._h, ctypes.c_void_p(stream or 0)))

    @staticmethod
    def run(stencils, a, b, program, steps, streams=None, work=None):
        """The drivers' whole loop (fdtd1.cu:68-80) on the slabs this process owns (one with RCCL, all in local mode).
        work: name of a third cube in every slab's pool, for the two-step loops that hop through one (radius-2 star)."""
        n = len(stencils)
        arr = (ctypes.c_void_p * n)(*[s._h for s in stencils])
        st = (ctypes.c_void_p * n)(*[ctypes.c_void_p(x or 0) for x in streams]) if streams else None
        _lib.check(_lib.load().fdtdx_run_work(arr, n, a.encode(), b.encode(), work.encode() if work else None, ctypes.byref(program), steps, st))

    # -- tuning ------------------------------------------------------------------------------
    def setOption(self, key, value):
        _lib.check(self._L.fdtdx_set_option(self._h, key.encode(), int(value)))

    def fusedDepth(self, program, named_shells=False):
        """Deepest fused launch (time steps) the library has for the program on this grid: 3, 2 or 0 (fdtdx_fused_depth)."""
        k = self._L.fdtdx_fused_depth(self._h, ctypes.byref(program), 1 if named_shells else 0)
        if k < 0:
            _lib.check(-k)
        return k

    def lastLaunchInfo(self):
        buf = ctypes.create_string_buffer(640)
        _lib.check(self._L.fdtdx_last_launch_info(self._h, buf, 640))
        return buf.value.decode()
