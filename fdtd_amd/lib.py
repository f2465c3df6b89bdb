"""ctypes binding of libfdtdx.so (include/fdtdx.h).

This is the only way Python reaches the sweeps: there is no CPU or eager fallback.  If the
shared library has not been built (fdtd_amd/csrc/Makefile, or __graft_entry__.build()) every
entry point raises FdtdxError.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FDTDX_LIB names another build of the same library (tuning experiments)
LIB_PATH = os.environ.get("FDTDX_LIB") or os.path.join(_HERE, "libfdtdx.so")

STATIC_ADDR_MAX = 262144  # FDTDX_STATIC_ADDR_MAX
OK, ERR_INVALID, ERR_NOTFOUND, ERR_EXISTS, ERR_HIP, ERR_UNSUPPORTED, ERR_NOMEM = range(7)
STAR, SPC13, JACOBI7 = 0, 1, 2
ORDER_CHAIN, ORDER_GOLDEN = 0, 1


class FdtdxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"fdtdx error {code}: {msg}")
        self.code = code


class Config(ctypes.Structure):
    _fields_ = [("Z", ctypes.c_uint64), ("Y", ctypes.c_uint64), ("X", ctypes.c_uint64),
                ("M", ctypes.c_uint32), ("rank", ctypes.c_uint32), ("nprocs", ctypes.c_uint32),
                ("device", ctypes.c_int32), ("pitch_x", ctypes.c_uint32), ("flags", ctypes.c_uint32)]


class Geometry(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in
                ("pitch_x", "plane", "owned_begin", "owned_planes", "ghost_lo", "ghost_hi",
                 "local_planes", "first_global", "update_lo", "update_hi", "cube_elems")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class Program(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_uint32), ("radius", ctypes.c_uint32), ("order", ctypes.c_uint32),
                ("reserved", ctypes.c_uint32), ("coef", ctypes.c_float * 16)]


class Composite(ctypes.Structure):
    """fdtdx_composite: the update of main.cu:102-119 / :55-72."""
    _fields_ = [("radius", ctypes.c_uint32), ("order", ctypes.c_uint32)] + [(n, ctypes.c_double) for n in
                ("cz", "cy", "cx", "scal_num", "scal_den", "wx", "wy", "wz")]


# every symbol include/fdtdx.h declares, with its signature
_P = ctypes.POINTER
_c = ctypes.c_char_p
_vp = ctypes.c_void_p
_fp = _P(ctypes.c_float)
SYMBOLS = {
    "fdtdx_decompose": (ctypes.c_int, [_P(Config), _P(Geometry)]),
    "fdtdx_create": (ctypes.c_int, [_P(Config), _P(_vp)]),
    "fdtdx_destroy": (ctypes.c_int, [_vp]),
    "fdtdx_last_error": (ctypes.c_char_p, []),
    "fdtdx_get_geometry": (ctypes.c_int, [_vp, _P(Geometry)]),
    "fdtdx_rank": (ctypes.c_int, [_vp]),
    "fdtdx_nprocs": (ctypes.c_int, [_vp]),
    "fdtdx_sync": (ctypes.c_int, [_vp]),
    "fdtdx_malloc_cube": (ctypes.c_int, [_vp, _c]),
    "fdtdx_attach_cube": (ctypes.c_int, [_vp, _c, _vp, ctypes.c_size_t]),
    "fdtdx_free": (ctypes.c_int, [_vp, _c]),
    "fdtdx_get_cube": (ctypes.c_int, [_vp, _c, _P(_vp)]),
    "fdtdx_transfer_cube_to_gpu": (ctypes.c_int, [_vp, _c, _vp]),
    "fdtdx_transfer_cube_to_cpu": (ctypes.c_int, [_vp, _vp, _c]),
    "fdtdx_upload_local": (ctypes.c_int, [_vp, _c, _vp]),
    "fdtdx_download_owned": (ctypes.c_int, [_vp, _vp, _c]),
    "fdtdx_init_mt": (ctypes.c_int, [_vp, _c, _c]),
    "fdtdx_init_jacobi": (ctypes.c_int, [_vp, _c]),
    "fdtdx_propagate": (ctypes.c_int, [_vp, _c, _c, _P(Program), _vp]),
    "fdtdx_propagate_planes": (ctypes.c_int, [_vp, _c, _c, _P(Program), ctypes.c_uint64, ctypes.c_uint64, _vp]),
    "fdtdx_propagate_ptr": (ctypes.c_int, [_vp, _vp, _vp, _P(Program), ctypes.c_uint64, ctypes.c_uint64, _vp]),
    "fdtdx_propagate_fused_planes": (ctypes.c_int, [_vp, _c, _c, _c, _P(Program), ctypes.c_uint32, ctypes.c_uint64,
                                                    ctypes.c_uint64, _vp]),
    "fdtdx_propagate_fused_shells": (ctypes.c_int, [_vp, _c, _c, _c, _c, _P(Program), ctypes.c_uint32, ctypes.c_uint64,
                                                    ctypes.c_uint64, _vp]),
    "fdtdx_propagate2_planes": (ctypes.c_int, [_vp, _c, _c, _c, _P(Program), ctypes.c_uint64, ctypes.c_uint64, _vp]),
    "fdtdx_propagate_two_ranges": (ctypes.c_int, [_vp, _c, _c, _P(Program), ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, _vp]),
    "fdtdx_propagate_fused_two_ranges": (ctypes.c_int, [_vp, _c, _c, _c, _c, _P(Program), ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64,
                                                        ctypes.c_uint64, ctypes.c_uint64, _vp]),
    "fdtdx_sweep": (ctypes.c_int, [_vp, _c, _c, _P(Program), ctypes.c_uint32, _vp]),
    "fdtdx_sweep_work": (ctypes.c_int, [_vp, _c, _c, _c, _P(Program), ctypes.c_uint32, _vp]),
    "fdtdx_propagate_composite_planes": (ctypes.c_int, [_vp, _c, _c, _c, _P(Composite), ctypes.c_uint64, ctypes.c_uint64, _vp]),
    "fdtdx_propagate_composite": (ctypes.c_int, [_vp, _c, _c, _c, _P(Composite), _vp]),
    "fdtdx_composite_step": (ctypes.c_int, [_vp, _c, _c, _c, _c, _c, _P(Composite), ctypes.c_float, ctypes.c_int64, ctypes.c_int64,
                                            ctypes.c_int64, ctypes.c_double, _vp]),
    "fdtdx_composite_step_static": (ctypes.c_int, [_vp, _c, _c, _c, _c, _c, _P(Composite), ctypes.c_float, ctypes.c_int64, ctypes.c_int64,
                                                   ctypes.c_int64, ctypes.c_double, _vp]),
    "fdtdx_malloc_addr": (ctypes.c_int, [_vp, _c, ctypes.c_uint64]),
    "fdtdx_transfer_addr_to_gpu": (ctypes.c_int, [_vp, _c, _P(ctypes.c_uint64), ctypes.c_uint64]),
    "fdtdx_addr_size": (ctypes.c_int, [_vp, _c, _P(ctypes.c_uint64)]),
    "fdtdx_inject": (ctypes.c_int, [_vp, _c, _c, ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64, _vp]),
    "fdtdx_filt_planes": (ctypes.c_int, [_vp, _c, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_double, ctypes.c_uint32,
                                         ctypes.c_uint64, ctypes.c_uint64, _vp]),
    "fdtdx_mul_acc": (ctypes.c_int, [_vp, _c, _c, _c, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64, _vp]),
    "fdtdx_yee3d_half": (ctypes.c_int, [_vp, _P(_c), ctypes.c_int, ctypes.c_float, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64, _vp]),
    "fdtdx_yee3d_step": (ctypes.c_int, [_vp, _P(_c), ctypes.c_float, ctypes.c_float, ctypes.c_uint32, _vp]),
    "fdtdx_yee3d_step_fused": (ctypes.c_int, [_vp, _P(_c), _P(_c), ctypes.c_float, ctypes.c_float, ctypes.c_uint32, _vp]),
    "fdtdx_yee2d_steps": (ctypes.c_int, [ctypes.c_int, _vp, _vp, _vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_float, ctypes.c_float, ctypes.c_uint32, _vp]),
    "fdtdx_halo_region": (ctypes.c_int, [_vp, _c, ctypes.c_int, ctypes.c_int, _P(_vp), _P(ctypes.c_size_t)]),
    "fdtdx_comm_unique_id": (ctypes.c_int, [_vp]),
    "fdtdx_comm_init": (ctypes.c_int, [_vp, _vp]),
    "fdtdx_comm_init_ranks": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "fdtdx_comm_init_local": (ctypes.c_int, [_P(_vp), ctypes.c_int]),
    "fdtdx_comm_destroy": (ctypes.c_int, [_vp]),
    "fdtdx_comm_ipc_export": (ctypes.c_int, [_vp, _P(_c), ctypes.c_int, _vp]),
    "fdtdx_comm_ipc_connect": (ctypes.c_int, [_vp, _vp, _vp]),
    "fdtdx_comm_barrier": (ctypes.c_int, [_vp]),
    "fdtdx_comm_state": (ctypes.c_int, [_vp, _c, ctypes.c_size_t]),
    "fdtdx_comm_boundary_done": (ctypes.c_int, [_vp, _vp]),
    "fdtdx_comm_exchange_start": (ctypes.c_int, [_vp, _c, _vp]),
    "fdtdx_comm_exchange_wait": (ctypes.c_int, [_vp, _vp]),
    "fdtdx_comm_cube_halo": (ctypes.c_int, [_vp, _c, _vp]),
    "fdtdx_step_begin": (ctypes.c_int, [_vp, _c, _c, _P(Program), _vp]),
    "fdtdx_hop_begin": (ctypes.c_int, [_vp, _c, _c, _c, _c, _P(Program), ctypes.c_uint32, _vp]),
    "fdtdx_step_end": (ctypes.c_int, [_vp, _vp]),
    "fdtdx_yee3d_slab_half_begin": (ctypes.c_int, [_vp, _P(_c), ctypes.c_int, ctypes.c_float, ctypes.c_uint32, _vp]),
    "fdtdx_run": (ctypes.c_int, [_P(_vp), ctypes.c_int, _c, _c, _P(Program), ctypes.c_uint32, _P(_vp)]),
    "fdtdx_run_work": (ctypes.c_int, [_P(_vp), ctypes.c_int, _c, _c, _c, _P(Program), ctypes.c_uint32, _P(_vp)]),
    "fdtdx_gather_cube": (ctypes.c_int, [_vp, _vp, _c]),
    "fdtdx_get_addr": (ctypes.c_int, [_vp, _c, _P(_vp)]),
    "fdtdx_fused_depth": (ctypes.c_int, [_vp, _P(Program), ctypes.c_int]),
    "fdtdx_set_option": (ctypes.c_int, [_vp, _c, ctypes.c_int64]),
    "fdtdx_last_launch_info": (ctypes.c_int, [_vp, _c, ctypes.c_size_t]),
}

IPC_DESC_BYTES = 976  # sizeof(fdtdx_ipc_desc)

_lib = None


def load():
    """Load libfdtdx.so; raises FdtdxError when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FdtdxError(-1, f"{LIB_PATH} is missing: build it with `make -C fdtd_amd/csrc` "
                                 "(there is no CPU fallback)")
        # PyTorch-ROCm wheels ship their own copy of the HIP runtime.  One process must not end up
        # with two: if the system runtime libfdtdx.so links against is loaded first, a later
        # `import torch` finds "No HIP GPUs".  Loading torch's libraries first makes both sides
        # share the runtime torch ships.  torch is optional: without it nothing changes.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != OK:
        raise FdtdxError(rc, load().fdtdx_last_error().decode())


def make_program(kind, radius=1, order=ORDER_CHAIN, coef=None):
    """kind: 'star' | 'spc13' | 'jacobi7' (or the integer constants)."""
    kinds = {"star": STAR, "spc13": SPC13, "jacobi7": JACOBI7}
    k = kinds.get(kind, kind)
    if coef is None:
        # coefficients of the shipped drivers: fdtd1.cu:46-49, fdtdspc.cu:66-72, jacobi.cu:15-21
        coef = {STAR: [0.01, 0.02, 0.03, 0.04, 0.05, 0.06, 0.07],
                SPC13: [0.01, 0.02, 0.03, 0.04, 0.05, 0.06, 0.07, 0.02, 0.03, 0.04, 0.05, 0.06, 0.07],
                JACOBI7: [0.01, 0.02, 0.03, 0.04, 0.05, 0.06, 0.07]}[k]
    p = Program()
    p.kind, p.radius, p.order, p.reserved = k, (radius if k == STAR else 1), order, 0
    for i, v in enumerate(coef):
        p.coef[i] = float(v)
    return p


def make_composite(radius=4, order=ORDER_CHAIN, cz=0.5, cy=-0.2, cx=-0.3, scal_num=1.0, scal_den=6 * 8 + 1, wx=1.0, wy=2.0, wz=3.0):
    """The constants of main.cu:58-70 / :104-116 by default."""
    return Composite(radius, order, cz, cy, cx, scal_num, scal_den, wx, wy, wz)


def yee2d_steps(ex_ptr, ey_ptr, hz_ptr, nx, ny, t0, steps, ce=0.5, ch=0.7, order=ORDER_CHAIN, device=-1, stream=None):
    """Extension (no counterpart in the reference): `steps` steps of the 2-D ex/ey/hz scheme on dense
    [nx][ny] device arrays given by their device pointers."""
    check(load().fdtdx_yee2d_steps(device, ctypes.c_void_p(ex_ptr), ctypes.c_void_p(ey_ptr), ctypes.c_void_p(hz_ptr), nx, ny, t0, steps,
                                   ce, ch, order, ctypes.c_void_p(stream or 0)))


def decompose(Z, Y, X, M, rank=0, nprocs=1, pitch_x=0):
    """Pure host call: the z-slab geometry of `rank` (stencil.h:398-412)."""
    cfg = Config(Z, Y, X, M, rank, nprocs, -1, pitch_x, 0)
    g = Geometry()
    check(load().fdtdx_decompose(ctypes.byref(cfg), ctypes.byref(g)))
    return g
