"""ctypes front-end of the CPU checker libraries -- TEST INFRASTRUCTURE ONLY.

Functions follow the reference's driver loops:
  init_mt      fdtd1.cu:21-31        run("star", M)   fdtdN.cu:40-52 (golden) / :61-64 (device order)
  init_jacobi  jacobi.cu:412-417     run("spc13")     fdtdspc.cu:40-55 / :64-73
                                     run("jacobi7")   jacobi.cu:474-490 / :31-38
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORDER_GOLDEN, ORDER_CHAIN = 0, 1
KINDS = {"star": 0, "spc13": 1, "jacobi7": 2}

# coefficient sets of the shipped drivers
STAR_COEF = np.array([0.01, 0.02, 0.03, 0.04, 0.05, 0.06, 0.07], dtype=np.float32)  # fdtd1.cu:46-49
SPC13_COEF = np.array([0.01, 0.02, 0.03, 0.04, 0.05, 0.06, 0.07, 0.02, 0.03, 0.04, 0.05, 0.06, 0.07],
                      dtype=np.float32)  # fdtdspc.cu:66-72
JACOBI_COEF = np.array([0.01, 0.02, 0.03, 0.04, 0.05, 0.06, 0.07], dtype=np.float32)  # jacobi.cu:15-21

_f32p = ctypes.POINTER(ctypes.c_float)


def _p(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_f32p)


def build():
    """(Re)build liboracle.so and, where the reference is present, oracle/_ref."""
    subprocess.check_call(["make", "-C", _HERE, "-s", "all"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        sz = ctypes.c_size_t
        L.oracle_init_mt.argtypes = [_f32p, _f32p, _f32p, ctypes.c_int, sz, sz, sz, sz]
        L.oracle_init_jacobi.argtypes = [_f32p, sz, sz, sz]
        L.oracle_star_step.argtypes = [_f32p, _f32p, sz, sz, sz, ctypes.c_int, _f32p, ctypes.c_int, sz, sz]
        L.oracle_spc13_step.argtypes = [_f32p, _f32p, sz, sz, sz, _f32p, ctypes.c_int, sz, sz]
        L.oracle_jacobi7_step.argtypes = [_f32p, _f32p, sz, sz, sz, _f32p, ctypes.c_int, sz, sz]
        L.oracle_run.argtypes = [ctypes.c_int, _f32p, _f32p, sz, sz, sz, ctypes.c_int, _f32p, ctypes.c_int,
                                 ctypes.c_int]
        _f64p, _u64p = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)
        L.oracle_composite_sweep.argtypes = [_f32p, _f32p, _f32p, sz, sz, sz, ctypes.c_int, _f64p, ctypes.c_int, sz, sz, sz]
        L.oracle_inject.argtypes = [_f32p, _u64p, sz, ctypes.c_float]
        L.oracle_filt.argtypes = [_f32p, sz, sz, sz, sz, sz, sz, sz, ctypes.c_double, ctypes.c_int, sz, sz]
        L.oracle_mul_acc.argtypes = [_f32p, _f32p, _f32p, sz, sz, sz, ctypes.c_int, ctypes.c_int, sz, sz]
        L.oracle_main_step.argtypes = [_f32p, _f32p, _f32p, _f32p, _u64p, sz, sz, sz, sz, ctypes.c_int]
        L.oracle_yee2d_init.argtypes = [_f32p, _f32p, _f32p, sz, sz]
        L.oracle_yee2d_step.argtypes = [_f32p, _f32p, _f32p, sz, sz, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int]
        L.oracle_yee3d_half.argtypes = [ctypes.POINTER(_f32p), ctypes.c_int, sz, sz, sz, ctypes.c_float, ctypes.c_int, sz, sz]
        for f in (L.oracle_init_mt, L.oracle_init_jacobi, L.oracle_star_step, L.oracle_spc13_step,
                  L.oracle_jacobi7_step, L.oracle_run, L.oracle_composite_sweep, L.oracle_inject, L.oracle_filt,
                  L.oracle_mul_acc, L.oracle_main_step, L.oracle_yee2d_init, L.oracle_yee2d_step, L.oracle_yee3d_half):
            f.restype = None
        _lib = L
    return _lib


def init_mt(nz, ny, nx, z_begin=0, nfields=2):
    """Initial fields of fdtdN/fdtdspc (2 fields) or main.cu (3 fields) for planes [z_begin, z_begin+nz)."""
    p0 = np.empty((nz, ny, nx), np.float32)
    p1 = np.empty((nz, ny, nx), np.float32)
    f2 = np.empty((nz, ny, nx), np.float32) if nfields == 3 else None
    lib().oracle_init_mt(_p(p0), _p(p1), _p(f2) if f2 is not None else None, nfields, z_begin, nz, ny, nx)
    return (p0, p1) if nfields == 2 else (p0, p1, f2)


def init_jacobi(nz, ny, nx):
    a = np.empty((nz, ny, nx), np.float32)
    lib().oracle_init_jacobi(_p(a), nz, ny, nx)
    return a


def default_coef(kind):
    return {"star": STAR_COEF, "spc13": SPC13_COEF, "jacobi7": JACOBI_COEF}[kind]


def step(kind, out, inp, M=1, coef=None, order=ORDER_CHAIN, z_lo=None, z_hi=None):
    """One sweep out = f(inp) over local planes [z_lo, z_hi) (default: the whole interior)."""
    nz, ny, nx = inp.shape
    coef = np.ascontiguousarray(default_coef(kind) if coef is None else coef, np.float32)
    h = M if kind == "star" else 1
    z_lo = h if z_lo is None else z_lo
    z_hi = nz - h if z_hi is None else z_hi
    L = lib()
    if kind == "star":
        L.oracle_star_step(_p(out), _p(inp), nz, ny, nx, M, _p(coef), order, z_lo, z_hi)
    elif kind == "spc13":
        L.oracle_spc13_step(_p(out), _p(inp), nz, ny, nx, _p(coef), order, z_lo, z_hi)
    else:
        L.oracle_jacobi7_step(_p(out), _p(inp), nz, ny, nx, _p(coef), order, z_lo, z_hi)


def run(kind, a, b, steps, M=1, coef=None, order=ORDER_CHAIN):
    """`steps` ping-pong steps in place: step 0 writes a = f(b), then the roles swap."""
    nz, ny, nx = a.shape
    coef = np.ascontiguousarray(default_coef(kind) if coef is None else coef, np.float32)
    lib().oracle_run(KINDS[kind], _p(a), _p(b), nz, ny, nx, M, _p(coef), order, steps)


# main.cu composite pipeline ---------------------------------------------------------------------
MAIN_COEF = np.array([0.5, -0.2, -0.3, 1.0, 6 * 8 + 1, 1.0, 2.0, 3.0], dtype=np.float64)  # main.cu:58-70


def _u64(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


def composite_sweep(out, inp, vel, M=4, coef=None, order=ORDER_CHAIN, z_lo=None, z_hi=None, zg0=0):
    """main.cu:55-72 (golden) / :104-118 (device order) on local planes [z_lo, z_hi)."""
    nz, ny, nx = inp.shape
    k = np.ascontiguousarray(MAIN_COEF if coef is None else coef, np.float64)
    z_lo = M if z_lo is None else z_lo
    z_hi = nz - M if z_hi is None else z_hi
    lib().oracle_composite_sweep(_p(out), _p(inp), _p(vel), nz, ny, nx, M, k.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                 order, z_lo, z_hi, zg0)


def inject(cube, addr, add=1.0):
    """main.cu:73: cube.flat[addr[i]] += add for every i (repeats add repeatedly)."""
    lib().oracle_inject(_p(cube), _u64(addr), len(addr), add)


def filt(cube, zsel, ysel, xsel, scale=0.9, order=ORDER_CHAIN, z_lo=None, z_hi=None, zg0=0):
    """main.cu:75-79 / :124-128 on local planes [z_lo, z_hi) (default all)."""
    nz, ny, nx = cube.shape
    lib().oracle_filt(_p(cube), nz, ny, nx, zg0, zsel, ysel, xsel, scale, order, 0 if z_lo is None else z_lo,
                      nz if z_hi is None else z_hi)


def mul_acc(s, a, b, M=4, order=ORDER_CHAIN, z_lo=None, z_hi=None):
    """main.cu:81-84 / :129-133."""
    nz, ny, nx = s.shape
    lib().oracle_mul_acc(_p(s), _p(a), _p(b), nz, ny, nx, M, order, M if z_lo is None else z_lo,
                         nz - M if z_hi is None else z_hi)


def main_step(p0, p1, vel, s, addr, order=ORDER_CHAIN):
    """One whole step of main.cu:54-84 (without the swap)."""
    nz, ny, nx = p0.shape
    lib().oracle_main_step(_p(p0), _p(p1), _p(vel), _p(s), _u64(addr), len(addr), nz, ny, nx, order)


def ref_main_step(p0, p1, vel, s, addr):
    """The reference's own loop body main.cu:54-84, compiled (oracle/_ref)."""
    nz, ny, nx = p0.shape
    a = np.ascontiguousarray(addr, np.uint64)
    ref().ref_main_step(_p(p0), _p(p1), _p(vel), _p(s), a.ctypes.data_as(ctypes.POINTER(ctypes.c_size_t)), len(a), nz, ny, nx)


# extension: E/H updates that the reference does not have (this file's C source is their definition) -------
def yee2d_init(nx, ny):
    ex, ey, hz = (np.empty((nx, ny), np.float32) for _ in range(3))
    lib().oracle_yee2d_init(_p(ex), _p(ey), _p(hz), nx, ny)
    return ex, ey, hz


def yee2d_step(ex, ey, hz, t, ce=0.5, ch=0.7, order=ORDER_CHAIN):
    nx, ny = ex.shape
    lib().oracle_yee2d_step(_p(ex), _p(ey), _p(hz), nx, ny, t, ce, ch, order)


def yee3d_half(fields, which, coef, order=ORDER_CHAIN, z_lo=None, z_hi=None):
    """fields = [Ex, Ey, Ez, Hx, Hy, Hz]; which 0: H half step, 1: E half step."""
    nz, ny, nx = fields[0].shape
    arr = (_f32p * 6)(*[_p(f) for f in fields])
    lib().oracle_yee3d_half(arr, which, nz, ny, nx, coef, order, 1 if z_lo is None else z_lo, nz - 1 if z_hi is None else z_hi)


def yee3d_step(fields, ce, ch, order=ORDER_CHAIN):
    yee3d_half(fields, 0, ch, order)
    yee3d_half(fields, 1, ce, order)


# ----------------------------------------------------------------------------------------------
# the reference's own loops (oracle/_ref), when built
# ----------------------------------------------------------------------------------------------
_ref = None


def ref_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libfdtd_ref_strict.so"))


def ref():
    global _ref
    if _ref is None:
        L = ctypes.CDLL(os.path.join(_HERE, "_ref", "libfdtd_ref_strict.so"))
        sz = ctypes.c_size_t
        i64 = ctypes.c_int64
        L.ref_init_mt.argtypes = [_f32p, _f32p, sz, sz, sz]
        L.ref_init_mt3.argtypes = [_f32p, _f32p, _f32p, _f32p, sz, sz, sz]
        L.ref_star_step.argtypes = [ctypes.c_int, _f32p, _f32p, sz, sz, sz]
        L.ref_spc13_step.argtypes = [_f32p, _f32p, sz, sz, sz]
        L.ref_init_jacobi.argtypes = [_f32p, _f32p, i64, i64, i64]
        L.ref_jacobi_step.argtypes = [_f32p, _f32p, i64, i64, i64]
        L.ref_main_step.argtypes = [_f32p, _f32p, _f32p, _f32p, ctypes.POINTER(ctypes.c_size_t), sz, sz, sz, sz]
        L.ref_run.argtypes = [ctypes.c_int, ctypes.c_int, _f32p, _f32p, sz, sz, sz, ctypes.c_int]
        L.ref_run.restype = ctypes.c_int
        _ref = L
    return _ref


def ref_init_mt(nz, ny, nx):
    p0 = np.empty((nz, ny, nx), np.float32)
    p1 = np.empty((nz, ny, nx), np.float32)
    ref().ref_init_mt(_p(p0), _p(p1), nz, ny, nx)
    return p0, p1


def ref_init_jacobi(nz, ny, nx):
    a = np.empty((nz, ny, nx), np.float32)
    b = np.empty((nz, ny, nx), np.float32)
    ref().ref_init_jacobi(_p(a), _p(b), nz, ny, nx)
    return a, b


def ref_run(kind, a, b, steps, M=1):
    nz, ny, nx = a.shape
    rc = ref().ref_run(KINDS[kind], M, _p(a), _p(b), nz, ny, nx, steps)
    assert rc == 0
