"""fdtdx_composite_step_static: the entry point that allocates nothing (its per-address buffer is a file-scope
device array) gives what fdtdx_composite_step gives, bit for bit, through the C ABI -- on two calls in a row,
the second with a smaller input (fewer addresses on a smaller grid), which finds anything a call would wrongly
keep from the one before -- and rejects address lists beyond the array's size."""
import numpy as np
import pytest

from fdtd_amd import Stencil, make_composite, FdtdxError
from fdtd_amd import lib as flib

pytestmark = pytest.mark.gpu


def run_steps(shape, naddr, static, steps=3, seed=0):
    Z, Y, X = shape
    rng = np.random.default_rng(seed)
    fields = {n: rng.random((Z, Y, X), dtype=np.float32) for n in ("p0", "p1", "vel")}
    fields["sum"] = np.zeros((Z, Y, X), np.float32)
    # repeats and addresses on the planes filt scales and on the shell are all in here
    addr = rng.integers(0, Z * Y * X, naddr).astype(np.uint64)
    addr[: naddr // 8] = addr[naddr // 8: 2 * (naddr // 8)]
    comp = make_composite()
    with Stencil(Z, Y, X, 4) as st:
        for n, f in fields.items():
            st.mallocCube(n)
            st.transferCubeToGPU(n, f)
        st.mallocAddr("addr", naddr)
        st.transferAddrToGPU("addr", addr)
        step = st.compositeStepStatic if static else st.compositeStep
        s0, s1 = "p0", "p1"
        for _ in range(steps):
            step(s0, s1, "vel", "sum", "addr", comp, 1.0, Z // 2, Y // 3, X // 4, 0.9)
            s0, s1 = s1, s0
        st.sync()
        out = {}
        for n in ("p0", "p1", "sum"):
            out[n] = np.empty((Z, Y, X), np.float32)
            st.transferCubeToCPU(out[n], n)
    return out


def test_static_scratch_step_equals_allocating_step_twice_in_a_row():
    cases = (((40, 52, 140), 6000), ((24, 33, 70), 900))  # second: smaller grid, fewer addresses
    got = [run_steps(shape, naddr, static=True) for shape, naddr in cases]  # back to back on the device's one array
    for (shape, naddr), g in zip(cases, got):
        ref = run_steps(shape, naddr, static=False)
        for n in ref:
            assert np.array_equal(g[n].view(np.uint32), ref[n].view(np.uint32)), (shape, naddr, n)


def test_static_scratch_rejects_lists_beyond_its_size():
    Z, Y, X = 40, 64, 128
    n = flib.STATIC_ADDR_MAX + 1
    addr = (np.arange(n, dtype=np.uint64) * 7) % np.uint64(Z * Y * X)
    with Stencil(Z, Y, X, 4) as st:
        for name in ("p0", "p1", "vel", "sum"):
            st.mallocCube(name)
        st.mallocAddr("addr", n)
        st.transferAddrToGPU("addr", addr)
        with pytest.raises(FdtdxError) as e:
            st.compositeStepStatic("p0", "p1", "vel", "sum", "addr", make_composite(), 1.0, Z // 2, Y // 3, X // 4, 0.9)
        assert e.value.code == flib.ERR_UNSUPPORTED and "static scratch" in str(e.value)
        # the allocating entry point takes the same list
        st.compositeStep("p0", "p1", "vel", "sum", "addr", make_composite(), 1.0, Z // 2, Y // 3, X // 4, 0.9)
        st.sync()


def _fields(shape, naddr, seed):
    Z, Y, X = shape
    rng = np.random.default_rng(seed)
    f = {n: rng.random((Z, Y, X), dtype=np.float32) for n in ("p0", "p1", "vel")}
    f["sum"] = np.zeros((Z, Y, X), np.float32)
    return f, rng.integers(0, Z * Y * X, naddr).astype(np.uint64)


def test_static_step_is_capturable_where_allocation_is_illegal():
    """The property the entry point exists for: nothing on its path allocates.  Under stream capture (torch.cuda.graph, error
    mode "global") hipMalloc is illegal and fails; the static step is captured on a context that has never run a step, the
    graph replays, and the fields equal those of an eager static step -- while the allocating entry point, on a context in
    the same state, is refused inside a capture (its first call has to allocate its per-address buffer)."""
    import torch
    shape, naddr = (24, 33, 70), 900
    Z, Y, X = shape
    fields, addr = _fields(shape, naddr, 5)
    comp = make_composite()
    args = ("p0", "p1", "vel", "sum", "addr", comp, 1.0, Z // 2, Y // 3, X // 4, 0.9)

    def context():
        st = Stencil(Z, Y, X, 4)
        for n, f in fields.items():
            st.mallocCube(n)
            st.transferCubeToGPU(n, f)
        st.mallocAddr("addr", naddr)
        st.transferAddrToGPU("addr", addr)
        return st

    def fetch(st):
        out = {}
        for n in ("p0", "sum"):
            out[n] = np.empty(shape, np.float32)
            st.transferCubeToCPU(out[n], n)
        return out

    with context() as ref:  # eager; also loads every kernel of the step before anything is captured
        ref.compositeStepStatic(*args)
        ref.sync()
        want = fetch(ref)
    with context() as st:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="global"):
            st.compositeStepStatic(*args, stream=torch.cuda.current_stream().cuda_stream)
        graph.replay()
        torch.cuda.synchronize()
        got = fetch(st)
        del graph
    for n in want:
        assert np.array_equal(got[n].view(np.uint32), want[n].view(np.uint32)), n
    with context() as st:
        graph = torch.cuda.CUDAGraph()
        with pytest.raises((FdtdxError, RuntimeError)):
            with torch.cuda.graph(graph, capture_error_mode="global"):
                st.compositeStep(*args, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
