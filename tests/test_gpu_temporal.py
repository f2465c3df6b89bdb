"""The fused kernels (two or three time steps per launch, fdtd_amd/csrc/tb_kernel.h) inside
fdtdx_sweep: bit-exact against the CPU oracle's step-by-step loop, for every compiled shape,
step counts of every residue, grids with several y tiles and z chunks, ragged and wide x, and
cubes whose two shells differ (the drivers' p0/p1 are independent random fields, fdtd1.cu:28-29)."""
import numpy as np
import pytest

from fdtd_amd import Stencil, make_program, ORDER_CHAIN, ORDER_GOLDEN

pytestmark = pytest.mark.gpu


def run(cpu, kind, shape, steps, tb_variant=-1, tb_zchunks=0, temporal=1, M=1, golden=False, fuse_wide=0, no_strip=0):
    Z, Y, X = shape
    if kind == "jacobi7":
        a = cpu.init_jacobi(Z, Y, X)
        b = a.copy()
    else:
        a, b = cpu.init_mt(Z, Y, X)
    ca, cb = a.copy(), b.copy()
    cpu.run(kind, ca, cb, steps, M=M, order=cpu.ORDER_GOLDEN if golden else cpu.ORDER_CHAIN)
    with Stencil(Z, Y, X, M) as st:
        st.mallocCube("p0")
        st.mallocCube("p1")
        st.transferCubeToGPU("p0", a)
        st.transferCubeToGPU("p1", b)
        st.setOption("temporal", temporal)
        st.setOption("tb_variant", tb_variant)
        st.setOption("tb_zchunks", tb_zchunks)
        st.setOption("fuse_wide", fuse_wide)
        st.setOption("no_strip", no_strip)
        st.sweep("p0", "p1", make_program(kind, radius=M, order=ORDER_GOLDEN if golden else ORDER_CHAIN), steps)
        st.sync()
        ga, gb = np.empty_like(a), np.empty_like(b)
        st.transferCubeToCPU(ga, "p0")
        st.transferCubeToCPU(gb, "p1")
        info = st.lastLaunchInfo()
    assert np.array_equal(ga.view(np.uint32), ca.view(np.uint32)), info
    assert np.array_equal(gb.view(np.uint32), cb.view(np.uint32)), info
    return info


@pytest.mark.parametrize("steps", [4, 5, 6, 7, 8, 9, 10, 11, 13])
@pytest.mark.parametrize("kind", ["star", "jacobi7"])
@pytest.mark.parametrize("kt", [2, 3])
def test_fused_step_counts(oracle_cpu, kt, kind, steps):
    info = run(oracle_cpu, kind, (13, 61, 509), steps, tb_zchunks=2, temporal=kt)
    assert f"fused{kt}" in info


@pytest.mark.parametrize("steps", [4, 7, 9, 10, 11])
@pytest.mark.parametrize("kind,M,kt", [("star", 1, 3), ("star", 1, 2), ("jacobi7", 1, 3), ("jacobi7", 1, 2), ("star", 2, 2)])
def test_fused_host_order(oracle_cpu, kind, M, kt, steps):
    """The fused kernels in the host loops' order (fdtd1.cu:46-49, fdtd2.cu:46-52, jacobi.cu:478-485: separate IEEE
    multiplies and adds), the order the reference's compiled loops pin: equal to the oracle's golden-order loop bit for bit."""
    info = run(oracle_cpu, kind, (13, 61, 509) if M == 1 else (17, 61, 509), steps, tb_zchunks=2, temporal=kt if M == 1 else 1, M=M, golden=True)
    assert f"fused{kt}" in info


def test_host_order_wide_rows(oracle_cpu):
    for kind, M in (("star", 1), ("jacobi7", 1), ("star", 2)):
        assert "fused" in run(oracle_cpu, kind, (9, 30, 1024), 7, M=M, golden=True)


def test_auto_fuses_three_steps(oracle_cpu):
    assert "fused3" in run(oracle_cpu, "star", (13, 61, 509), 10)


@pytest.mark.parametrize("kt,variant", [(2, v) for v in range(9)] + [(3, 0), (3, 1), (3, 2)])
@pytest.mark.parametrize("kind", ["star", "jacobi7"])
def test_every_fused_shape(oracle_cpu, kind, kt, variant):
    info = run(oracle_cpu, kind, (21, 75, 512), 7, tb_variant=variant, tb_zchunks=3, temporal=kt)
    assert f"fused{kt}" in info


@pytest.mark.parametrize("shape", [(3, 3, 3), (5, 7, 9), (9, 20, 21), (40, 19, 300), (7, 130, 64), (31, 9, 40)])
@pytest.mark.parametrize("kt", [2, 3])
def test_fused_small_and_ragged(oracle_cpu, kt, shape):
    run(oracle_cpu, "star", shape, 7, temporal=kt)


@pytest.mark.parametrize("X", [513, 700, 1000, 1024, 1531])
@pytest.mark.parametrize("kind", ["star", "jacobi7"])
@pytest.mark.parametrize("kt", [2, 3])
def test_fused_wide_rows_are_cut_into_x_tiles(oracle_cpu, kt, kind, X):
    """Rows wider than a wave holds (X > 512) are fused too, as x tiles overlapping by one quad."""
    info = run(oracle_cpu, kind, (9, 30, X), 7, temporal=kt)
    assert f"fused{kt}" in info


@pytest.mark.parametrize("shape", [(9, 130, 1024), (12, 47, 513), (9, 30, 1531), (14, 75, 1036)])
@pytest.mark.parametrize("kind", ["star", "jacobi7"])
def test_fused_strip_tail(oracle_cpu, kind, shape):
    """The few quads a wide row has left after its last full x tile go to the strip shape (segments of 16 quads, each in another
    y tile): several rows of strip workgroups, the last one ragged; the half-width tile in its place gives the same bits."""
    info = run(oracle_cpu, kind, shape, 7, temporal=3, tb_zchunks=2)
    assert "fused3" in info and "+strip tail" in info, info
    info = run(oracle_cpu, kind, shape, 7, temporal=3, no_strip=1)
    assert "fused3" in info and "+half-width tail" in info, info


@pytest.mark.parametrize("golden", [False, True])
@pytest.mark.parametrize("shape", [(13, 130, 1024), (14, 47, 513), (13, 75, 1036)])
def test_fused_strip_tail_radius2_and_host_order(oracle_cpu, shape, golden):
    info = run(oracle_cpu, "star", shape, 7, M=2, tb_zchunks=2, golden=golden)
    assert "fused2" in info and "+strip tail" in info, info
    info = run(oracle_cpu, "star", shape, 7, M=2, no_strip=1, golden=golden)
    assert "fused2" in info and "+half-width tail" in info, info
    if golden:
        for kind in ("star", "jacobi7"):
            info = run(oracle_cpu, kind, shape, 7, temporal=3, golden=True)
            assert "fused3" in info and "+strip tail" in info, info


@pytest.mark.parametrize("golden", [False, True])
@pytest.mark.parametrize("shape", [(9, 130, 1024), (12, 47, 513), (14, 75, 1036), (9, 12, 1024)])
def test_fused_strip_tail_13_point(oracle_cpu, shape, golden):
    """The 13-point two-step kernel's strip shape: a wave's four segments lie in four y tiles, so "this row is shell throughout" is
    a per-lane fact there (first and last y tile, one-row-high grids included); in-place loop over two cubes, both orders."""
    info = run(oracle_cpu, "spc13", shape, 7, tb_zchunks=2, golden=golden)
    assert "fused2" in info and "+strip tail" in info, info
    info = run(oracle_cpu, "spc13", shape, 6, golden=golden)
    assert "fused2" in info and "+strip tail" in info, info
    info = run(oracle_cpu, "spc13", shape, 7, no_strip=1, golden=golden)
    assert "fused2" in info and "+half-width tail" in info, info


@pytest.mark.parametrize("steps", [4, 5, 6, 7, 8, 9, 11])
def test_fused_radius2_step_counts(oracle_cpu, steps):
    """fdtd2's update fused two steps per launch (tbm_kernel.h): several y tiles and z chunks, ragged x."""
    info = run(oracle_cpu, "star", (17, 61, 509), steps, tb_zchunks=2, M=2)
    assert "fused2" in info and "r=2" in info


@pytest.mark.parametrize("shape", [(5, 5, 5), (7, 9, 11), (9, 20, 21), (40, 19, 300), (9, 130, 64), (31, 9, 40), (9, 30, 513), (9, 30, 1024),
                                   (9, 30, 1531)])
def test_fused_radius2_small_ragged_wide(oracle_cpu, shape):
    run(oracle_cpu, "star", shape, 7, M=2)


@pytest.mark.parametrize("steps", [4, 5, 6, 7, 8, 9, 11])
@pytest.mark.parametrize("M", [3, 4])
def test_fused_radius3_and_4_step_counts(oracle_cpu, M, steps):
    """The two-step kernels of radius 3 and 4 (tbm_kernel.h with three rows per wave; behind the "fuse_wide" option because they
    run slower than single steps): bit-equal to the step-by-step loop at every residue of the step count, several y tiles and
    z chunks, ragged x; without the option the same loop runs single steps."""
    info = run(oracle_cpu, "star", (8 * M + 5, 61, 253), steps, tb_zchunks=2, M=M, fuse_wide=1)
    assert "fused2" in info and f"r={M}" in info.split("|")[-1]
    assert "fused" not in run(oracle_cpu, "star", (8 * M + 5, 61, 253), steps, M=M)


@pytest.mark.parametrize("M", [3, 4])
@pytest.mark.parametrize("shape", [(9, 9, 9), (11, 13, 17), (13, 20, 21), (40, 19, 300), (13, 130, 64), (31, 11, 40), (11, 30, 513), (11, 30, 1024),
                                   (11, 30, 1531), (21, 40, 512)])
def test_fused_radius3_and_4_small_ragged_wide(oracle_cpu, M, shape):
    """...on small and ragged grids and on rows cut into x tiles, which overlap by two quads on each side at these radii."""
    run(oracle_cpu, "star", shape, 7, M=M, fuse_wide=1)


@pytest.mark.parametrize("golden", [False, True], ids=["device-order", "golden-order"])
@pytest.mark.parametrize("steps", [4, 5, 6, 7, 8, 9, 10, 11, 12, 13])
def test_fused_spc13_step_counts(oracle_cpu, steps, golden):
    """fdtdspc's update fused two steps per launch (tbs_kernel.h), in place between the two cubes (hops b -> a -> b with
    the level-0 shell patched in on every second hop): several y tiles and z chunks, ragged x, every residue of the
    step count, both arithmetic orders."""
    info = run(oracle_cpu, "spc13", (17, 61, 509), steps, tb_zchunks=2, golden=golden)
    assert ("fused2" in info) == (steps >= 5)


@pytest.mark.parametrize("shape", [(3, 3, 3), (5, 7, 9), (9, 20, 21), (40, 19, 300), (9, 130, 64), (31, 9, 40), (9, 30, 513), (9, 30, 1024),
                                   (9, 30, 1531), (64, 40, 512), (9, 20, 257), (9, 20, 260), (9, 20, 261), (9, 20, 256), (9, 20, 253)])
def test_fused_spc13_small_ragged_wide(oracle_cpu, shape):
    run(oracle_cpu, "spc13", shape, 9)


def test_temporal_off_equals_on(oracle_cpu):
    info = run(oracle_cpu, "star", (12, 40, 256), 8, temporal=0)
    assert "fused" not in info


@pytest.mark.parametrize("kind,steps,G,R", [("star", 8, 2, 1), ("jacobi7", 7, 2, 1), ("star", 11, 2, 1), ("star", 8, 3, 1),
                                            ("jacobi7", 7, 3, 1), ("star", 11, 3, 1), ("star", 9, 4, 2), ("star", 8, 4, 2)])
def test_fused_on_two_slabs_equals_one(oracle_cpu, G, kind, steps, R):
    """Fused hops on a 2-way z split (ghost depth G = steps per launch x radius R, one exchange per hop) on one
    GPU: the two slabs run in this process and trade ghost planes by device copies.  Equal to the
    oracle's undecomposed step-by-step result bit for bit."""
    import torch
    from fdtd_amd.slab import HipSlabBackend, SlabRunner
    Z, Y, X = 48, 75, 512
    prog = make_program(kind, radius=R)
    if kind == "jacobi7":
        a = oracle_cpu.init_jacobi(Z, Y, X)
        b = a.copy()
    else:
        a, b = oracle_cpu.init_mt(Z, Y, X)
    bes = [HipSlabBackend(Z, Y, X, G, r, 2, 0) for r in range(2)]
    for be in bes:
        g = be.g
        lo, n = int(g.first_global), int(g.local_planes)
        for name, arr in (("p0", a), ("p1", b)):
            be.alloc(name)
            be.st.uploadLocal(name, arr[lo:lo + n])
        be.alloc("c")
        be.copy_cube("c", "p1")
    runs = [SlabRunner(be, G, prog, r, 2) for r, be in enumerate(bes)]
    assert all(r.fusable() == G // R for r in runs)

    def exchange(name):
        l0 = runs[0].local
        bes[1].planes(name, 0, G).copy_(bes[0].planes(name, l0 - 2 * G, G))
        bes[0].planes(name, l0 - G, G).copy_(bes[1].planes(name, G, G))

    def single(out, inp):
        for run in runs:
            run.b.sweep(out, inp, prog, run.z_lo, run.z_hi)
        exchange(out)

    def hop(out, inp, mid):
        for run in runs:
            run.b.sweep_fused(out, inp, mid, prog, G // R, run.z_lo, run.z_hi)
        exchange(out)

    done = 0
    if G == 3:  # the loop of SlabRunner.run, with the exchange done by device copies
        for _ in range((steps - 1) // 3):
            out, inp = ("p0", "p1") if done % 2 == 0 else ("p1", "p0")
            hop(out, inp, out)
            done += 3
    else:
        pairs = steps // 2 - 1 if steps % 2 == 0 else ((steps - 1) // 2) & ~1
        cur, oth = "p1", "c"
        for _ in range(pairs):
            hop(oth, cur, "p0")
            cur, oth = oth, cur
        done = 2 * pairs
        if cur != "p1":
            single("p0", cur)
            done += 1
    for t in range(done, steps):
        single("p0", "p1") if t % 2 == 0 else single("p1", "p0")
    torch.cuda.synchronize()
    got0 = np.concatenate([be.st.downloadOwned("p0") for be in bes], axis=0)
    got1 = np.concatenate([be.st.downloadOwned("p1") for be in bes], axis=0)
    oracle_cpu.run(kind, a, b, steps, M=R, order=oracle_cpu.ORDER_CHAIN)
    for name, got, ref in (("p0", got0, a), ("p1", got1, b)):
        bad = np.argwhere(got.view(np.uint32) != ref.view(np.uint32))
        assert len(bad) == 0, f"{name}: {len(bad)} cells differ, first {bad[:4].tolist()} last {bad[-2:].tolist()}"


@pytest.mark.parametrize("work", ["w", None], ids=["named-third-cube", "no-third-cube"])
@pytest.mark.parametrize("steps", [6, 9])
def test_sweep_work_takes_the_third_cube_by_name(oracle_cpu, steps, work):
    """fdtdx_sweep_work: the radius-2 two-step loop hops through a cube the caller names (no allocation inside the
    library); without one the loop runs single steps.  Same bits either way."""
    Z, Y, X, M = 17, 61, 509, 2
    a, b = oracle_cpu.init_mt(Z, Y, X)
    ca, cb = a.copy(), b.copy()
    oracle_cpu.run("star", ca, cb, steps, M=M)
    with Stencil(Z, Y, X, M) as st:
        for name in ("p0", "p1", "w"):
            st.mallocCube(name)
        st.transferCubeToGPU("p0", a)
        st.transferCubeToGPU("p1", b)
        st.sweep("p0", "p1", make_program("star", radius=M), steps, work=work)
        st.sync()
        ga, gb = np.empty_like(a), np.empty_like(b)
        st.transferCubeToCPU(ga, "p0")
        st.transferCubeToCPU(gb, "p1")
        info = st.lastLaunchInfo()
        with pytest.raises(Exception):
            st.sweep("p0", "p1", make_program("star", radius=M), steps, work="p1")
    assert ("fused2" in info) == (work is not None), info
    assert np.array_equal(ga.view(np.uint32), ca.view(np.uint32)) and np.array_equal(gb.view(np.uint32), cb.view(np.uint32)), info


@pytest.mark.parametrize("golden", [False, True], ids=["device-order", "golden-order"])
@pytest.mark.parametrize("X", [61, 64, 130, 509, 515])
@pytest.mark.parametrize("M", [1, 2, 3, 5, 6, 7])
def test_single_step_loops_take_shell_cells_from_the_strips(oracle_cpu, M, X, golden):
    """Loops of eight and more single steps gather the shell cells of the quads that straddle the x shell into strips once
    and merge them from there (instead of re-reading the output cube every step): every radius that is not a multiple of
    4, row lengths that put one, two or three interior cells into the last such quad, both orders."""
    info = run(oracle_cpu, "star", (2 * M + 9, 2 * M + 23, X), 9, temporal=0, M=M, golden=golden)
    assert "fused" not in info


def test_single_step_jacobi_loop_with_strips(oracle_cpu):
    run(oracle_cpu, "jacobi7", (11, 25, 130), 10, temporal=0)

