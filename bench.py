#!/usr/bin/env python3
"""bench.py -- cell-update throughput of the sweep hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--variant config3|fdtd4|...] [--size 512]

Default workload ("config3"): every radius of the fdtdN drivers (fdtd1.cu .. fdtd8.cu) on the drivers' 512^3
grid, each timed for K steps; `value` is the SLOWEST of the eight (so the line tracks "every fdtdN above the
bar"), `variants` holds all eight plus fdtdspc and jacobi.  --variant NAME times that one variant only.
Started plainly with --gpus N > 1 the script launches its N ranks itself (fresh child processes with
RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, before anything touches a GPU) and relays rank 0's line; started
under torch.distributed.run it is one of the ranks.

A "step" is one time step of the drivers' loop (fdtd1.cu:68-80): one sweep of the whole slab
plus, on N > 1 GPUs, the ghost-plane exchange with the z neighbours.  Weak scaling: every GPU
owns an n^3 slab of a (N*n) x n x n grid, as the reference's z-slab split gives (stencil.h:398-412).
Inputs are the drivers' synthetic fields (fdtd1.cu:21-31 / jacobi.cu:412-417), resident in HBM
before the timed region.  Rank 0 prints ONE JSON line.

metric   Gcell/s = updated cells x steps / wall seconds (max over ranks), updated cells per
         step = the interior the reference updates (fdtd1.cu:43-45) summed over all ranks.
roofline bound = HBM; algorithmic traffic 8 B per cell-update (one fp32 read + one fp32 write);
         achieved = 8 B x cells x time steps one launch of the dominant kernel advances / that kernel's
         launch duration, measured here with HIP events on the launch stream over a run of launches of
         that kernel alone (a fused loop's closing single steps are a different kernel); peak 8 TB/s.
         A kernel that fuses k steps moves fewer than 8 B per update, so `achieved` can exceed what the
         memory system delivers; `traffic` (rocprofv3 FETCH_SIZE/WRITE_SIZE per launch of the same
         kernel, profiles/traffic.json) and `hbm_achieved` = traffic / launch duration say what it
         really moved.
spin-up  the device needs 15-30 ms of continuous work after an idle spell (the host-side field generation)
         before its clocks settle: a sweep is up to 1.6x slower in that window.  Before the W warm-up steps
         the same loop therefore runs untimed for --spinup-ms (default 100 ms) and the fields are then put
         back to their initial values by device copies, so warm-up and timed steps see the same data as
         without it.  --spinup-ms 0 switches it off.
cpu_baseline  the reference's own golden loop compiled from its sources (oracle/_ref/ref_bench,
         CMakeLists.txt:17 flags) or, where that binary is absent, this repo's restatement
         (oracle/liboracle.so), timed on this box's host cores on the same grid for a bounded
         number of steps.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VARIANTS = {  # name -> (program kind, radius, reference driver)
    **{f"fdtd{m}": ("star", m, f"fdtd{m}.cu") for m in range(1, 9)},
    "fdtdspc": ("spc13", 1, "fdtdspc.cu"),
    "jacobi": ("jacobi7", 1, "jacobi.cu"),
    "main": ("composite", 4, "main.cu"),  # sweep + inject + filt + mul (main.cu:149-171), one rank
    # extensions without a counterpart in the reference (defined by oracle/fdtd_oracle.c):
    "yee3d": ("yee3d", 1, "none: extension"),  # six-field E/H step, 48 B per cell-step fused (72 as two half steps)
    "yee2d": ("yee2d", 1, "none: extension"),  # ex/ey/hz on n x n, 36 B per cell-step as two launches (--size 2048)
}
BYTES_PER_UPDATE_MAIN = 24.0  # p1, vel, old p0, sum read + p0, sum write
PEAK_HBM_GBS = 8000.0
BYTES_PER_UPDATE = 8.0


def cpu_baseline(kind, radius, n, budget_s=12.0):
    """Time the CPU golden loop on the host cores; returns the cpu_baseline object."""
    kid = {"star": 0, "spc13": 1, "jacobi7": 2, "composite": 3}[kind]
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_bench")
    # every host core this process may use: OMP_NUM_THREADS where the box sets it (its CPU share), else the affinity mask
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or avail
    env = dict(os.environ, OMP_NUM_THREADS=str(threads))
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
    except OSError:
        pass
    host = {"nproc": os.cpu_count(), "affinity": avail, "OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS"), "cpu_model": model}
    if kind == "jacobi7":
        threads = 1  # jacobi.cu:475-485 is a serial loop in the reference
    if os.path.exists(exe):
        def run(steps):
            out = subprocess.run([exe, str(kid), str(radius), str(n), str(n), str(n), str(steps)], env=env,
                                 capture_output=True, text=True, timeout=600, check=True).stdout
            return json.loads(out.strip().splitlines()[-1])
        probe = run(2)
        steps = max(2, min(2000, int(budget_s / max(probe["seconds"] / 2, 1e-4))))
        r = run(steps)
        return {"value": round(r["gcells_per_s"], 3), "unit": "Gcell/s", "cores": threads if kind != "jacobi7" else 1,
                "kind": "reference", "host": host,
                "sample": f"{steps} steps of the {n}^3 golden loop (g++ -std=c++11 -Ofast -fopenmp), {r['seconds']:.2f} s"}
    # fall back to this repo's CPU restatement
    import numpy as np
    from oracle import cpu
    if kind == "composite":
        p0, p1, vel = cpu.init_mt(n, n, n, nfields=3)
        s = np.zeros_like(p0)
        addr = np.random.default_rng(0).integers(0, n ** 3, 10000).astype(np.uint64)
        t0 = time.time()
        steps = 0
        while time.time() - t0 < budget_s and steps < 2000:
            cpu.main_step(p0, p1, vel, s, addr, order=cpu.ORDER_GOLDEN)
            p0, p1 = p1, p0
            steps += 1
        dt = time.time() - t0
        return {"value": round(float(n - 8) ** 3 * steps / dt * 1e-9, 3), "unit": "Gcell/s", "cores": threads, "kind": "port", "host": host,
                "sample": f"{steps} steps of the {n}^3 composite loop (oracle/fdtd_oracle.c, gcc -O2 strict IEEE), {dt:.2f} s"}
    if kind == "jacobi7":
        a = cpu.init_jacobi(n, n, n)
        b = a.copy()
    else:
        a, b = cpu.init_mt(n, n, n)
    cpu.run(kind, a, b, 1, M=radius, order=cpu.ORDER_GOLDEN)
    t0 = time.time()
    steps = 0
    while time.time() - t0 < budget_s and steps < 2000:
        cpu.run(kind, a, b, 2, M=radius, order=cpu.ORDER_GOLDEN)
        steps += 2
    dt = time.time() - t0
    cells = float(n - 2 * radius) ** 3 * steps
    return {"value": round(cells / dt * 1e-9, 3), "unit": "Gcell/s", "cores": threads, "kind": "port", "host": host,
            "sample": f"{steps} steps of the {n}^3 loop (oracle/fdtd_oracle.c, gcc -O2 strict IEEE), {dt:.2f} s"}


def spawn_ranks(n):
    """Start n fresh copies of this script, one per GPU, with the rendezvous variables torch.distributed
    reads; nothing in this (parent) process has touched a GPU.  Rank 0's stdout is relayed; the exit
    status is non-zero if any rank fails."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # a rank that dies (no such GPU, a failed launch) must not leave the others waiting at a rendezvous: watch them all,
    # and end the rest (the children this function started, by their own handles) when one fails
    import threading
    box = []
    reader = threading.Thread(target=lambda: box.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    while True:
        rcs = [p.poll() for p in procs]
        if all(c is not None for c in rcs):
            break
        if any(c not in (None, 0) for c in rcs):
            time.sleep(5.0)  # let the failing rank's message out, then stop the ranks still running
            for p in procs:
                if p.poll() is None:
                    p.kill()
            rcs = [p.wait() for p in procs]
            break
        time.sleep(0.2)
    reader.join(timeout=10.0)
    sys.stdout.write(box[0] if box else "")
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(rcs) if c != 0]
    if bad:
        raise SystemExit(f"bench.py: ranks failed (rank, exit status): {bad}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000, help="timed steps (1000: the drivers' own run, fdtd1.cu:68)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--variant", default="config3", choices=["config3"] + sorted(VARIANTS),
                    help="config3 (default): fdtd1..8 + fdtdspc + jacobi, headline = the slowest fdtdN")
    ap.add_argument("--size", dest="n", type=int, default=512, help="slab edge: every GPU owns n planes of n x n cells")
    ap.add_argument("--grid", default="",
                    help="Z,Y,X of a fixed global grid cut into z-slabs over the ranks (strong scaling, e.g. 2048,1024,1024); "
                         "sweep variants only; default: weak scaling with --size")
    ap.add_argument("--temporal", type=int, default=1,
                    help="time steps fused per launch where a fused kernel exists: 1 = deepest (default), 2, 3, 0 = none")
    ap.add_argument("--spinup-ms", type=float, default=100.0,
                    help="untimed device spin-up before the warm-up steps (fields restored afterwards); 0 = none")
    ap.add_argument("--reserve-cus", type=int, default=0,
                    help="N > 1, native exchange: CUs the sweep launches leave free for the exchange's kernels")
    ap.add_argument("--order", default="chain", choices=["chain", "golden"],
                    help="arithmetic order of the sweeps: chain = the device lambdas' (every + c*v one FMA), golden = the host loops' "
                         "(separate IEEE multiplies and adds: the order the reference's compiled loops pin bit for bit)")
    ap.add_argument("--shape", type=int, default=-1,
                    help="tuning: index of the single-step kernel's tile shape among the compiled ones (tools/tune.py lists them); -1 = the library's choice")
    ap.add_argument("--fuse-wide", action="store_true",
                    help="one rank: run radius 3 and 4 through their two-step kernels (slower than single steps; kept for measurement)")
    ap.add_argument("--no-strip", action="store_true",
                    help="tuning: rows wider than the fused kernels' tiles run their last few quads through the half-width tile instead of the strip shape")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-standing-start", dest="standing_start", action="store_false",
                    help="skip the extra 1000-step run from a standing start (one rank, sweep variants)")
    ap.add_argument("--all-variants", action="store_true", help="also time every other variant (N=1 only)")
    ap.add_argument("--exchange", default="ipc", choices=["native", "ipc", "torch"],
                    help="N > 1: who moves the ghost planes -- native: the library's own RCCL send/recv on its exchange stream "
                         "(fdtdx_run); ipc: the library's copy transport (neighbours' cubes mapped between the processes, planes pushed by "
                         "device copies, no exchange kernel on any CU; falls back to native); torch: torch.distributed P2P ops issued "
                         "from Python (fdtd_amd.slab.SlabRunner)")
    ap.add_argument("--rehearse-shared-gpu", action="store_true",
                    help="run the N>1 loop with every rank on cuda:0 and host-staged gloo exchange (1-GPU box rehearsal; "
                         "the line is marked rehearsal and is not a measurement)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(a.gpus)  # decided before anything here touches a GPU

    import torch
    import torch.distributed as dist
    from fdtd_amd import Stencil, make_program, ORDER_CHAIN, ORDER_GOLDEN
    from fdtd_amd.slab import HipSlabBackend, SlabRunner

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    a.gpus = world
    if a.rehearse_shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.rehearse_shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def spin_up(loop, tensors, block=30):
        """Keep the device busy with the hot path for --spinup-ms, then restore the fields (all on the stream)."""
        if a.spinup_ms <= 0:
            return
        if sum(t.numel() * t.element_size() for t in tensors) > 0.45 * torch.cuda.mem_get_info()[0]:
            a.spinup_ms = 0.0  # no room for the copies of the initial fields; steps on such grids take far longer than the ramp
            return
        saved = [t.clone() for t in tensors]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop(block)
        torch.cuda.synchronize()
        per_block = max(time.perf_counter() - t0, 1e-5)
        nblocks = torch.tensor([min(400.0, a.spinup_ms * 1e-3 / per_block)], dtype=torch.float64, device="cuda")
        if world > 1:  # every rank must run the same number of exchanges
            dist.all_reduce(nblocks, op=dist.ReduceOp.MAX)
        for _ in range(int(nblocks.item())):
            loop(block)
        for t, s in zip(tensors, saved):
            t.copy_(s)

    def time_main(steps, warmup):
        """The composite pipeline of main.cu on one rank: four launches per step."""
        import numpy as np
        from fdtd_amd import Stencil, make_composite
        n = a.n
        st = Stencil(n, n, n, 4, device=local_rank)
        bufs = {}
        for name in ("p0", "p1", "vel", "sum"):
            bufs[name] = torch.zeros(int(st.geometry.cube_elems), dtype=torch.float32, device=f"cuda:{local_rank}")
            st.attachCube(name, bufs[name].data_ptr(), bufs[name].numel() * 4)
        st.initMt("p0", "p1")
        st.initMt("vel", "sum")  # any field in [0,1) serves as vel; sum restarts from zero below
        bufs["sum"].zero_()
        addr = np.random.default_rng(0).integers(0, n ** 3, 10000).astype(np.uint64)  # main.cu:11,37-41
        st.mallocAddr("addr", len(addr))
        st.transferAddrToGPU("addr", addr)
        comp = make_composite()
        cs = torch.cuda.current_stream().cuda_stream

        def loop(k):
            s0, s1 = "p0", "p1"
            for _ in range(k):
                if a.temporal:  # product-accumulate folded into the update kernel
                    st.compositeStep(s0, s1, "vel", "sum", "addr", comp, 1.0, n // 2, n // 3, n // 4, 0.9, stream=cs)
                else:           # the reference's four launches (main.cu:160-163)
                    st.propagateComposite(s0, s1, "vel", comp, stream=cs)
                    st.inject(s0, "addr", 1.0, stream=cs)
                    st.filt(s0, n // 2, n // 3, n // 4, 0.9, stream=cs)
                    st.mulAcc("sum", s0, s1, 4, stream=cs)
                s0, s1 = s1, s0

        spin_up(loop, list(bufs.values()))
        loop(warmup)
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        loop(steps)
        e1.record()
        barrier()
        dt = time.perf_counter() - t0
        cells = float(n - 8) ** 3
        info = st.lastLaunchInfo()
        st.close()
        return dict(dt=dt, ev_ms=e0.elapsed_time(e1), cells_rank=cells, cells_all=cells, info=info, kind="composite", radius=4,
                    launches=steps)

    def time_yee(kind, steps, warmup):
        from fdtd_amd import Stencil, lib as flib
        n = a.n
        dev = f"cuda:{local_rank}"
        if kind == "yee3d":
            st = Stencil(n, n, n, 1, device=local_rank)
            names = ("ex", "ey", "ez", "hx", "hy", "hz")
            bufs = {}
            for i, name in enumerate(names):
                bufs[name] = torch.rand(int(st.geometry.cube_elems), dtype=torch.float32, device=dev,
                                        generator=torch.Generator(device=dev).manual_seed(i))
                st.attachCube(name, bufs[name].data_ptr(), bufs[name].numel() * 4)
            cs = torch.cuda.current_stream().cuda_stream
            if a.temporal:  # one launch per step, out of place between two sets of six cubes (48 B per cell)
                names_b = tuple(n_ + "_b" for n_ in names)
                for n_, nb in zip(names, names_b):
                    bufs[nb] = bufs[n_].clone()
                    st.attachCube(nb, bufs[nb].data_ptr(), bufs[nb].numel() * 4)
                sets, done = (names, names_b), [0]

                def loop(k):
                    for _ in range(k):
                        t = done[0]
                        st.yee3dStepFused(sets[t % 2], sets[(t + 1) % 2], 0.31, 0.27, stream=cs)
                        done[0] = t + 1
                info = "yee3d_fused_kernel, one launch per step (H half + E half, out of place)"
            else:
                loop = lambda k: [st.yee3dStep(names, 0.31, 0.27, stream=cs) for _ in range(k)]
                info = "yee3d_half_kernel x2 per step (H half, E half)"
            cells = float(n - 2) ** 3
            fields = list(bufs.values())
        else:
            ex, ey, hz = (torch.rand(n * n, dtype=torch.float32, device=dev) for _ in range(3))
            cs = torch.cuda.current_stream().cuda_stream
            loop = lambda k: flib.yee2d_steps(ex.data_ptr(), ey.data_ptr(), hz.data_ptr(), n, n, 0, k, 0.5, 0.7, device=local_rank, stream=cs)
            cells = float(n) * float(n)
            info = "yee2d_e_kernel + yee2d_h_kernel per step"
            fields = [ex, ey, hz]
        spin_up(loop, fields, block=100 if kind == "yee2d" else 30)
        loop(warmup)
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        loop(steps)
        e1.record()
        barrier()
        dt = time.perf_counter() - t0
        nl = steps if (kind == "yee3d" and a.temporal) else 2 * steps
        return dict(dt=dt, ev_ms=e0.elapsed_time(e1), cells_rank=cells, cells_all=cells, info=info, kind=kind, radius=1, launches=nl)

    def time_yee_slabs(steps, warmup):
        """The E/H extension on z-slabs, one per rank (fdtd_amd/slab.py:YeeRunner): two half-step launches per rank and step,
        after each half two fields move one plane to one neighbour.  Random fields per rank: the ghost planes do not match
        the neighbours' at the start, which the timing does not see."""
        from fdtd_amd.slab import YeeRunner
        n = a.n
        Zg = n * world
        be = HipSlabBackend(Zg, n, n, 1, rank, world, local_rank)
        names = ("ex", "ey", "ez", "hx", "hy", "hz")
        for i, name in enumerate(names):
            be.alloc(name).uniform_(0.0, 1.0, generator=torch.Generator(device=be.device).manual_seed(6 * rank + i))
        mode = "host" if a.rehearse_shared_gpu else "device"
        if a.exchange in ("native", "ipc") and not a.rehearse_shared_gpu:  # (the copy transport moves whole ghost blocks of one cube: not these half steps)
            # the library's own communicator, as for the sweeps; every rank takes the same path
            ok = torch.ones(1, device="cuda")
            box = [None]
            if rank == 0:
                try:
                    box = [Stencil.commUniqueId()]
                except Exception as e:
                    print(f"rank 0: no RCCL id ({e})", file=sys.stderr)
            dist.broadcast_object_list(box, src=0)
            try:
                if box[0] is None:
                    raise RuntimeError("rank 0 could not make a communicator id")
                be.st.commInit(box[0])
            except Exception as e:
                print(f"rank {rank}: native exchange unavailable ({e})", file=sys.stderr)
                ok.zero_()
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            mode = "native" if ok.item() != 0 else "device"
        run = YeeRunner(be, names, rank, world, exchange=mode)
        loop = lambda k: run.run(k, 0.31, 0.27)
        spin_up(loop, [be.bufs[nm] for nm in names])
        loop(warmup)
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        loop(steps)
        e1.record()
        barrier()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        cells_rank = float(run.z_hi - run.z_lo) * float(n - 2) ** 2
        be.st.close()
        return dict(dt=float(t.item()), ev_ms=e0.elapsed_time(e1), cells_rank=cells_rank, cells_all=float(Zg - 2) * float(n - 2) ** 2,
                    info="yee3d_half_kernel x2 per step and rank; Hx, Hy one plane up after the H half, Ex, Ey one plane down after the E half",
                    kind="yee3d", radius=1, launches=4 * steps, bpu=72.0, slabs=True, exchange=mode)

    def time_variant(name, steps, warmup):
        kind, radius, _ = VARIANTS[name]
        if kind in ("yee3d", "yee2d"):
            if world != 1 and kind == "yee2d":
                raise SystemExit("--variant yee2d runs on one rank")
            return time_yee_slabs(steps, warmup) if world != 1 else time_yee(kind, steps, warmup)
        if kind == "composite":
            if world != 1:
                raise SystemExit("--variant main runs on one rank")
            return time_main(steps, warmup)
        n = a.n
        # ghost depth: the program radius, or, on several slabs, what the fused kernels need to run with one
        # exchange per launch: 3 for radius-1 programs (three steps), 4 for radius 2 (two steps)
        use_ipc = world > 1 and a.exchange == "ipc"  # works between processes that share a GPU too
        native = world > 1 and a.exchange in ("native", "ipc") and not a.rehearse_shared_gpu
        ghost = radius
        if world > 1 and a.temporal != 0:
            if kind in ("star", "jacobi7") and radius == 1:
                ghost = 3
            elif kind == "star" and radius == 2:
                ghost = 4  # two steps per launch through a third cube (fdtdx_run_work / the Python loop's scratch cube)
            elif kind == "spc13":
                ghost = 2
        Zg, Yg, Xg = ([int(v) for v in a.grid.split(",")] if a.grid else [n * world, n, n])
        be = HipSlabBackend(Zg, Yg, Xg, ghost, rank, world, local_rank)
        be.alloc("p0")
        be.alloc("p1")
        if kind == "jacobi7":
            be.st.initJacobi("p0")
            be.st.initJacobi("p1")
        else:
            be.st.initMt("p0", "p1")
        prog = make_program(kind, radius=radius, order=ORDER_GOLDEN if a.order == "golden" else ORDER_CHAIN)
        be.st.setOption("temporal", a.temporal)
        if a.no_strip:
            be.st.setOption("no_strip", 1)
        if a.shape >= 0:
            be.st.setOption("variant", a.shape)
        # the two-step loops that hop through a third cube (radius 2; radius 1 under --temporal 2) get it from here, by name:
        # nothing is allocated inside the timed loop
        work = None
        if world == 1 and a.temporal != 0 and kind == "star" and (radius == 2 or (radius == 1 and a.temporal == 2)):
            be.alloc("work")
            work = "work"
        if world == 1 and kind == "jacobi7" and a.temporal == 2:
            be.alloc("work")
            work = "work"
        if world == 1 and a.fuse_wide and a.temporal != 0 and kind == "star" and radius in (3, 4):
            be.st.setOption("fuse_wide", 1)
            be.alloc("work")
            work = "work"
        if (native or use_ipc) and a.temporal != 0 and kind == "star" and radius == 2:
            be.alloc("work")
            work = "work"
        run = SlabRunner(be, ghost, prog, rank, world, exchange="host" if a.rehearse_shared_gpu else "device")
        cs = torch.cuda.current_stream().cuda_stream
        flag_dev = "cpu" if a.rehearse_shared_gpu else "cuda"  # where the ranks' agreement flags live (gloo / nccl process group)
        if use_ipc:
            # the copy transport: every rank exports the cubes it exchanges, torch.distributed carries the descriptors around,
            # every rank maps its neighbours' cubes; all ranks take this path or none does
            ok = torch.ones(1, device=flag_dev)
            desc = None
            try:
                desc = be.st.commIpcExport(["p0", "p1"] + (["work"] if work else []))
            except Exception as e:
                print(f"rank {rank}: IPC export failed ({e})", file=sys.stderr)
                ok.zero_()
            descs = [None] * world
            dist.all_gather_object(descs, desc)
            if ok.item() != 0 and all(d is not None for d in descs):
                try:
                    if os.environ.get("FDTDX_BENCH_FAIL_CONNECT") and rank == world - 1:
                        raise RuntimeError("failure asked for (FDTDX_BENCH_FAIL_CONNECT), to rehearse the fallback")
                    be.st.commIpcConnect(descs[rank - 1] if rank > 0 else None, descs[rank + 1] if rank < world - 1 else None)
                except Exception as e:
                    print(f"rank {rank}: IPC connect failed ({e})", file=sys.stderr)
                    ok.zero_()
            else:
                ok.zero_()
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            dist.barrier()
            if ok.item() == 0:
                use_ipc = False
                be.st.commDestroy()
                a.exchange = "native" if native else "torch"
            else:
                native = False
        if native:
            # the library's own communicator: rank 0 makes the id, torch.distributed carries its 128 bytes to the others
            ok = torch.ones(1, device="cuda")
            box = [None]
            if rank == 0:
                try:
                    box = [Stencil.commUniqueId()]
                except Exception as e:
                    print(f"rank 0: no RCCL id ({e})", file=sys.stderr)
            dist.broadcast_object_list(box, src=0)
            try:
                if box[0] is None:
                    raise RuntimeError("rank 0 could not make a communicator id")
                be.st.commInit(box[0])
                if a.reserve_cus:
                    be.st.setOption("reserve_cus", a.reserve_cus)
            except Exception as e:  # every rank must take the same path: agree below
                print(f"rank {rank}: native exchange unavailable ({e})", file=sys.stderr)
                ok.zero_()
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if ok.item() == 0:
                native = False
                a.exchange = "torch"

        if native or use_ipc:
            # Before anything is timed: does the transport deliver?  Every rank writes numbers that name it, the side and the plane
            # into the planes its neighbours are due, exchanges once through the library and looks at what arrived in its ghost
            # planes; the planes are then put back.  All ranks agree on the outcome; a transport that fails is replaced by the next.
            g = be.g
            glo, ghi, L = int(g.ghost_lo), int(g.ghost_hi), int(g.local_planes)
            regions = [(0, glo), (glo, ghost), (L - ghi - ghost, ghost), (L - ghi, ghi)]
            saved = [be.planes("p0", z0, cnt).clone() for z0, cnt in regions]
            pl = int(g.plane)

            def pattern(r, side, cnt, rnd):
                return (torch.arange(cnt, device="cuda", dtype=torch.float32) + (1000.0 * r + 100.0 * side + 10000.0 * rnd)).repeat_interleave(pl)

            ok = torch.ones(1, device=flag_dev)
            err = None
            # two rounds: the second one arrives in ghost planes this device has just read (the comparison of the first), i.e.
            # over lines its caches may still hold.  A rank that has failed keeps taking part in the barriers.
            for rnd in (0, 1):
                try:
                    if err is None:
                        if glo:
                            be.planes("p0", glo, ghost).copy_(pattern(rank, 0, ghost, rnd))
                        if ghi:
                            be.planes("p0", L - ghi - ghost, ghost).copy_(pattern(rank, 1, ghost, rnd))
                        torch.cuda.synchronize()
                        be.st.commCubeHalo("p0", cs)
                        torch.cuda.synchronize()
                        if glo and not torch.equal(be.planes("p0", 0, glo), pattern(rank - 1, 1, glo, rnd)):
                            err = f"round {rnd}: the ghost planes below do not hold what the neighbour below sent"
                        elif ghi and not torch.equal(be.planes("p0", L - ghi, ghi), pattern(rank + 1, 0, ghi, rnd)):
                            err = f"round {rnd}: the ghost planes above do not hold what the neighbour above sent"
                except Exception as e:
                    err = str(e)
                dist.barrier()  # nobody overwrites what a neighbour is still comparing
            if err is None and os.environ.get("FDTDX_BENCH_FAIL_PROBE") == a.exchange and rank == world - 1:
                err = "failure asked for (FDTDX_BENCH_FAIL_PROBE), to rehearse the fallback"
            if err is not None:
                print(f"rank {rank}: exchange probe failed over {a.exchange} ({err})", file=sys.stderr)
                ok.zero_()
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            dist.barrier()
            torch.cuda.synchronize()
            for (z0, cnt), t in zip(regions, saved):
                if cnt:
                    be.planes("p0", z0, cnt).copy_(t)
            torch.cuda.synchronize()
            dist.barrier()
            del saved
            if ok.item() == 0:
                be.st.close()  # the communicator goes with the context
                del be, run
                torch.cuda.empty_cache()
                a.exchange = "native" if a.exchange == "ipc" else "torch"
                return time_variant(name, steps, warmup)

        def loop(k):
            if world == 1:
                # one rank: the whole time loop runs inside the library (fdtdx_sweep, the drivers'
                # loop of fdtd1.cu:68-80), which fuses two steps per launch where it can
                be.st.sweep("p0", "p1", prog, k, cs, work=work)
            elif native or use_ipc:
                Stencil.run([be.st], "p0", "p1", prog, k, [cs], work=work)
            else:
                run.run("p0", "p1", k, fuse=a.temporal != 0)

        def launches():
            return int(be.st.lastLaunchInfo().rsplit("launches=", 1)[1])

        spin_up(loop, [be.bufs["p0"], be.bufs["p1"]])
        loop(warmup)
        barrier()
        l0 = launches()  # sweep launches this rank has issued so far
        stream = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        loop(steps)
        e1.record(stream)
        barrier()
        dt = time.perf_counter() - t0
        ev_ms = e0.elapsed_time(e1)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        cells_rank = float(run.z_hi - run.z_lo) * float(Yg - 2 * radius) * float(Xg - 2 * radius)   # this rank, one step
        cells_all = float(Zg - 2 * radius) * float(Yg - 2 * radius) * float(Xg - 2 * radius)
        info = be.st.lastLaunchInfo()
        nl = launches() - l0
        res = dict(dt=dt, ev_ms=ev_ms, cells_rank=cells_rank, cells_all=cells_all, info=info, kind=kind, radius=radius, launches=nl)
        if world == 1:
            # The dominant kernel by itself: a run of launches of the kernel the loop spends its time in (a fused loop
            # closes with one to four single steps, which are another kernel), bracketed by HIP events on the launch stream.
            k = 3 if "fused3" in info else 2 if "fused2" in info else 1
            lo, hi = run.z_lo, run.z_hi
            nprobe = max(6, min(60, steps // k))

            def probe(cnt):
                # the hops of the library's own loop: between the two cubes, or (two-step loops with a third cube) between p1 and it
                o, i = ("work", "p1") if (work and k == 2) else ("p0", "p1")
                if k == 1:  # single steps: the library's loop itself (it keeps strips with the shell components of partly stored quads beside the cubes)
                    be.st.sweep("p0", "p1", prog, cnt, cs, work=work)
                    return
                for _ in range(cnt):
                    if k == 1:
                        be.st.propagatePlanes(o, i, prog, lo, hi, cs)
                    else:
                        be.st.propagateFusedPlanes(o, i, "p0" if (work and k == 2) else o, prog, k, lo, hi, cs)
                    o, i = i, o
            if k == 1:
                # a loop of single steps IS a run of launches of its one kernel: the events around the timed loop give the launch duration
                res.update(kernel_us=ev_ms * 1e3 / max(nl, 1), kernel_steps=1, kernel_launches=nl, kernel_src="HIP events around the timed loop")
            else:
                try:
                    probe(8)
                    e0.record(stream)
                    probe(nprobe)
                    e1.record(stream)
                    torch.cuda.synchronize()
                    ku, src = e0.elapsed_time(e1) * 1e3 / nprobe, "HIP events around a closing run of the fused kernel alone"
                    cap = ev_ms * 1e3 / steps * k  # the loop's own average per k steps (its closing single steps are slower than a fused launch)
                    if ku > cap:
                        ku, src = cap, "timed loop average (the closing run of the fused kernel was slower than the loop it describes)"
                    res.update(kernel_us=ku, kernel_steps=k, kernel_launches=nprobe, kernel_src=src)
                except Exception:  # no such launch for this program: the loop average stands in
                    pass
            if a.standing_start:
                # the drivers' own run: 1000 steps from a standing start (fdtd1.cu:68), device clocks not yet settled
                torch.cuda.synchronize()
                time.sleep(1.0)
                t0 = time.perf_counter()
                loop(1000)
                torch.cuda.synchronize()
                res["standing_start_gcells"] = cells_all * 1000 / (time.perf_counter() - t0) * 1e-9
        be.st.close()
        del be
        torch.cuda.empty_cache()
        return res

    tf = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        traffic_db = json.load(open(tf))
    except Exception:
        traffic_db = {}

    def summarise(name, r, steps):
        """Per-variant figures: throughput of the timed loop and the roofline position of its dominant kernel."""
        value = r["cells_all"] * steps / r["dt"] * 1e-9
        # bytes per cell-step: sweeps 8; main 24 (one fused step); yee3d 48 in one launch (72 as two half steps), yee2d 36 as two launches
        bpu = {"composite": BYTES_PER_UPDATE_MAIN, "yee3d": 48.0 if a.temporal else 72.0, "yee2d": 36.0}.get(r["kind"], BYTES_PER_UPDATE)
        bpu = r.get("bpu", bpu)
        src = "HIP events around the timed loop, averaged over its launches"
        if "kernel_us" in r:
            launch_s, spl, navg = r["kernel_us"] * 1e-6, float(r["kernel_steps"]), r["kernel_launches"]
            src = r.get("kernel_src", src)
        else:
            # several ranks (a step is split into boundary-plane and interior launches of the same kernel) and the
            # pipelines with several kernels per step: average over the launches of the timed region
            nlaunch = r["launches"] or steps
            launch_s, spl, navg = r["ev_ms"] * 1e-3 / nlaunch, steps / nlaunch, nlaunch
        alg = bpu * r["cells_rank"] * spl
        achieved = alg / launch_s * 1e-9
        traffic = None
        key = name + ("_singlestep" if a.temporal == 0 and VARIANTS[name][1] == 1 and name not in ("fdtdspc", "yee3d", "yee2d") else "")
        ent = traffic_db.get(f"{key}_n{a.n}")
        # the profiled figure is per whole-slab launch of one named shape: it does not describe split launches on several
        # ranks, other grids, or a kernel whose shape has changed since it was profiled
        shape_now = r["info"].rsplit(" launches=", 1)[0]
        if ent and world == 1 and not a.grid and ent.get("launch_info") in (None, shape_now):
            traffic = ent.get("hbm_bytes_per_launch")
        out = {"gcells": round(value, 2), "ms_per_step": round(r["dt"] / steps * 1e3, 5), "launch_us": round(launch_s * 1e6, 2),
               "steps_per_launch": round(spl, 3), "launches": navg, "achieved": round(achieved, 1), "frac": round(achieved / PEAK_HBM_GBS, 4),
               "algorithmic_bytes_per_launch": alg, "launch_us_source": src, "traffic": traffic,
               "traffic_source": "profiles/traffic.json: rocprofv3 FETCH_SIZE/WRITE_SIZE passes of this kernel shape in an earlier run, not measured in this one" if traffic else None,
               "hbm_achieved": round(traffic / launch_s * 1e-9, 1) if traffic else None,
               "hbm_frac": round(traffic / launch_s * 1e-9 / PEAK_HBM_GBS, 4) if traffic else None,
               "kernel": shape_now}
        if "standing_start_gcells" in r:
            out["standing_start_1000_steps_gcells"] = round(r["standing_start_gcells"], 2)
        return out

    if a.variant == "config3":
        names = [f"fdtd{m}" for m in range(1, 9)] + ["fdtdspc", "jacobi"]
        table, raw = {}, {}
        for name in names:
            raw[name] = time_variant(name, a.steps, a.warmup)
            table[name] = summarise(name, raw[name], a.steps)
        head = min(names[:8], key=lambda v: table[v]["gcells"])  # the slowest radius carries the line
    else:
        names, head = [a.variant], a.variant
        raw = {head: time_variant(head, a.steps, a.warmup)}
        table = {head: summarise(head, raw[head], a.steps)}
    r, t = raw[head], table[head]

    grid = [a.n, a.n] if r["kind"] == "yee2d" else [a.n * world, a.n, a.n]
    fixed = bool(a.grid) and r["kind"] in ("star", "spc13", "jacobi7")
    if fixed:
        grid = [int(v) for v in a.grid.split(",")]
    if r["kind"] == "yee3d":
        workload = (f"yee3d (extension): six-field E/H step = H half step + E half step "
                    f"({'one fused launch' if (a.temporal and not r.get('slabs')) else 'two launches'}), "
                    + (f"{a.n * world}x{a.n}x{a.n} fp32 grid in {world} z-slabs" if r.get("slabs") else f"{a.n}^3 fp32 grid") + f", {a.steps} steps")
    elif r["kind"] == "yee2d":
        workload = f"yee2d (extension): ex/ey/hz step on a {a.n}x{a.n} fp32 grid, {a.steps} steps"
    else:
        workload = (f"{head}: {VARIANTS[head][0]} radius {r['radius']} sweep ({VARIANTS[head][2]}), "
                    f"{grid[0]}x{grid[1]}x{grid[2]} fp32 grid, {a.steps} steps, "
                    + (f"cut into {world} z-slabs" if fixed else f"z-slabs of {a.n} planes per GPU"))
        if a.variant == "config3":
            workload = ("fdtd1..fdtd8 (star radius 1..8, fdtd1.cu .. fdtd8.cu) each timed on this grid; value = the slowest, " + workload)
    # with the drivers' coefficients the field grows by 0.01 + 0.27 M per step: from radius 5 up every value is inf after this many steps
    overflow_after = {5: 290, 6: 183, 7: 139, 8: 115}.get(r["radius"]) if r["kind"] == "star" else None
    line = {
        "metric": "cell-updates/sec", "value": t["gcells"], "unit": "Gcell/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": t["ms_per_step"],
        "higher_is_better": True, "scaling": "strong" if fixed else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "grid": grid, "radius": r["radius"], "parallelism": f"zslab{world}", "order": a.order, "kernel": t["kernel"],
                   "spinup_ms": a.spinup_ms},
        "roofline": {"bound": "hbm", "achieved": t["achieved"], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": t["frac"],
                     "traffic": t["traffic"], "hbm_achieved": t["hbm_achieved"], "hbm_frac": t["hbm_frac"],
                     "algorithmic_bytes_per_launch": t["algorithmic_bytes_per_launch"], "steps_per_launch": t["steps_per_launch"],
                     "launch_us": t["launch_us"], "launches": t["launches"], "launch_us_source": t["launch_us_source"],
                     "traffic_source": t["traffic_source"]},
    }
    if overflow_after:
        line["config"]["values"] = (f"finite throughout the timed steps" if a.steps + a.warmup <= overflow_after else
                                    f"overflow to inf after about {overflow_after} steps (fdtd{r['radius']}.cu's coefficients): most timed sweeps run on "
                                    f"infinities, which is up to 8 % faster than on finite numbers (higher clock); --steps 20 times finite fields")
    if len(names) > 1:
        line["variants"] = table
    if a.all_variants and world == 1 and a.variant != "config3":
        others = {}
        for name in sorted(VARIANTS):
            if name == a.variant or name in ("main", "yee3d", "yee2d"):
                continue
            o = time_variant(name, min(a.steps, 200), 5)
            others[name] = round(o["cells_all"] * min(a.steps, 200) / o["dt"] * 1e-9, 1)
        line["other_variants_gcells"] = others
    if r["kind"] in ("yee3d", "yee2d"):
        line["extension"] = "no counterpart in the reference: defined by oracle/fdtd_oracle.c, parity unpinned"
    if rank == 0 and world == 1 and not a.no_cpu and r["kind"] not in ("yee3d", "yee2d"):
        try:
            line["cpu_baseline"] = cpu_baseline(r["kind"], r["radius"], a.n)
        except Exception as e:  # the GPU number stands on its own
            line["cpu_baseline"] = {"value": None, "unit": "Gcell/s", "cores": 0, "kind": "reference",
                                    "sample": f"failed: {e}"}
    if a.rehearse_shared_gpu:
        line["rehearsal"] = ("ranks share one GPU" + (", planes copied between the processes' mapped cubes (IPC)" if a.exchange == "ipc" else ", host-staged gloo exchange")
                             + ": NOT a measurement")
    if world > 1:
        who = [None] * world
        dist.all_gather_object(who, {"rank": rank, "device": local_rank, "name": torch.cuda.get_device_name(local_rank)})
        line["ranks"] = who
        line["config"]["exchange"] = ("host-staged gloo (rehearsal)" if (a.rehearse_shared_gpu and a.exchange != "ipc") else
                                      "device copies into the neighbours' mapped cubes (IPC), issued by the library on one copy stream per direction, hosts pacing; checked by a probe exchange before the timed steps" if a.exchange == "ipc" else
                                      "RCCL send/recv issued by the library on its exchange stream" if (a.exchange == "native" and r.get("exchange", "native") == "native")
                                      else "torch.distributed batch_isend_irecv on the nccl (= RCCL) backend")
        line["config"]["world_size"] = dist.get_world_size()
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
