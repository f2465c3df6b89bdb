"""Times every compiled tile shape of a sweep at full size on the GPU (tuning aid, not a test).

usage: python tools/tune.py [--kinds star,spc13,jacobi7] [--radii 1,2,..] [--n 512] [--steps 20]
Prints one line per (kind, radius, variant, zchunks) with Gcell/s; inputs are the drivers'
mt19937 fields (fdtd1.cu:21-31) generated once per radius.
The sweeps run against the power limit, so their speed depends on the values in the cubes, and shapes rank differently on
the normalised fields this tool sweeps by default (--bounded 1) than on the drivers' own first steps, which is what
bench.py --steps 20 times: a shape that leads here by 7 % can be level there.  Take the decision from tools/shape_scan.sh /
tools/lib_shape_scan.sh (bench.py per shape on the drivers' fields); use this tool to list the shapes and for a first ranking.
"""
import argparse
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fdtd_amd import Stencil, make_program, FdtdxError


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kinds", default="star")
    ap.add_argument("--radii", default="1,2,3,4,5,6,7,8")
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--zchunks", default="0")
    ap.add_argument("--out", default="")
    ap.add_argument("--temporal", type=int, default=0, help="1: let sweep() fuse time steps")
    ap.add_argument("--tb-variant", type=int, default=-1, help="shape index of the fused kernel (-1: default)")
    ap.add_argument("--tb-zchunks", type=int, default=0, help="z chunks of the fused kernel (0: the library's choice)")
    ap.add_argument("--variants", default="", help="comma list of variant indices (default: all)")
    ap.add_argument("--bounded", type=int, default=1,
                    help="1 (default): star coefficients are divided by their sum (0.01 + 0.27 M), so the fields stay finite however many steps "
                         "run; 0: the drivers' coefficients, whose fields overflow to inf within 100-300 steps at radius 5-8 -- and sweeps of "
                         "inf run up to 8 %% faster than sweeps of finite numbers (less switching, higher clocks)")
    a = ap.parse_args()
    n = a.n
    stream = torch.cuda.current_stream().cuda_stream
    results = []
    for kind in a.kinds.split(","):
        radii = [int(r) for r in a.radii.split(",")] if kind == "star" else [1]
        for M in radii:
            with Stencil(n, n, n, M) as st:
                st.mallocCube("p0")
                st.mallocCube("p1")
                st.initMt("p0", "p1")
                st.setOption("temporal", a.temporal)
                st.setOption("tb_variant", a.tb_variant)
                st.setOption("tb_zchunks", a.tb_zchunks)
                prog = make_program(kind, radius=M)
                if a.bounded and kind == "star":
                    for i in range(13):
                        prog.coef[i] /= (0.01 + 0.27 * M)
                cells = float(n - 2 * M) ** 3
                vlist = [int(x) for x in a.variants.split(',')] if a.variants else None
                # the device needs 15-30 ms of continuous work after the idle spell of initMt before its
                # clocks settle; without this the first shapes timed look up to 1.6x slower than they are
                for _ in range(8):
                    st.sweep("p0", "p1", prog, 60, stream)
                torch.cuda.synchronize()
                v = 0
                while True:
                    if vlist is not None:
                        if not vlist:
                            break
                        v = vlist.pop(0)
                    st.setOption("variant", v)
                    ok = True
                    for zc in [int(z) for z in a.zchunks.split(",")]:
                        st.setOption("zchunks", zc)
                        try:
                            st.sweep("p0", "p1", prog, 2, stream)
                        except FdtdxError:
                            ok = False
                            break
                        torch.cuda.synchronize()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        st.sweep("p0", "p1", prog, a.steps, stream)
                        e1.record()
                        torch.cuda.synchronize()
                        ms = e0.elapsed_time(e1) / a.steps
                        gc = cells / ms * 1e-6
                        info = st.lastLaunchInfo()
                        print(f"{kind} M={M} v={v} zc={zc}: {ms*1e3:8.1f} us/step {gc:8.1f} Gcell/s  [{info}]", flush=True)
                        results.append(dict(kind=kind, M=M, variant=v, zchunks=zc, us=ms * 1e3, gcells=gc, info=info))
                    if not ok:
                        break
                    v += 1
    if a.out:
        json.dump(results, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
