"""What cutting the E/H step into z-slabs costs on one GPU: the six-field half-step loop on one n^3 grid against the same grid
as R slabs of this process (fdtd_amd/slab.py:YeeRunner, exchange="local": boundary plane first, device copies into the
neighbour's ghost plane, the rest of the planes after).  Run on the GPU box: python tools/yee_slabs_cost.py [n] [steps]."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdtd_amd.slab import HipSlabBackend, YeeRunner  # noqa: E402

NAMES = ("ex", "ey", "ez", "hx", "hy", "hz")


def timed(runs, steps):
    def loop(k):
        for _ in range(k):
            for which, coef in ((0, 0.27), (1, 0.31)):
                for r in runs:
                    r.half(which, coef)
    loop(steps)  # clocks up, caches of launch parameters filled
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(steps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    for R in (1, 2, 4, 8):
        bes = [HipSlabBackend(n, n, n, 1, r, R, 0) for r in range(R)]
        for be in bes:
            for i, nm in enumerate(NAMES):
                be.alloc(nm).uniform_(0.0, 1.0, generator=torch.Generator(device=be.device).manual_seed(i))
        peers = dict(enumerate(bes))
        runs = [YeeRunner(be, NAMES, r, R, exchange="local", peers=peers) for r, be in enumerate(bes)]
        ms = timed(runs, steps)
        cells = float(n - 2) ** 3
        print(f"{n}^3 in {R} slab(s): {ms:.3f} ms per step, {cells / ms * 1e-6:.1f} Gcell-steps/s", flush=True)
        for be in bes:
            be.st.close()
        del bes, runs, peers
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
