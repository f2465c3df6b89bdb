"""Generates tests/golden/golden_v1.npz from the reference's OWN CPU golden loops.

Run in the build container only (needs /root/reference; oracle/build_ref.sh compiles the loops
of fdtd1..8.cu:42-70, fdtdspc.cu:42-52, jacobi.cu:475-485 and the initialisers fdtd1.cu:21-31,
jacobi.cu:412-417 with -O2 -ffp-contract=off into oracle/_ref/libfdtd_ref_strict.so).

The fixture holds DATA only: grid sizes, step counts and the fields the reference loops
produce.  Inputs are not stored: they are the reference's deterministic initial fields and the
file stores CRC32s of them so that a test can prove its own generator reproduces them.
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cpu  # noqa: E402

CASES = [("star", M, (2 * M + 6, 2 * M + 9, 2 * M + 13), 5) for M in range(1, 9)]
CASES += [("star", 1, (12, 20, 70), 12), ("star", 4, (20, 24, 40), 2), ("spc13", 1, (9, 14, 37), 6),
          ("jacobi7", 1, (10, 13, 35), 7)]


def main():
    assert cpu.ref_available(), "run `make -C oracle ref` where /root/reference exists"
    out = {}
    meta = []
    for idx, (kind, M, (nz, ny, nx), steps) in enumerate(CASES):
        if kind == "jacobi7":
            a, b = cpu.ref_init_jacobi(nz, ny, nx)
        else:
            a, b = cpu.ref_init_mt(nz, ny, nx)
        crc = (zlib.crc32(a.tobytes()), zlib.crc32(b.tobytes()))
        cpu.ref_run(kind, a, b, steps, M=M)
        out[f"c{idx}_a"] = a
        out[f"c{idx}_b"] = b
        meta.append((kind, M, nz, ny, nx, steps, crc[0], crc[1]))
    out["meta"] = np.array([(k, str(M), str(nz), str(ny), str(nx), str(s), str(c0), str(c1))
                            for (k, M, nz, ny, nx, s, c0, c1) in meta])
    path = os.path.join(ROOT, "tests", "golden", "golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(meta), "cases")


if __name__ == "__main__":
    main()
