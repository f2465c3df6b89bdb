"""Generates tests/golden/golden_main_v1.npz from the reference's OWN composite loop (main.cu:54-84).

Run in the build container only (needs /root/reference; oracle/build_ref.sh compiles main.cu:24-36
and :54-84 with -O2 -ffp-contract=off into oracle/_ref/libfdtd_ref_strict.so).  The fixture holds
data only: grid size, step count, the injection addresses used (main.cu:37-41 draws them from an
unseeded rand(); here a seeded numpy generator, with one address repeated on purpose), CRC32s of the
reference's initial fields, and the three fields the reference loop leaves behind.
"""
import ctypes
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cpu  # noqa: E402

SHAPE, STEPS, NA = (22, 19, 26), 6, 300


def main():
    assert cpu.ref_available(), "run `make -C oracle ref` where /root/reference exists"
    nz, ny, nx = SHAPE
    p0, p1, vel, s = (np.empty(SHAPE, np.float32) for _ in range(4))
    cpu.ref().ref_init_mt3(cpu._p(p0), cpu._p(p1), cpu._p(vel), cpu._p(s), nz, ny, nx)
    crc = [zlib.crc32(a.tobytes()) for a in (p0, p1, vel)]
    addr = np.random.default_rng(19950918).integers(0, nz * ny * nx, NA).astype(np.uint64)
    addr[17] = addr[3]
    for _ in range(STEPS):
        cpu.ref_main_step(p0, p1, vel, s, addr)
        p0, p1 = p1, p0
    path = os.path.join(ROOT, "tests", "golden", "golden_main_v1.npz")
    np.savez_compressed(path, shape=np.array(SHAPE), steps=np.array(STEPS), addr=addr, crc=np.array(crc, np.uint64),
                        p0=p0, p1=p1, sum=s)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
