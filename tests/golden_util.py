"""Loader of tests/golden/golden_v1.npz (fields produced by the reference's compiled CPU loops)."""
import os

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v1.npz")


def cases():
    z = np.load(PATH, allow_pickle=False)
    for idx, row in enumerate(z["meta"]):
        kind, M, nz, ny, nx, steps, c0, c1 = row
        yield dict(idx=idx, kind=str(kind), M=int(M), shape=(int(nz), int(ny), int(nx)), steps=int(steps),
                   crc=(int(c0), int(c1)), a=z[f"c{idx}_a"], b=z[f"c{idx}_b"])
