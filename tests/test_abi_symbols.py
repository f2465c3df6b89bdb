"""The C-ABI library loads and exports every symbol include/fdtdx.h declares (no GPU needed)."""
import os
import re

from fdtd_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "fdtdx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fdtdx_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(lib.SYMBOLS)


def test_library_exports_every_symbol():
    L = lib.load()
    for name in declared_symbols():
        assert hasattr(L, name), name


def test_decompose_is_host_only():
    g = lib.decompose(512, 512, 512, 4, rank=1, nprocs=4)
    assert g.pitch_x == 544 and g.owned_planes == 128 and g.ghost_lo == 4 and g.ghost_hi == 4
    assert g.first_global == 124 and g.local_planes == 136
