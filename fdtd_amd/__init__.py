"""fdtd_amd -- MI355X (gfx950) stencil sweep engine behind the reference's stencil.h interface.

  lib      ctypes binding of libfdtdx.so (the C ABI of include/fdtdx.h)
  stencil  `Stencil`: Python face of the reference's host class, same method names
  slab     z-slab time loop with ghost-plane exchange over torch.distributed (RCCL / gloo)
"""
from .lib import (FdtdxError, make_program, make_composite, decompose, STAR, SPC13, JACOBI7, ORDER_CHAIN,  # noqa: F401
                  ORDER_GOLDEN)
from .stencil import Stencil  # noqa: F401
