// tb_r3_star.hip -- the fused (two steps per launch) radius-3 and radius-4 star sweeps (fdtd3.cu:69-72, fdtd4.cu:72-76).
// Behind the "fuse_wide" option only: with the z queues of both levels in registers a wave holds at most three rows of 64
// quads (radius 3: 168 of its 256 registers for the queues; radius 4: 216), so a workgroup of 8 waves covers 24 rows and
// stores 24 - 4 x radius of them, 60 of every 64 quads.
#include "tbm_kernel.h"
using namespace fdtdx;
FDTDX_TBM_VARIANT(3, 8, 3, 1, 0)
FDTDX_TBM_VARIANT(4, 8, 3, 1, 0)
