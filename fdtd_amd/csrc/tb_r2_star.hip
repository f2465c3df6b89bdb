// tb_r2_star.hip -- shapes of the fused (two steps per launch) radius-2 star sweep (fdtd2.cu:66-69).
#include "tbm_kernel.h"
using namespace fdtdx;
// Two rows per wave: one row per wave (16 waves, 128 VGPRs each) spills its z queues and runs at half the speed.
FDTDX_TBM_VARIANT(2, 8, 2, 2, 0)
FDTDX_TBM_VARIANT(2, 8, 2, 1, 0)  // half-width tile: the last x tile of rows wider than 128 quads
FDTDX_TBM_STRIP_VARIANT(2, 8, 2, 0, 4)  // strip: rows with at most 14 quads left after their last full x tile
