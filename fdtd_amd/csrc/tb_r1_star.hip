// tb_r1_star.hip -- shapes of the fused (two or three steps per launch) radius-1 sweep in fdtd1.cu's order (fdtd1.cu:61-64).
#include "tb_kernel.h"
using namespace fdtdx;
// the first shape listed for a step count and order is the default (tools/tune.py --tb-variant)
FDTDX_TB_VARIANT(3, 16, 1, 2, ORD_STAR_CHAIN)
FDTDX_TB_VARIANT(3, 8, 2, 2, ORD_STAR_CHAIN)
FDTDX_TB_VARIANT(2, 8, 3, 2, ORD_STAR_CHAIN)
FDTDX_TB_VARIANT(2, 8, 2, 2, ORD_STAR_CHAIN)
FDTDX_TB_VARIANT(2, 8, 4, 2, ORD_STAR_CHAIN)
FDTDX_TB_VARIANT(2, 4, 6, 2, ORD_STAR_CHAIN)
FDTDX_TB_VARIANT(2, 4, 4, 2, ORD_STAR_CHAIN)
FDTDX_TB_VARIANT(2, 5, 4, 2, ORD_STAR_CHAIN)
FDTDX_TB_VARIANT(2, 6, 4, 2, ORD_STAR_CHAIN)
FDTDX_TB_VARIANT(2, 6, 3, 2, ORD_STAR_CHAIN)
FDTDX_TB_VARIANT(2, 16, 1, 2, ORD_STAR_CHAIN)
FDTDX_TB_VARIANT(3, 16, 1, 1, ORD_STAR_CHAIN)  // half-width tile: the last x tile of rows wider than 128 quads
FDTDX_TB_STRIP_VARIANT(3, 16, 1, ORD_STAR_CHAIN, 4)  // strip: rows with at most 14 quads left after their last full x tile
