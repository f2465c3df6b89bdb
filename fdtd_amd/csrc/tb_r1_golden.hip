// tb_r1_golden.hip -- the fused radius-1 star sweep in the host loop's order (fdtd1.cu:46-49: separate IEEE multiplies and adds,
// (+r, -r) pairs parenthesised), the order whose results the reference's own compiled loop pins bit for bit.
#include "tb_kernel.h"
using namespace fdtdx;
FDTDX_TB_VARIANT(3, 16, 1, 2, ORD_STAR_GOLDEN)
FDTDX_TB_VARIANT(2, 8, 3, 2, ORD_STAR_GOLDEN)
FDTDX_TB_VARIANT(3, 16, 1, 1, ORD_STAR_GOLDEN)  // half-width tile: the last x tile of rows wider than 128 quads
FDTDX_TB_STRIP_VARIANT(3, 16, 1, ORD_STAR_GOLDEN, 4)  // strip: rows with at most 14 quads left after their last full x tile
