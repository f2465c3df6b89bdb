// tb_r2_golden.hip -- the fused radius-2 star sweep in the host loop's order (fdtd2.cu:46-52: separate IEEE multiplies and
// adds, (+r, -r) pairs parenthesised), the order whose results the reference's own compiled loop pins bit for bit.
#include "tbm_kernel.h"
using namespace fdtdx;
FDTDX_TBM_VARIANT(2, 8, 2, 2, 1)
FDTDX_TBM_VARIANT(2, 8, 2, 1, 1)  // half-width tile: the last x tile of rows wider than 128 quads
FDTDX_TBM_STRIP_VARIANT(2, 8, 2, 1, 4)  // strip: rows with at most 14 quads left after their last full x tile
