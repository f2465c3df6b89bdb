// tb_r1_golden_jacobi.hip -- the fused 7-point jacobi sweep in the host loop's order (jacobi.cu:478-485: separate IEEE
// multiplies and adds, summed left to right), the order whose results the reference's own compiled loop pins bit for bit.
#include "tb_kernel.h"
using namespace fdtdx;
FDTDX_TB_VARIANT(3, 16, 1, 2, ORD_JACOBI_GOLDEN)
FDTDX_TB_VARIANT(2, 8, 3, 2, ORD_JACOBI_GOLDEN)
FDTDX_TB_VARIANT(3, 16, 1, 1, ORD_JACOBI_GOLDEN)  // half-width tile: the last x tile of rows wider than 128 quads
FDTDX_TB_STRIP_VARIANT(3, 16, 1, ORD_JACOBI_GOLDEN, 4)  // strip: rows with at most 14 quads left after their last full x tile
