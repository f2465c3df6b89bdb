// tb_r1_golden.hip -- the fused radius-1 sweep in the host loops' orders (fdtd1.cu:46-49, jacobi.cu:478-485: separate IEEE
// multiplies and adds), the order whose results the reference's own compiled loops pin bit for bit.
#include "tb_kernel.h"
using namespace fdtdx;
FDTDX_TB_VARIANT(3, 16, 1, 2, ORD_STAR_GOLDEN)
FDTDX_TB_VARIANT(2, 8, 3, 2, ORD_STAR_GOLDEN)
FDTDX_TB_VARIANT(3, 16, 1, 1, ORD_STAR_GOLDEN)  // half-width tile: the last x tile of rows wider than 128 quads
FDTDX_TB_VARIANT(3, 16, 1, 2, ORD_JACOBI_GOLDEN)
FDTDX_TB_VARIANT(2, 8, 3, 2, ORD_JACOBI_GOLDEN)
FDTDX_TB_VARIANT(3, 16, 1, 1, ORD_JACOBI_GOLDEN)
