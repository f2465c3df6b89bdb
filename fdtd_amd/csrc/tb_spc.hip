// tb_spc.hip -- shapes of the fused (two steps per launch) 13-point sweep (fdtdspc.cu:64-73), both orders.
#include "tbs_kernel.h"
using namespace fdtdx;
// Two rows per wave over the whole row: three planes of two rows per level plus the neighbouring rows kept an
// iteration longer are 20 quads per lane and level.
FDTDX_TBS_VARIANT(8, 2, 2, 0)
FDTDX_TBS_VARIANT(8, 2, 1, 0)  // half-width tile: the last x tile of rows wider than 128 quads
FDTDX_TBS_VARIANT(8, 2, 2, 1)
FDTDX_TBS_VARIANT(8, 2, 1, 1)
// strip shape: the four quads a row of 1024 cells has left after two full tiles, as segments of 16 quads in four y tiles
FDTDX_TBS_STRIP_VARIANT(8, 2, 0, 4)
FDTDX_TBS_STRIP_VARIANT(8, 2, 1, 4)
