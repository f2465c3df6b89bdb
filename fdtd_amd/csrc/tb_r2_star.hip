// tb_r2_star.hip -- shapes of the fused (two steps per launch) radius-2 star sweep (fdtd2.cu:66-69).
#include "tbm_kernel.h"
using namespace fdtdx;
// the first shape listed is the default (tools/tune.py --tb-variant)
FDTDX_TBM_VARIANT(2, 8, 2, 2)
FDTDX_TBM_VARIANT(2, 16, 1, 2)
