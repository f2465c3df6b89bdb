// spc13.hip -- tile shapes of the 13-point sweep (fdtdspc.cu).
#include "spc_kernel.h"
using namespace fdtdx;
FDTDX_SPC_VARIANT(128, 8, 1, 0)
FDTDX_SPC_VARIANT(128, 4, 2, 0)
FDTDX_SPC_VARIANT(64, 16, 1, 0)
FDTDX_SPC_VARIANT(64, 8, 2, 0)
FDTDX_SPC_VARIANT(64, 4, 2, 0)
FDTDX_SPC_VARIANT(32, 32, 1, 0)
FDTDX_SPC_VARIANT(32, 16, 2, 0)
FDTDX_SPC_VARIANT(32, 8, 4, 0)
FDTDX_SPC_VARIANT(128, 8, 1, 1)
FDTDX_SPC_VARIANT(32, 8, 4, 1)
