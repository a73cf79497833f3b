// gemm_tiled_int4ps_bf16.hip -- LDS-staged tiled grouped GEMMs, uint4b8 weights in the fast mode (scale on partial sums).
#include "gemm_tiled.h"
namespace lkm {
LKM_DEFINE_TILED_LAUNCHERS(int4ps_bf16, LKM_W_INT4_PS, LKM_DT_BF16)
}  // namespace lkm
