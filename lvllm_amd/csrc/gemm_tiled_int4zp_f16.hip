// gemm_tiled_int4zp_f16.hip -- LDS-staged tiled grouped GEMMs, uint4 weights with zero points (LkmConfig.int4_mode = LKM_INT4_ZP).
#include "gemm_prefill.h"
namespace lkm {
LKM_DEFINE_TILED_LAUNCHERS(int4zp_f16, LKM_W_INT4_ZP, LKM_DT_F16)
}  // namespace lkm
