// gemm_int4zp_bf16.hip -- skinny grouped-GEMM kernels for uint4 weights with zero points (LkmConfig.int4_mode = LKM_INT4_ZP).
#include "gemm_skinny.h"
namespace lkm {
LKM_DEFINE_GEMM_LAUNCHERS(int4zp_bf16, LKM_W_INT4_ZP, LKM_DT_BF16)
}  // namespace lkm
