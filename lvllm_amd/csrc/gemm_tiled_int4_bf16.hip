// gemm_tiled_int4_bf16.hip -- instantiates the LDS-staged tiled grouped-GEMM kernels (gemm_tiled.h)
// for one (weight format, activation dtype) pair.
#include "gemm_prefill.h"
namespace lkm {
LKM_DEFINE_TILED_LAUNCHERS(int4_bf16, LKM_W_INT4_B8, LKM_DT_BF16)
}  // namespace lkm
