// gemm_tiled_nvfp4_f16.hip -- instantiates the LDS-staged tiled grouped-GEMM kernels (gemm_tiled.h)
// for one (weight format, activation dtype) pair.
#include "gemm_prefill.h"
namespace lkm {
LKM_DEFINE_TILED_LAUNCHERS(nvfp4_f16, LKM_W_NVFP4, LKM_DT_F16)
}  // namespace lkm
