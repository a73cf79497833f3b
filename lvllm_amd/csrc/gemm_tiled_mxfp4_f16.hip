// gemm_tiled_mxfp4_f16.hip -- instantiates the LDS-staged tiled grouped-GEMM kernels (gemm_tiled.h)
// for one (weight format, activation dtype) pair.
#include "gemm_prefill.h"
namespace lkm {
LKM_DEFINE_TILED_LAUNCHERS(mxfp4_f16, LKM_W_MXFP4, LKM_DT_F16)
}  // namespace lkm
