// gemm_mxfp4_f16.hip -- instantiates the skinny streamer kernels (gemm_skinny.h) for one
// (weight format, activation dtype) pair.
#include "gemm_skinny.h"
namespace lkm {
LKM_DEFINE_GEMM_LAUNCHERS(mxfp4_f16, LKM_W_MXFP4, LKM_DT_F16)
}  // namespace lkm
