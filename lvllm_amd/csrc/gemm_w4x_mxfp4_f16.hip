// gemm_w4x_mxfp4_f16.hip -- instantiates the 32x32-MFMA 4-bit decode kernels (gemm_w4x.h, gemm_w4e.h) for one (weight format,
// activation dtype) pair.
#include "gemm_w4e.h"
namespace lkm {
LKM_DEFINE_W4X_LAUNCHER(mxfp4_f16, LKM_W_MXFP4, LKM_DT_F16)
}  // namespace lkm
