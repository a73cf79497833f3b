// gemm_w4x_nvfp4_bf16.hip -- instantiates the 32x32-MFMA 4-bit decode kernels (gemm_w4x.h, gemm_w4e.h) for one (weight format,
// activation dtype) pair.
#include "gemm_w4e.h"
namespace lkm {
LKM_DEFINE_W4X_LAUNCHER(nvfp4_bf16, LKM_W_NVFP4, LKM_DT_BF16)
}  // namespace lkm
