// gemm_w4x_int4zp_f16.hip -- instantiates the 32x32-MFMA 4-bit decode kernels (gemm_w4x.h, gemm_w4e.h) for uint4 weights with
// zero points (LkmConfig.int4_mode = LKM_INT4_ZP).
#include "gemm_w4e.h"
namespace lkm {
LKM_DEFINE_W4X_LAUNCHER(int4zp_f16, LKM_W_INT4_ZP, LKM_DT_F16)
}  // namespace lkm
