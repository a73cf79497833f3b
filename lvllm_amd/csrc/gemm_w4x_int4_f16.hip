// gemm_w4x_int4_f16.hip -- instantiates the 32x32-MFMA 4-bit decode kernels (gemm_w4x.h, gemm_w4e.h, gemm_w4s.h) for one
// (weight format, activation dtype) pair.
#include "gemm_w4s.h"
namespace lkm {
LKM_DEFINE_W4S_LAUNCHER(int4_f16, LKM_W_INT4_B8, LKM_DT_F16)
}  // namespace lkm
