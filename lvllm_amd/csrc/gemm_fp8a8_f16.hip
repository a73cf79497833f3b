// gemm_fp8a8_f16.hip -- skinny grouped-GEMM kernels for fp8 weights x fp8 activations (native fp8 MFMA).
#include "gemm_skinny.h"
namespace lkm {
LKM_DEFINE_GEMM_LAUNCHERS(fp8a8_f16, LKM_W_FP8_A8, LKM_DT_F16)
}  // namespace lkm
