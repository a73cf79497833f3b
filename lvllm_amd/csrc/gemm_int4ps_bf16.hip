// gemm_int4ps_bf16.hip -- skinny grouped-GEMM kernels for uint4b8 weights in the fast mode (scale on partial sums).
#include "gemm_skinny.h"
namespace lkm {
LKM_DEFINE_GEMM_LAUNCHERS(int4ps_bf16, LKM_W_INT4_PS, LKM_DT_BF16)
}  // namespace lkm
