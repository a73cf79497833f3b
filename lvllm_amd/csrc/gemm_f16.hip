// gemm_f16.hip -- instantiates the skinny grouped-GEMM kernels (gemm_skinny.h) for one
// (weight format, activation dtype) pair so the formats compile in parallel.
#include "gemm_skinny.h"
namespace lkm {
LKM_DEFINE_GEMM_LAUNCHERS(f16, LKM_W_F16, LKM_DT_F16)
}  // namespace lkm
