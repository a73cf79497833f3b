// gemm_tiled_fp8a8_f16.hip -- LDS-staged tiled grouped GEMMs, fp8 weights x fp8 activations (W8A8).
#include "gemm_prefill_a8w.h"
namespace lkm {
LKM_DEFINE_TILED_LAUNCHERS(fp8a8_f16, LKM_W_FP8_A8, LKM_DT_F16)
}  // namespace lkm
