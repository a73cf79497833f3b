// gemm_a8w_dbg.hip -- ablation builds of the round-3 fp8 x fp8 prefill kernel (gemm_prefill_a8w.h), gated GEMM1 with
// bf16 activations only: tuning key "dbg" selects one; results are wrong by construction, only the time is read.
// MEASUREMENT SCAFFOLDING: compiled only into a development library (python -m lvllm_amd.build --flag=-DLKM_ABLATIONS
// -> liblkm_<tag>.so).  The default liblkm.so -- the one `lk_moe` loads -- carries no kernel that can return wrong numbers:
// there this unit is the stub below and lkm_set_tuning refuses the "dbg" key.
#include "gemm_prefill_a8w.h"
namespace lkm {
#ifndef LKM_ABLATIONS
int launch_prefill_a8w_dbg(hipStream_t, const GemmParams&, int, int dbg, bool) {
    set_error("fp8 W8A8 prefill kernel: ablation dbg=%d needs a library built with -DLKM_ABLATIONS", dbg);
    return LKM_E_UNSUPPORTED;
}
#else
int launch_prefill_a8w_dbg(hipStream_t st, const GemmParams& p, int max_tiles, int dbg, bool gemm2) {
    if (gemm2) {
        switch (dbg) {
#define LKM_A8W_DBG_CASE(D) case D: return launch_prefill_a8w_t<LKM_DT_BF16, false, false, D>(st, p, max_tiles);
            LKM_A8W_DBG_CASE(1)
            LKM_A8W_DBG_CASE(2)
            LKM_A8W_DBG_CASE(1 | 2)
            LKM_A8W_DBG_CASE(512)          // no epilogue
            LKM_A8W_DBG_CASE(1 | 2 | 512)
#undef LKM_A8W_DBG_CASE
        default:
            set_error("fp8 W8A8 prefill kernel: GEMM2 ablation dbg=%d not built", dbg);
            return LKM_E_INVALID;
        }
    }
    switch (dbg) {
#define LKM_A8W_DBG_CASE(D) case D: return launch_prefill_a8w_t<LKM_DT_BF16, true, true, D>(st, p, max_tiles);
        LKM_A8W_DBG_CASE(1)            // compute skeleton: no loads / DMA in the K loop
        LKM_A8W_DBG_CASE(2)            // data movement + barriers only
        LKM_A8W_DBG_CASE(1 | 8)        // skeleton without LDS reads
        LKM_A8W_DBG_CASE(1 | 16)       // skeleton without VALU
        LKM_A8W_DBG_CASE(1 | 8 | 16)   // MFMAs only
        LKM_A8W_DBG_CASE(1 | 32)       // skeleton without MFMA
        LKM_A8W_DBG_CASE(1 | 64)       // skeleton without the per-unit barrier
        LKM_A8W_DBG_CASE(1 | 8 | 16 | 64)
        LKM_A8W_DBG_CASE(2 | 64)       // data movement without the barrier
        LKM_A8W_DBG_CASE(1 | 2)        // neither: prologue + epilogue + the loop's own bookkeeping and barriers
        LKM_A8W_DBG_CASE(1 | 2 | 64)
        LKM_A8W_DBG_CASE(128)          // the whole kernel, blocks do not wait for their LDS reads
        LKM_A8W_DBG_CASE(1 | 128)
        LKM_A8W_DBG_CASE(8)            // the whole kernel without: LDS reads / VALU / MFMA / barrier
        LKM_A8W_DBG_CASE(16)
        LKM_A8W_DBG_CASE(32)
        LKM_A8W_DBG_CASE(64)
        LKM_A8W_DBG_CASE(8 | 16)
        LKM_A8W_DBG_CASE(256)          // timing of the three synchronisation points (printed by two waves)
        LKM_A8W_DBG_CASE(256 | 1)
        LKM_A8W_DBG_CASE(512)          // no epilogue
        LKM_A8W_DBG_CASE(1 | 2 | 512)
#undef LKM_A8W_DBG_CASE
    default:
        set_error("fp8 W8A8 prefill kernel: ablation dbg=%d not built", dbg);
        return LKM_E_INVALID;
    }
}
#endif  // LKM_ABLATIONS
}  // namespace lkm
