// gemm_dispatch.hip -- routes (weight format, activation dtype) to the per-format launchers.
#include "lkm_kernels.h"
namespace lkm {
#define LKM_DECL(SUFFIX)                                                                            \
    int launch_gemm1_##SUFFIX(hipStream_t, const LaunchCfg&, const GemmParams&, bool, int);         \
    int launch_gemm2_##SUFFIX(hipStream_t, const LaunchCfg&, const GemmParams&, int);               \
    int launch_gemm2_direct_##SUFFIX(hipStream_t, const LaunchCfg&, const GemmParams&, int);        \
    int launch_gemm1_tiled_##SUFFIX(hipStream_t, const LaunchCfg&, const GemmParams&, bool, int);   \
    int launch_gemm2_tiled_##SUFFIX(hipStream_t, const LaunchCfg&, const GemmParams&, int);
LKM_DECL(bf16) LKM_DECL(f16) LKM_DECL(int4_bf16) LKM_DECL(int4_f16) LKM_DECL(fp8_bf16) LKM_DECL(fp8_f16)
LKM_DECL(mxfp4_bf16) LKM_DECL(mxfp4_f16) LKM_DECL(nvfp4_bf16) LKM_DECL(nvfp4_f16)
LKM_DECL(int4ps_bf16) LKM_DECL(int4ps_f16) LKM_DECL(int4zp_bf16) LKM_DECL(int4zp_f16)
#undef LKM_DECL
int launch_gemm1_fp8a8_bf16(hipStream_t, const LaunchCfg&, const GemmParams&, bool, int);
int launch_gemm2_fp8a8_bf16(hipStream_t, const LaunchCfg&, const GemmParams&, int);
int launch_gemm2_direct_fp8a8_bf16(hipStream_t, const LaunchCfg&, const GemmParams&, int);
int launch_gemm2_direct_fp8a8_f16(hipStream_t, const LaunchCfg&, const GemmParams&, int);
int launch_gemm1_fp8a8_f16(hipStream_t, const LaunchCfg&, const GemmParams&, bool, int);
int launch_gemm2_fp8a8_f16(hipStream_t, const LaunchCfg&, const GemmParams&, int);
int launch_gemm1_tiled_fp8a8_bf16(hipStream_t, const LaunchCfg&, const GemmParams&, bool, int);
int launch_gemm2_tiled_fp8a8_bf16(hipStream_t, const LaunchCfg&, const GemmParams&, int);
int launch_gemm1_tiled_fp8a8_f16(hipStream_t, const LaunchCfg&, const GemmParams&, bool, int);
int launch_gemm2_tiled_fp8a8_f16(hipStream_t, const LaunchCfg&, const GemmParams&, int);

// round 4: the 32x32-MFMA kernels of the 4-bit formats (gemm_w4x.h: cfg.pf == 5; gemm_w4e.h, loader wave: cfg.pf == 6;
// round 6: gemm_w4s.h, token loader wave + register-streamed weights: cfg.pf == 7);
// false = not taken (shape / variant)
#define LKM_DECL_W4X(SUFFIX) bool launch_w4x_##SUFFIX(hipStream_t, const LaunchCfg&, const GemmParams&, bool, bool, int, int*);
LKM_DECL_W4X(int4_bf16) LKM_DECL_W4X(int4_f16) LKM_DECL_W4X(mxfp4_bf16) LKM_DECL_W4X(mxfp4_f16) LKM_DECL_W4X(nvfp4_bf16) LKM_DECL_W4X(nvfp4_f16)
LKM_DECL_W4X(int4zp_bf16) LKM_DECL_W4X(int4zp_f16)
#undef LKM_DECL_W4X
static bool launch_w4x(hipStream_t st, int wf, int adt, const LaunchCfg& cfg, const GemmParams& p, bool gated, bool is_g1,
                       int max_tiles, int* rc) {
    if (cfg.pf != 5 && cfg.pf != 6 && cfg.pf != 7) return false;     // (7: gemm_w4s.h, uint4b8 only)
    if (wf == LKM_W_INT4_B8 && adt == LKM_DT_BF16) return launch_w4x_int4_bf16(st, cfg, p, gated, is_g1, max_tiles, rc);
    if (wf == LKM_W_INT4_B8 && adt == LKM_DT_F16) return launch_w4x_int4_f16(st, cfg, p, gated, is_g1, max_tiles, rc);
    if (wf == LKM_W_MXFP4 && adt == LKM_DT_BF16) return launch_w4x_mxfp4_bf16(st, cfg, p, gated, is_g1, max_tiles, rc);
    if (wf == LKM_W_MXFP4 && adt == LKM_DT_F16) return launch_w4x_mxfp4_f16(st, cfg, p, gated, is_g1, max_tiles, rc);
    if (wf == LKM_W_NVFP4 && adt == LKM_DT_BF16) return launch_w4x_nvfp4_bf16(st, cfg, p, gated, is_g1, max_tiles, rc);
    if (wf == LKM_W_NVFP4 && adt == LKM_DT_F16) return launch_w4x_nvfp4_f16(st, cfg, p, gated, is_g1, max_tiles, rc);
    if (cfg.pf != 7 && wf == LKM_W_INT4_ZP && adt == LKM_DT_BF16) return launch_w4x_int4zp_bf16(st, cfg, p, gated, is_g1, max_tiles, rc);
    if (cfg.pf != 7 && wf == LKM_W_INT4_ZP && adt == LKM_DT_F16) return launch_w4x_int4zp_f16(st, cfg, p, gated, is_g1, max_tiles, rc);
    return false;
}

int launch_gemm1(hipStream_t st, int wf, int adt, const LaunchCfg& cfg, const GemmParams& p,
                 bool gated, int max_active) {
    if (max_active <= 0 || p.groups <= 0) return LKM_OK;
    if (wf == LKM_W_BF16 && adt == LKM_DT_BF16) return launch_gemm1_bf16(st, cfg, p, gated, max_active);
    if (wf == LKM_W_F16 && adt == LKM_DT_F16) return launch_gemm1_f16(st, cfg, p, gated, max_active);
    if (wf == LKM_W_INT4_B8 && adt == LKM_DT_BF16) return launch_gemm1_int4_bf16(st, cfg, p, gated, max_active);
    if (wf == LKM_W_INT4_B8 && adt == LKM_DT_F16) return launch_gemm1_int4_f16(st, cfg, p, gated, max_active);
    if (wf == LKM_W_MXFP4 && adt == LKM_DT_BF16) return launch_gemm1_mxfp4_bf16(st, cfg, p, gated, max_active);
    if (wf == LKM_W_MXFP4 && adt == LKM_DT_F16) return launch_gemm1_mxfp4_f16(st, cfg, p, gated, max_active);
    if (wf == LKM_W_NVFP4 && adt == LKM_DT_BF16) return launch_gemm1_nvfp4_bf16(st, cfg, p, gated, max_active);
    if (wf == LKM_W_NVFP4 && adt == LKM_DT_F16) return launch_gemm1_nvfp4_f16(st, cfg, p, gated, max_active);
    if (wf == LKM_W_FP8_E4M3 && adt == LKM_DT_BF16) return launch_gemm1_fp8_bf16(st, cfg, p, gated, max_active);
    if (wf == LKM_W_FP8_E4M3 && adt == LKM_DT_F16) return launch_gemm1_fp8_f16(st, cfg, p, gated, max_active);
    if (wf == LKM_W_INT4_PS && adt == LKM_DT_BF16) return launch_gemm1_int4ps_bf16(st, cfg, p, gated, max_active);
    if (wf == LKM_W_INT4_PS && adt == LKM_DT_F16) return launch_gemm1_int4ps_f16(st, cfg, p, gated, max_active);
    if (wf == LKM_W_INT4_ZP && adt == LKM_DT_BF16) return launch_gemm1_int4zp_bf16(st, cfg, p, gated, max_active);
    if (wf == LKM_W_INT4_ZP && adt == LKM_DT_F16) return launch_gemm1_int4zp_f16(st, cfg, p, gated, max_active);
    if (wf == LKM_W_FP8_A8 && adt == LKM_DT_BF16) return launch_gemm1_fp8a8_bf16(st, cfg, p, gated, max_active);
    if (wf == LKM_W_FP8_A8 && adt == LKM_DT_F16) return launch_gemm1_fp8a8_f16(st, cfg, p, gated, max_active);
    set_error("gemm1: unsupported weight format %d with activation dtype %d", wf, adt);
    return LKM_E_UNSUPPORTED;
}

int launch_gemm2(hipStream_t st, int wf, int adt, const LaunchCfg& cfg, const GemmParams& p,
                 int max_active) {
    if (max_active <= 0 || p.groups <= 0) return LKM_OK;
    if (wf == LKM_W_BF16 && adt == LKM_DT_BF16) return launch_gemm2_bf16(st, cfg, p, max_active);
    if (wf == LKM_W_F16 && adt == LKM_DT_F16) return launch_gemm2_f16(st, cfg, p, max_active);
    if (wf == LKM_W_INT4_B8 && adt == LKM_DT_BF16) return launch_gemm2_int4_bf16(st, cfg, p, max_active);
    if (wf == LKM_W_INT4_B8 && adt == LKM_DT_F16) return launch_gemm2_int4_f16(st, cfg, p, max_active);
    if (wf == LKM_W_MXFP4 && adt == LKM_DT_BF16) return launch_gemm2_mxfp4_bf16(st, cfg, p, max_active);
    if (wf == LKM_W_MXFP4 && adt == LKM_DT_F16) return launch_gemm2_mxfp4_f16(st, cfg, p, max_active);
    if (wf == LKM_W_NVFP4 && adt == LKM_DT_BF16) return launch_gemm2_nvfp4_bf16(st, cfg, p, max_active);
    if (wf == LKM_W_NVFP4 && adt == LKM_DT_F16) return launch_gemm2_nvfp4_f16(st, cfg, p, max_active);
    if (wf == LKM_W_FP8_E4M3 && adt == LKM_DT_BF16) return launch_gemm2_fp8_bf16(st, cfg, p, max_active);
    if (wf == LKM_W_FP8_E4M3 && adt == LKM_DT_F16) return launch_gemm2_fp8_f16(st, cfg, p, max_active);
    if (wf == LKM_W_INT4_PS && adt == LKM_DT_BF16) return launch_gemm2_int4ps_bf16(st, cfg, p, max_active);
    if (wf == LKM_W_INT4_PS && adt == LKM_DT_F16) return launch_gemm2_int4ps_f16(st, cfg, p, max_active);
    if (wf == LKM_W_INT4_ZP && adt == LKM_DT_BF16) return launch_gemm2_int4zp_bf16(st, cfg, p, max_active);
    if (wf == LKM_W_INT4_ZP && adt == LKM_DT_F16) return launch_gemm2_int4zp_f16(st, cfg, p, max_active);
    if (wf == LKM_W_FP8_A8 && adt == LKM_DT_BF16) return launch_gemm2_fp8a8_bf16(st, cfg, p, max_active);
    if (wf == LKM_W_FP8_A8 && adt == LKM_DT_F16) return launch_gemm2_fp8a8_f16(st, cfg, p, max_active);
    set_error("gemm2: unsupported weight format %d with activation dtype %d", wf, adt);
    return LKM_E_UNSUPPORTED;
}

int launch_gemm2_direct(hipStream_t st, int wf, int adt, const LaunchCfg& cfg, const GemmParams& p, int K) {
    if (K <= 0) return LKM_OK;
    if (wf == LKM_W_BF16 && adt == LKM_DT_BF16) return launch_gemm2_direct_bf16(st, cfg, p, K);
    if (wf == LKM_W_F16 && adt == LKM_DT_F16) return launch_gemm2_direct_f16(st, cfg, p, K);
    if (wf == LKM_W_INT4_B8 && adt == LKM_DT_BF16) return launch_gemm2_direct_int4_bf16(st, cfg, p, K);
    if (wf == LKM_W_INT4_B8 && adt == LKM_DT_F16) return launch_gemm2_direct_int4_f16(st, cfg, p, K);
    if (wf == LKM_W_MXFP4 && adt == LKM_DT_BF16) return launch_gemm2_direct_mxfp4_bf16(st, cfg, p, K);
    if (wf == LKM_W_MXFP4 && adt == LKM_DT_F16) return launch_gemm2_direct_mxfp4_f16(st, cfg, p, K);
    if (wf == LKM_W_NVFP4 && adt == LKM_DT_BF16) return launch_gemm2_direct_nvfp4_bf16(st, cfg, p, K);
    if (wf == LKM_W_NVFP4 && adt == LKM_DT_F16) return launch_gemm2_direct_nvfp4_f16(st, cfg, p, K);
    if (wf == LKM_W_FP8_E4M3 && adt == LKM_DT_BF16) return launch_gemm2_direct_fp8_bf16(st, cfg, p, K);
    if (wf == LKM_W_FP8_E4M3 && adt == LKM_DT_F16) return launch_gemm2_direct_fp8_f16(st, cfg, p, K);
    if (wf == LKM_W_INT4_PS && adt == LKM_DT_BF16) return launch_gemm2_direct_int4ps_bf16(st, cfg, p, K);
    if (wf == LKM_W_INT4_PS && adt == LKM_DT_F16) return launch_gemm2_direct_int4ps_f16(st, cfg, p, K);
    if (wf == LKM_W_INT4_ZP && adt == LKM_DT_BF16) return launch_gemm2_direct_int4zp_bf16(st, cfg, p, K);
    if (wf == LKM_W_INT4_ZP && adt == LKM_DT_F16) return launch_gemm2_direct_int4zp_f16(st, cfg, p, K);
    if (wf == LKM_W_FP8_A8 && adt == LKM_DT_BF16) return launch_gemm2_direct_fp8a8_bf16(st, cfg, p, K);
    if (wf == LKM_W_FP8_A8 && adt == LKM_DT_F16) return launch_gemm2_direct_fp8a8_f16(st, cfg, p, K);
    set_error("gemm2 direct: unsupported weight format %d with activation dtype %d", wf, adt);
    return LKM_E_UNSUPPORTED;
}

int launch_gemm1_tiled(hipStream_t st, int wf, int adt, const LaunchCfg& cfg, const GemmParams& p,
                       bool gated, int max_tiles) {
    if (max_tiles <= 0) return LKM_OK;
    {
        int rc = LKM_OK;
        if (launch_w4x(st, wf, adt, cfg, p, gated, true, max_tiles, &rc)) return rc;
    }
    if (wf == LKM_W_BF16 && adt == LKM_DT_BF16) return launch_gemm1_tiled_bf16(st, cfg, p, gated, max_tiles);
    if (wf == LKM_W_F16 && adt == LKM_DT_F16) return launch_gemm1_tiled_f16(st, cfg, p, gated, max_tiles);
    if (wf == LKM_W_INT4_B8 && adt == LKM_DT_BF16) return launch_gemm1_tiled_int4_bf16(st, cfg, p, gated, max_tiles);
    if (wf == LKM_W_INT4_B8 && adt == LKM_DT_F16) return launch_gemm1_tiled_int4_f16(st, cfg, p, gated, max_tiles);
    if (wf == LKM_W_MXFP4 && adt == LKM_DT_BF16) return launch_gemm1_tiled_mxfp4_bf16(st, cfg, p, gated, max_tiles);
    if (wf == LKM_W_MXFP4 && adt == LKM_DT_F16) return launch_gemm1_tiled_mxfp4_f16(st, cfg, p, gated, max_tiles);
    if (wf == LKM_W_NVFP4 && adt == LKM_DT_BF16) return launch_gemm1_tiled_nvfp4_bf16(st, cfg, p, gated, max_tiles);
    if (wf == LKM_W_NVFP4 && adt == LKM_DT_F16) return launch_gemm1_tiled_nvfp4_f16(st, cfg, p, gated, max_tiles);
    if (wf == LKM_W_FP8_E4M3 && adt == LKM_DT_BF16) return launch_gemm1_tiled_fp8_bf16(st, cfg, p, gated, max_tiles);
    if (wf == LKM_W_FP8_E4M3 && adt == LKM_DT_F16) return launch_gemm1_tiled_fp8_f16(st, cfg, p, gated, max_tiles);
    if (wf == LKM_W_INT4_PS && adt == LKM_DT_BF16) return launch_gemm1_tiled_int4ps_bf16(st, cfg, p, gated, max_tiles);
    if (wf == LKM_W_INT4_PS && adt == LKM_DT_F16) return launch_gemm1_tiled_int4ps_f16(st, cfg, p, gated, max_tiles);
    if (wf == LKM_W_INT4_ZP && adt == LKM_DT_BF16) return launch_gemm1_tiled_int4zp_bf16(st, cfg, p, gated, max_tiles);
    if (wf == LKM_W_INT4_ZP && adt == LKM_DT_F16) return launch_gemm1_tiled_int4zp_f16(st, cfg, p, gated, max_tiles);
    if (wf == LKM_W_FP8_A8 && adt == LKM_DT_BF16) return launch_gemm1_tiled_fp8a8_bf16(st, cfg, p, gated, max_tiles);
    if (wf == LKM_W_FP8_A8 && adt == LKM_DT_F16) return launch_gemm1_tiled_fp8a8_f16(st, cfg, p, gated, max_tiles);
    set_error("gemm1 tiled: unsupported weight format %d with activation dtype %d", wf, adt);
    return LKM_E_UNSUPPORTED;
}

int launch_gemm2_tiled(hipStream_t st, int wf, int adt, const LaunchCfg& cfg, const GemmParams& p,
                       int max_tiles) {
    if (max_tiles <= 0) return LKM_OK;
    {
        int rc = LKM_OK;
        if (launch_w4x(st, wf, adt, cfg, p, false, false, max_tiles, &rc)) return rc;
    }
    if (wf == LKM_W_BF16 && adt == LKM_DT_BF16) return launch_gemm2_tiled_bf16(st, cfg, p, max_tiles);
    if (wf == LKM_W_F16 && adt == LKM_DT_F16) return launch_gemm2_tiled_f16(st, cfg, p, max_tiles);
    if (wf == LKM_W_INT4_B8 && adt == LKM_DT_BF16) return launch_gemm2_tiled_int4_bf16(st, cfg, p, max_tiles);
    if (wf == LKM_W_INT4_B8 && adt == LKM_DT_F16) return launch_gemm2_tiled_int4_f16(st, cfg, p, max_tiles);
    if (wf == LKM_W_MXFP4 && adt == LKM_DT_BF16) return launch_gemm2_tiled_mxfp4_bf16(st, cfg, p, max_tiles);
    if (wf == LKM_W_MXFP4 && adt == LKM_DT_F16) return launch_gemm2_tiled_mxfp4_f16(st, cfg, p, max_tiles);
    if (wf == LKM_W_NVFP4 && adt == LKM_DT_BF16) return launch_gemm2_tiled_nvfp4_bf16(st, cfg, p, max_tiles);
    if (wf == LKM_W_NVFP4 && adt == LKM_DT_F16) return launch_gemm2_tiled_nvfp4_f16(st, cfg, p, max_tiles);
    if (wf == LKM_W_FP8_E4M3 && adt == LKM_DT_BF16) return launch_gemm2_tiled_fp8_bf16(st, cfg, p, max_tiles);
    if (wf == LKM_W_FP8_E4M3 && adt == LKM_DT_F16) return launch_gemm2_tiled_fp8_f16(st, cfg, p, max_tiles);
    if (wf == LKM_W_INT4_PS && adt == LKM_DT_BF16) return launch_gemm2_tiled_int4ps_bf16(st, cfg, p, max_tiles);
    if (wf == LKM_W_INT4_PS && adt == LKM_DT_F16) return launch_gemm2_tiled_int4ps_f16(st, cfg, p, max_tiles);
    if (wf == LKM_W_INT4_ZP && adt == LKM_DT_BF16) return launch_gemm2_tiled_int4zp_bf16(st, cfg, p, max_tiles);
    if (wf == LKM_W_INT4_ZP && adt == LKM_DT_F16) return launch_gemm2_tiled_int4zp_f16(st, cfg, p, max_tiles);
    if (wf == LKM_W_FP8_A8 && adt == LKM_DT_BF16) return launch_gemm2_tiled_fp8a8_bf16(st, cfg, p, max_tiles);
    if (wf == LKM_W_FP8_A8 && adt == LKM_DT_F16) return launch_gemm2_tiled_fp8a8_f16(st, cfg, p, max_tiles);
    set_error("gemm2 tiled: unsupported weight format %d with activation dtype %d", wf, adt);
    return LKM_E_UNSUPPORTED;
}
}  // namespace lkm
