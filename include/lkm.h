/*
 * lkm.h -- C ABI of liblkm.so, the MI355X-native replacement for the `lk_moe` engine that
 * LvLLM (guqiong96/Lvllm) delegates its routed-expert MoE hot path to.
 *
 * Plain C: pointers, sizes and a config struct; no torch / HIP types in the signatures.
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference tree).  The Python module `lk_moe` (repo root) binds these with ctypes and
 * re-creates the reference's class surface (MOEConfigV2, MOE_BF16, ... .cpu_decode(),
 * .cpu_prefill(), .gpu_prefill()) so `vllm/model_executor/layers/fused_moe/routed_experts.py`
 * runs unchanged -- see INTEGRATION.md.
 *
 * All functions return 0 on success and a negative LKM_E_* code on failure;
 * lkm_last_error() returns a thread-local human-readable message.  Nothing here ever
 * falls back to a CPU path: if no gfx950 device is usable the call fails.
 */
#ifndef LKM_H
#define LKM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LKM_ABI_VERSION 1

/* error codes */
#define LKM_OK 0
#define LKM_E_INVALID (-1)     /* bad argument / unsupported configuration */
#define LKM_E_HIP (-2)         /* HIP runtime error (message has the hipError string) */
#define LKM_E_NOMEM (-3)
#define LKM_E_UNSUPPORTED (-4) /* format known but not built yet (NVFP4 / MXFP4) */

/* activation (hidden-state) dtypes: suffix of the lk_moe class name
 * (MOE_BF16 vs MOE_FP16 ..., routed_experts.py:1514-1533) */
#define LKM_DT_F32 0
#define LKM_DT_BF16 1
#define LKM_DT_F16 2

/* expert weight formats = the lk_moe class family (routed_experts.py:1363-1397) */
#define LKM_W_BF16 0     /* MOE_BF16          : w13 [E,2I,H] bf16, w2 [E,H,I] bf16            */
#define LKM_W_F16 1      /* MOE_FP16          : same in fp16                                   */
#define LKM_W_FP8_E4M3 2 /* MOE_FP8[_FP16]    : e4m3fn + fp32 block scales [E,N/gN,K/gK]       */
#define LKM_W_INT4_B8 3  /* MOE_WNA16[_FP16]  : uint4b8 bytes [E,N,K/2] + act-dtype scales     */
#define LKM_W_NVFP4 4    /* MOE_NVFP4[_FP16]  : E2M1 bytes [E,N,K/2] + fp8 e4m3fn block scales       *
                          *   [E,N,K/16] (linear) + per-expert f32 multipliers [E] (global)  */
#define LKM_W_MXFP4 5    /* MOE_MXFP4[_FP16]  : E2M1 bytes [E,N,K/2] + uint8 E8M0 scales [E,N,K/32]  */

/* MOEConfigV2.activation_type (routed_experts.py:160-164) */
#define LKM_ACT_SILU 0
#define LKM_ACT_SWIGLUOAI 1
#define LKM_ACT_RELU2 2

/* fp8 compute mode */
#define LKM_FP8_W8A16 0 /* lk_moe semantics: weight-only fp8, activations stay bf16/fp16      */
#define LKM_FP8_W8A8 1  /* in-tree operator semantics: dynamic 1 x groupK activation quant     */

/* uint4b8 compute mode */
#define LKM_INT4_EXACT 0 /* weights dequantised to T((q-8)*s), the reference's rounding (fused_moe.py:237-276)      */
#define LKM_INT4_FAST 1  /* opt-in: group scale applied to the fp32 partial sum of each 128-k block instead of to  *
                          * the weights -- (q-8) exact, no per-weight rounding; 2.5x fewer VALU instructions per  *
                          * weight, inside the reference's int4 tolerance (atol 2e-2) but not bit-identical to it; *
                          * groupK must be a multiple of 128                                                        */
#define LKM_INT4_ZP 2    /* asymmetric uint4 (AWQ / GPTQ with zero points, the in-tree operator's has_zp grid:     *
                          * fused_moe.py:207-208,237-238,272-276): weights dequantised to T((q - zp) * s).  The     *
                          * zero points ride in the two global-scale pointer slots of lkm_create: uint8            *
                          * [E, rows, K / group], one byte (0..15) per weight row and scale group                   */

/*
 * Mirrors lk_moe.MOEConfigV2 field for field (routed_experts.py:1490-1511), plus the three
 * things the reference encodes in the class name / pointer set (weight format, activation
 * dtype, fp8 mode).
 */
typedef struct LkmConfig {
    int32_t abi_version;      /* = LKM_ABI_VERSION */
    int32_t num_processes;    /* TP or EP world size            (:1435-1438) */
    int32_t process_id;       /* rank in that group                          */
    int32_t gpu_id;           /* HIP device ordinal                          */
    int32_t has_gate_proj;    /* 0 => non-gated (relu2) experts              */
    int32_t expert_num;       /* LOCAL experts                               */
    int32_t top_k;
    int32_t hidden_size;
    int32_t intermediate_size; /* per TP partition                           */
    int32_t max_batch_size;   /* max_num_batched_tokens                      */
    int32_t max_num_seqs;     /* decode batch bound (x (1+spec tokens))      */
    int32_t stride;           /* CPU tiling hint of the original engine: accepted, unused */
    int32_t group_min_len;    /* idem                                        */
    int32_t group_max_len;    /* prefill chunk length (:1323-1330)           */
    int32_t groupN;           /* quant block shape (rows)                    */
    int32_t groupK;           /* quant block shape (K).  The reference hands over ONE value for both GEMMs, the maximum of
                               * w13's and w2's (routed_experts.py:1440-1453).  Rule: the scale group of a GEMM is
                               * min(groupK, that GEMM's K) -- groupK >= K means one group per weight row, the scale array
                               * of that GEMM is [E, N / groupN, ceil(K / groupK)].  Per-channel checkpoints
                               * (_process_fp8(False), :1381-1383: scales [E, N, 1]; _process_wna16("channel")) therefore
                               * arrive as groupN = 1, groupK = max(hidden, intermediate) and are accepted (round 4).
                               * fp8: groupK a multiple of 128 or covering the whole K; int4: 32, 64, 128, a multiple of
                               * 128 dividing K, or the whole K (itself a multiple of 128). */
    int32_t activation_type;  /* LKM_ACT_*                                   */
    float swiglu_alpha;
    float swiglu_limit;
    int32_t use_gpu_prefill;
    int32_t weight_format;    /* LKM_W_*                                     */
    int32_t act_dtype;        /* LKM_DT_BF16 | LKM_DT_F16                    */
    int32_t fp8_mode;         /* LKM_FP8_*                                   */
    int32_t int4_mode;        /* LKM_INT4_*                                  */
    int32_t reserved[7];
} LkmConfig;

typedef struct LkmEngine* LkmHandle;

/* -------------------------------------------------------------------------------------------
 * Engine life cycle.
 * Replaces lk_moe.MOE_*(cfg, w13_ptr, w2_ptr, w13_scale_ptr, w2_scale_ptr,
 *                        w13_global_scale_ptr, w2_global_scale_ptr)
 * (routed_experts.py:1514-1533, 1596-1616, 1648-1668).  Pointers may be host OR device
 * memory (detected); contiguous, layouts of SURVEY 8(a5).  The engine COPIES the weights
 * (pre-shuffled into its MFMA-native HBM layout) -- the caller frees its tensors right
 * after (routed_experts.py:1420-1432).  NULL = absent.  The two global-scale slots carry the NVFP4 per-expert multipliers
 * (fp32 [E]) or, with weight_format LKM_W_INT4_B8 and int4_mode LKM_INT4_ZP, the zero points (uint8 [E, rows, K / group]).
 */
int lkm_create(const LkmConfig* cfg, const void* w13, const void* w2, const void* w13_scale,
               const void* w2_scale, const void* w13_global_scale, const void* w2_global_scale,
               LkmHandle* out);
void lkm_destroy(LkmHandle h);

/*
 * Replaces lk_moe.MOE_*.cpu_decode(stream, num_tokens, top_k, hidden_ptr, topk_ids_ptr,
 *                                  topk_weights_ptr, out_f32_ptr)   (routed_experts.py:1840-1855)
 * All DEVICE pointers; asynchronous on `stream` (a hipStream_t, 0 = default stream);
 * capturable in a hipGraph (no allocation, no synchronisation).  hidden [num_tokens,H] in the
 * activation dtype; ids int32 [num_tokens,top_k], local ids, <0 = skip; weights fp32;
 * out fp32 [num_tokens,H], every row written.
 */
int lkm_decode(LkmHandle h, void* stream, int32_t num_tokens, int32_t top_k, const void* hidden,
               const int32_t* topk_ids, const float* topk_weights, float* out_f32);

/*
 * Replaces lk_moe.MOE_*.cpu_prefill(num_tokens, top_k, ids_ptr, weights_ptr, hidden_ptr,
 *                                   out_f32_ptr)                  (routed_experts.py:1858-1882)
 * All HOST pointers; blocking.  (The data crosses PCIe both ways; kept for API parity.)
 */
int lkm_prefill_host(LkmHandle h, int32_t num_tokens, int32_t top_k, const int32_t* topk_ids,
                     const float* topk_weights, const void* hidden, float* out_f32);

/*
 * Replaces lk_moe.MOE_*.gpu_prefill(hidden_ptr, out_ptr, topk_ids_ptr, topk_weights_ptr,
 *                                   num_tokens, top_k, stream)    (routed_experts.py:1884-1899)
 * DEVICE pointers; out has the activation dtype; asynchronous on `stream`.
 */
int lkm_prefill_device(LkmHandle h, const void* hidden, void* out, const int32_t* topk_ids,
                       const float* topk_weights, int32_t num_tokens, int32_t top_k, void* stream);

/*
 * The same operator on STRIDED inputs, output dtype chosen by the caller: what lkm_decode (out fp32) and
 * lkm_prefill_device (out = activation dtype) are special cases of.  Exists for the expert-parallel
 * exchange (lkm_ep_pack_tokens below): rows arrive as [token row | ids | weights] records and are consumed
 * in place, with no unpack copy.  Strides are in ELEMENTS of the respective array (hidden_ld >= H and a
 * multiple of 8; ids_ld, weights_ld >= top_k); `id_offset` is subtracted from every id >= 0 before use
 * (ids that fall outside [0, expert_num) are skipped like -1): a receiver of GLOBAL ids passes its first
 * expert here (the linear form of RoutedExperts.global_to_local_expert_ids, routed_experts.py:1332-1342).
 * out [num_tokens,H] contiguous in out_dtype (LKM_DT_F32 or the activation dtype), every row written.
 * DEVICE pointers, asynchronous on `stream`, graph-capturable.
 */
int lkm_forward_strided(LkmHandle h, void* stream, int32_t num_tokens, int32_t top_k, const void* hidden,
                        int64_t hidden_ld, const int32_t* topk_ids, int64_t ids_ld, int32_t id_offset,
                        const float* topk_weights, int64_t weights_ld, void* out, int32_t out_dtype);

/* -------------------------------------------------------------------------------------------
 * Routing (runs just before the engine call, moe_runner.py:577-600).
 *
 * Replaces torch.ops._moe_C.topk_softmax / topk_sigmoid
 * (vllm/_custom_ops.py:2228-2290 -> csrc/libtorch_stable/moe/topk_softmax_kernels.cu:822-860).
 * logits [M,E] in `logits_dtype` (LKM_DT_*); bias fp32 [E] or NULL; scoring 0=softmax 1=sigmoid.
 * Outputs: weights fp32 [M,K], ids int32 [M,K].  DEVICE pointers, async on stream.
 */
int lkm_topk_softmax(void* stream, const void* logits, int32_t logits_dtype, const float* bias,
                     int32_t M, int32_t E, int32_t K, int32_t scoring, int32_t renormalize,
                     float routed_scaling, float* out_weights, int32_t* out_ids);

/*
 * Replaces grouped_topk / torch.ops._moe_C.grouped_topk
 * (vllm/model_executor/layers/fused_moe/router/grouped_topk_router.py:80-161).
 */
int lkm_grouped_topk(void* stream, const void* logits, int32_t logits_dtype, const float* bias,
                     int32_t M, int32_t E, int32_t K, int32_t n_group, int32_t topk_group,
                     int32_t scoring, int32_t renormalize, float routed_scaling,
                     float* out_weights, int32_t* out_ids);

/*
 * Routing + experts in one call (the decode step of FusedMoE.forward_impl: router.select_experts followed by
 * quant_method.apply, moe_runner.py:577-614): the routing of lkm_topk_softmax (n_group == 0) or lkm_grouped_topk
 * (n_group > 0) on `router_logits` [M, router_experts], then lkm_forward_strided on its result.  Same outputs, bit
 * for bit, as the two calls made one after the other; for decode batches (M * top_k <= 1024, router_experts <= 256,
 * at most two passes of one workgroup over the rows) the router and the token->expert scatter metadata come out of ONE
 * launch, and for one to four tokens (top_k <= 16, experts small enough that a repeated expert costs less than two
 * launches: always at M = 1) the first wavefront of every GEMM1 workgroup routes its token's row itself and GEMM2 forms
 * the weighted sum per token: the whole step is two launches.  topk_weights_out [M,K] fp32 and
 * topk_ids_out [M,K] int32 (the router's global ids) are always written: callers need them for EPLB load recording
 * and shared-expert handling.  id_offset as in lkm_forward_strided.  DEVICE pointers, asynchronous, capturable.
 */
int lkm_forward_routed(LkmHandle h, void* stream, int32_t num_tokens, int32_t top_k, const void* hidden,
                       int64_t hidden_ld, const void* router_logits, int32_t logits_dtype, int32_t router_experts,
                       const float* score_bias, int32_t n_group, int32_t topk_group, int32_t scoring,
                       int32_t renormalize, float routed_scaling, int32_t id_offset, float* topk_weights_out,
                       int32_t* topk_ids_out, void* out, int32_t out_dtype);

/*
 * Router GEMM + routing in one call: logits = hidden . gate_w^T (+ gate_bias), then the
 * routing of lkm_topk_softmax (n_group == 0) or lkm_grouped_topk (n_group > 0) on those logits.
 * Replaces the gate projection + router of moe_runner.py:903-908 / router/gate_linear.py:17-34
 * (F.linear or the small-M router GEMMs; numerical reference tests/kernels/test_fp32_router_gemm.py:
 * F.linear in fp32) followed by fused_topk / grouped_topk.
 *   hidden [M,H] bf16|fp16 (x_dtype); gate_w [E,H] in x_dtype or fp32 (w_dtype); gate_bias [E] fp32 or
 *   NULL; score_bias = e_score_correction_bias [E] or NULL; logits_dtype: LKM_DT_F32 keeps the fp32
 *   logits (router GEMMs with fp32 output), x_dtype rounds them first (F.linear in the gate's dtype);
 *   workspace: >= lkm_router_workspace_bytes(M,H,E) bytes of device memory (split-K partials);
 *   logits_out [M,E] fp32 or NULL.  H must be a multiple of 32 (16 with fp32 gate weights).
 * Two launches, stream-ordered, graph-capturable.
 */
int64_t lkm_router_workspace_bytes(int32_t M, int32_t H, int32_t E);
int lkm_router_gemm_topk(void* stream, const void* hidden, int32_t x_dtype, const void* gate_w,
                         int32_t w_dtype, const float* gate_bias, const float* score_bias, int32_t M,
                         int32_t H, int32_t E, int32_t K, int32_t scoring, int32_t renormalize,
                         float routed_scaling, int32_t n_group, int32_t topk_group,
                         int32_t logits_dtype, void* workspace, int64_t workspace_bytes,
                         float* logits_out, float* out_weights, int32_t* out_ids);

/*
 * Replaces RoutedExperts.global_to_local_expert_ids (routed_experts.py:1332-1342):
 * out[i] = ids[i] < 0 ? -1 : expert_map[clamp(ids[i], 0, E-1)].  DEVICE pointers.
 */
int lkm_map_expert_ids(void* stream, const int32_t* ids, int64_t n, const int32_t* expert_map,
                       int32_t E, int32_t* out);

/*
 * Expert-parallel exchange (the MI355X replacement of the CPU-NUMA tier: experts sharded over the GPUs of a
 * node; stands where the reference's all-to-all prepare/finalize backends stand,
 * vllm/model_executor/layers/fused_moe/all2all_utils.py + modular_kernel.py:257-418).  TOKEN-granular and
 * fixed-capacity: a token travels to a rank at most ONCE, whatever number of its top-k experts live there, as
 * one record  [ H x 16-bit activations | top_k x int32 ids | top_k x fp32 weights ]  padded to 16 bytes
 * (lkm_ep_row_bytes); every destination gets `capacity` record slots, so the all-to-all that follows has
 * equal splits: no split-size exchange, no host synchronisation, capturable in a hipGraph.  capacity =
 * num_tokens is the exact worst case (no overflow possible); with a smaller capacity the tokens that do not
 * fit are NOT sent and are counted in *overflow (the caller re-runs the step through its ragged path).
 * Placement = the reference's linear map (expert_map_manager.py:62-79).
 *   send    [ep][capacity][row_bytes]; ids of a record are LOCAL ids at the destination (global_ids = 0) or
 *           GLOBAL ids (global_ids = 1, for receivers that apply expert_map themselves), -1 for the slots of
 *           the token that live on other ranks; unused record slots carry ids = -1 (their activations are
 *           not written and never read);
 *   slot_of int32 [ep][num_tokens]: record index of token m in the block sent to rank p, -1 = not sent;
 *   overflow int32 [1], incremented by the number of (token, rank) pairs dropped for lack of capacity.
 */
int64_t lkm_ep_row_bytes(int32_t H, int32_t K);
int lkm_ep_pack_tokens(void* stream, const void* hidden, const int32_t* topk_ids, const float* topk_weights,
                       int32_t M, int32_t K, int32_t H, int32_t num_experts, int32_t ep_size, int32_t capacity,
                       int32_t global_ids, void* send, int32_t* slot_of, int32_t* overflow);
/*
 * The return leg: back [ep][capacity][H] (back_dtype: fp32 or a 16-bit activation dtype) holds, from every
 * rank p, the weighted sum over p's experts of the records this rank sent there;
 * out[m] = sum_p back[p][slot_of[p][m]] in fp32, p ascending (fixed order), written in out_dtype.
 * Every row of out [num_tokens,H] is written (zeros for a token no rank computed).
 */
int lkm_ep_combine(void* stream, const void* back, int32_t back_dtype, const int32_t* slot_of, int32_t M,
                   int32_t H, int32_t ep_size, int32_t capacity, void* out, int32_t out_dtype);

/*
 * Token->expert scatter metadata, exposed for tests and for the expert-parallel host code.
 * Stable counting sort of the n_slots = M*K assignments by expert
 * (csrc/cpu/cpu_fused_moe.cpp:200-227; moe_permute's stable sort, moe_permute_unpermute_kernel.cu:45-60).
 * counts [E], offsets [E+1], sorted_slot [n_slots] (tail = -1), pos_of_slot [n_slots] (-1 = skipped).
 */
int lkm_sort_slots(void* stream, const int32_t* ids, int32_t n_slots, int32_t E, int32_t* counts,
                   int32_t* offsets, int32_t* sorted_slot, int32_t* pos_of_slot);

/*
 * Dynamic per-token-group fp8 quantisation of activation rows, group = 128 columns: the operator the in-tree block-fp8
 * path applies to both GEMM inputs (per_token_group_quant_fp8, model_executor/layers/quantization/utils/fp8_utils.py:
 * 533-660, csrc/libtorch_stable/quantization/w8a8/fp8/per_token_group_quant.cu:100; spec tests/kernels/quant_utils.py:
 * 157-180): scale = max(amax, 1e-10) / 448, q = clamp(x / scale, +-448) -> e4m3fn.  x [rows][ld_x] (16-bit, dtype
 * x_dtype, cols a multiple of 8), q [rows][cols] bytes, scales fp32 [rows][ceil(cols / 128)] (row-major).  The engine
 * runs the same kernel on its W8A8 inputs; exposed so that the bytes can be checked against the reference arithmetic.
 */
int lkm_per_token_group_quant_fp8(void* stream, const void* x, int32_t x_dtype, int64_t ld_x, int32_t rows, int32_t cols,
                                  void* q, float* scales);

/*
 * Weight-only integer experts of the in-tree operator surface (secondary boundary): int4_w4a16 / int8_w8a16 with or
 * without zero points (vllm/model_executor/layers/fused_moe/fused_moe.py:207-276; config.py int4_w4a16_moe_quant_config /
 * int8_w8a16_moe_quant_config; grid tests/kernels/moe/test_moe.py:565-693) expanded ONCE, at weight hand-off, to the
 * 16-bit weights that kernel feeds its dot product: out[r][k] = T((q[r][k] - zp[r][k / group]) * scale[r][k / group]),
 * one rounding, zp = 8 / 128 when `zeros` is null.  rows = E*N; qweight [rows][K/2] (low nibble = even k) or
 * [rows][K]; scales [rows][K/group] in out_dtype; zeros 4-bit [rows/2][K/group] (low nibble = even row) or 8-bit
 * [rows][K/group]; out [rows][K] 16-bit.  The engine then runs its 16-bit kernels on the image (288 GB of HBM pay
 * for exact zero-point arithmetic with no extra decoder; the symmetric 4-bit case keeps its native packed format).
 * The primary boundary does not take this path: lk_moe refuses AWQ / zero points (routed_experts.py:1539-1547).
 */
int lkm_wna16_expand(void* stream, const void* qweight, const void* scales, const void* zeros, void* out, int64_t rows,
                     int32_t K, int32_t group, int32_t weight_bits, int32_t out_dtype);

/* -------------------------------------------------------------------------------------------
 * The scatter / gather step as stand-alone operators.  The engine never materialises these forms (its GEMMs
 * gather through the sort's index lists); they exist for callers that want the reference operators' outputs.  All device
 * pointers; `workspace`: lkm_moe_ops_workspace_bytes(n_slots, n_keys) bytes of device scratch, 4-byte aligned -- no
 * allocation, no host synchronisation, graph-capturable.  Stable: the rows of an expert keep token order.
 *
 * lkm_moe_align_block_size -- moe_align_block_size (vllm/model_executor/layers/fused_moe/moe_align_block_size.py:11-103;
 * golden tests/kernels/moe/test_moe_align_block_size.py:96-172).  topk_ids [n_slots] int32; every expert's rows padded to a
 * multiple of block_size.  sorted_ids [sorted_cap]: slot indices expert by expert, padding = n_slots (the tail too);
 * expert_ids [blocks_cap]: the expert of each block, -1 past the last one; num_tokens_post_pad [1].  expert_map (NULL or
 * [num_experts]) = the reference's ignore_invalid_experts=True: ids mapped to -1 (and ids < 0) take no part, expert_ids
 * hold the mapped (local) ids.  n_keys for the workspace: num_experts.  num_experts <= 512.
 *
 * lkm_moe_permute -- moe_permute (moe_permute_unpermute.py:105-242; golden tests/kernels/moe/test_moe_permute_unpermute.py:
 * 37-89).  hidden [n_token][row_bytes] (row_bytes % 16 == 0), topk_ids [n_token][topk].  Rows sorted by local expert id;
 * slots of experts that are not local (expert_map[id] == -1) sort behind them by global id.  expert_first_token_offset
 * int64 [n_local_expert + 1]; inv_permuted_idx int32 [n_token * topk]: slot -> permuted row; permuted_idx int32
 * [n_token * topk]: permuted row -> slot, n_token * topk for the rows of non-local experts; permuted_hidden
 * [n_token * topk][row_bytes]: the rows of the LOCAL experts (the others are not written).  n_keys for the workspace:
 * n_expert without an expert_map, min(n_local_expert + n_expert, 512) with one.  n_expert <= 512 (the engine's limit,
 * lkm_create); a 512-expert model under expert parallelism is accepted (non-local experts are ranked into n_expert keys).
 *
 * lkm_moe_unpermute -- moe_unpermute (moe_permute_unpermute.py:245-283): out[t] = T(sum_k w[t][k] * rows[inv[t][k]]) over
 * the rows below expert_first_token_offset[n_local_expert] (NULL: all), fp32 sum in slot order, one rounding.  rows / out
 * in `dtype` (LKM_DT_BF16 / LKM_DT_F16), n_hidden % 8 == 0.
 */
int64_t lkm_moe_ops_workspace_bytes(int32_t n_slots, int32_t n_keys);
int lkm_moe_align_block_size(void* stream, const int32_t* topk_ids, int32_t n_slots, int32_t num_experts,
                             int32_t block_size, const int32_t* expert_map, int32_t* sorted_ids, int32_t sorted_cap,
                             int32_t* expert_ids, int32_t blocks_cap, int32_t* num_tokens_post_pad, void* workspace);
int lkm_moe_permute(void* stream, const void* hidden, int32_t row_bytes, int32_t n_token, const int32_t* topk_ids,
                    int32_t topk, const int32_t* expert_map, int32_t n_expert, int32_t n_local_expert,
                    void* permuted_hidden, int64_t* expert_first_token_offset, int32_t* inv_permuted_idx,
                    int32_t* permuted_idx, void* workspace);
int lkm_moe_unpermute(void* stream, const void* permuted_hidden, int32_t dtype, const float* topk_weights,
                      const int32_t* inv_permuted_idx, const int64_t* expert_first_token_offset, int32_t n_local_expert,
                      int32_t n_token, int32_t topk, int32_t n_hidden, void* out);

/* -------------------------------------------------------------------------------------------
 * Introspection / measurement.
 */
/* 1 when `p` is device (or managed) memory of a HIP device, 0 for host memory -- the rule lkm_create applies to its six
 * weight pointers.  The host-side `lk_moe` classes use it to decide whether an engine that does not fit HBM can fall back to the
 * spill tier (host-resident weights streamed through a window of device slots: routed_experts.py:1344-1357, 1884-1899). */
int lkm_pointer_is_device(const void* p);

const char* lkm_last_error(void);
int lkm_abi_version(void);
/* number of visible HIP devices and the gfx arch name of device 0 (buf >= 32 bytes) */
int lkm_device_info(int32_t* n_devices, char* arch_buf, int32_t buf_len);

/* When enabled the engine brackets every kernel of the next decode/prefill_device call with
 * hipEvents on the call's stream (NOT graph-capturable while on). */
int lkm_set_profiling(LkmHandle h, int32_t enable);
#define LKM_PROF_SORT 0
#define LKM_PROF_GEMM1 1
#define LKM_PROF_GEMM2 2
#define LKM_PROF_COMBINE 3
#define LKM_PROF_N 4
/* Synchronises the profiled stream and returns per-kernel milliseconds of the last call. */
int lkm_get_profile(LkmHandle h, float* ms /* [LKM_PROF_N] */);
/* HBM bytes held by this engine (weights + scales), and its launch geometry as text. */
int64_t lkm_weight_bytes(LkmHandle h);
int lkm_describe(LkmHandle h, char* buf, int32_t buf_len);
/* The GEMM kernels the engine's last step launched, as the profilers print them (demangled device-function names,
 * template arguments included): "gemm1=<name>[+<name>];gemm2=<name>[+<name>]" -- two names where a hybrid plan ran the
 * streamer and the tile kernel.  bench.py's roofline.kernel and the FETCH passes of tools/update_hbm_traffic.py take
 * the name from here (what benchmarks/kernels/benchmark_moe.py:304-333 gets from its config dict). */
int lkm_last_kernels(LkmHandle h, char* buf, int32_t buf_len);
/* First-call autotune ("autotune" = 1, or 2 on an expert-parallel engine) and expert parallelism: the plans remembered so
 * far as (shape key, index into the shape's candidate list) in ascending key order -- returns their number (fills at most
 * `cap`) -- and the call that replaces one by the index a group agreed on.  The candidate list is a function of the
 * engine's configuration and the step shape only, so equal local shapes give equal lists on every rank. */
int lkm_tuned_plans(LkmHandle h, int64_t* keys, int32_t* index, int32_t cap);
int lkm_tuned_plan_set(LkmHandle h, int64_t key, int32_t index);
/* tuning knobs (bench / tests): key in {"nt1","nt2","kw1","sk2","tbmax","tiled","waves","hybrid","pd1",
 * "pd2","xcd","pf","direct","valid_den","prof_rep"}; value 0 = auto ("tiled": -1 forces the skinny
 * streamer, 64 / 128 / 256 force a token-tile size; "hybrid": -1 disables the skinny+tiled split by
 * rows-per-expert; "prof_rep": N > 1 launches each GEMM N times between its two profiling events and
 * lkm_get_profile divides the interval -- profiling mode only, results unchanged; "mixed": mixed tile heights of
 * decode-sized steps, opt-in -- n > 0 = experts with more than n rows take 128-row tiles (measured: no gain in the step); "autotune": 0 off,
 * 1 first-call plan search, 2 the same on an expert-parallel engine whose host agrees on the plans across the group) */
int lkm_set_tuning(LkmHandle h, const char* key, int32_t value);

/* Measures the HBM *read* ceiling of the device with the access shape of the expert-weight stream
 * (bench / DESIGN.md reference point): streams `bytes` of device memory `reps` times through
 * n_blocks x 256 threads with `unroll` 1-KiB nontemporal loads in flight per wave. */
int lkm_hbm_read_probe(void* stream, const void* device_buf, int64_t bytes, int32_t n_blocks,
                       int32_t unroll, int32_t reps, float* ms_per_rep);

#ifdef __cplusplus
}
#endif
#endif /* LKM_H */
