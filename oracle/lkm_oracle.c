/*
 * lkm_oracle.c -- CPU restatement of the LvLLM / lk_moe MoE expert hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * `cpu_baseline` leg and __graft_entry__.smoke() may load it.  The product path
 * (lvllm_amd / lk_moe, liblkm.so) never links, imports or calls anything here.
 *
 * PARITY STATUS: the engine that executes this path in the reference, the
 * closed PyPI binary lk_moe==2.3.3 (reference requirements/cuda.txt:37), is not
 * in /root/reference, so the lk_moe boundary itself is "parity unpinned".  What
 * IS pinned are the in-tree operators that sit at the same position
 * (moe_runner.py:602-654) and their torch oracles; this file restates those and
 * is checked against golden vectors produced by running the reference's own
 * python oracle functions (tests/golden/make_golden.py), AND against the
 * reference's own in-tree CPU fused-MoE kernel (csrc/cpu/cpu_fused_moe.cpp) run
 * here: oracle/_ref, built by `make ref` from the reference sources where they
 * lie (tests/test_oracle_ref.py).
 *
 * Each function cites the reference file:line (relative to /root/reference) it
 * follows.  Plain C99 + OpenMP; fp32 arithmetic with the rounding points of the
 * reference; integer/index work is exact.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LKM_OR_API __attribute__((visibility("default")))

/* dtype codes shared with include/lkm.h (kept numerically identical) */
enum { OR_F32 = 0, OR_BF16 = 1, OR_F16 = 2 };
/* weight formats */
enum { OR_W_BF16 = 0, OR_W_F16 = 1, OR_W_FP8_E4M3 = 2, OR_W_INT4_B8 = 3, OR_W_NVFP4 = 4, OR_W_MXFP4 = 5 };
/* activation types: routed_experts.py:160-164 */
enum { OR_ACT_SILU = 0, OR_ACT_SWIGLUOAI = 1, OR_ACT_RELU2 = 2 };

/* ------------------------------------------------------------------ scalars */

static inline float u32_as_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f32_as_u32(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static inline float bf16_to_f32(uint16_t h) { return u32_as_f32((uint32_t)h << 16); }

/* round-to-nearest-even, NaN kept quiet: same as torch .to(bfloat16) */
static inline uint16_t f32_to_bf16(float f) {
    uint32_t u = f32_as_u32(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

static inline float f16_to_f32(uint16_t h) {
    uint32_t s = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    if (e == 0) {
        if (m == 0) return u32_as_f32(s);
        /* subnormal: m * 2^-24 */
        float v = (float)m * 5.9604644775390625e-8f;
        return (s ? -v : v);
    }
    if (e == 31) return u32_as_f32(s | 0x7f800000u | (m << 13));
    return u32_as_f32(s | ((e + 112u) << 23) | (m << 13));
}

static inline uint16_t f32_to_f16(float f) {
    uint32_t u = f32_as_u32(f);
    uint32_t s = (u >> 16) & 0x8000u;
    uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)(s | 0x7e00u);
    if (a >= 0x477ff000u) {            /* >= 65520 -> inf (RNE) */
        return (uint16_t)(s | 0x7c00u);
    }
    if (a < 0x38800000u) {             /* < 2^-14: subnormal or zero */
        if (a < 0x33000000u) return (uint16_t)s;   /* < 2^-25 -> 0 */
        /* value = a_f * 2^24 rounded to integer (RNE) */
        float af = u32_as_f32(a);
        float scaled = af * 16777216.0f;           /* 2^24, exact */
        float r = nearbyintf(scaled);
        return (uint16_t)(s | (uint32_t)r);
    }
    uint32_t e = ((a >> 23) - 112u);
    uint32_t m = a & 0x7fffffu;
    uint32_t h = (e << 10) | (m >> 13);
    uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h += 1u;
    return (uint16_t)(s | h);
}

/* OCP e4m3fn (gfx950 native; reference: platforms/rocm.py gfx950 -> e4m3fn) */
static inline float fp8e4m3_to_f32(uint8_t b) {
    uint32_t s = (uint32_t)(b & 0x80u) << 24;
    uint32_t e = (b >> 3) & 0xfu, m = b & 7u;
    float v;
    if (e == 0) v = (float)m * 0.001953125f;                 /* m * 2^-9 */
    else if (e == 15 && m == 7) return u32_as_f32(s | 0x7fc00000u);
    else v = u32_as_f32(((e + 120u) << 23) | (m << 20));
    return s ? -v : v;
}

/* saturating RNE f32 -> e4m3fn, as torch .clamp(-448,448).to(float8_e4m3fn) */
static inline uint8_t f32_to_fp8e4m3(float f) {
    uint32_t u = f32_as_u32(f);
    uint8_t s = (uint8_t)((u >> 24) & 0x80u);
    float a = fabsf(f);
    if (a != a) return (uint8_t)(s | 0x7f);
    if (a >= 448.0f) return (uint8_t)(s | 0x7e);
    if (a < 0.015625f) {                        /* < 2^-6: subnormal grid 2^-9 */
        float r = nearbyintf(a * 512.0f);
        return (uint8_t)(s | (uint8_t)r);       /* r==8 -> 0x08 = 2^-6, correct */
    }
    uint32_t au = f32_as_u32(a);
    uint32_t e = (au >> 23) - 120u;
    uint32_t m = au & 0x7fffffu;
    uint32_t h = (e << 3) | (m >> 20);
    uint32_t rem = m & 0xfffffu;
    if (rem > 0x80000u || (rem == 0x80000u && (h & 1u))) h += 1u;
    if (h > 0x7eu) h = 0x7eu;
    return (uint8_t)(s | h);
}

static inline float load_act(const void* p, int dtype, size_t i) {
    switch (dtype) {
    case OR_F32: return ((const float*)p)[i];
    case OR_BF16: return bf16_to_f32(((const uint16_t*)p)[i]);
    default: return f16_to_f32(((const uint16_t*)p)[i]);
    }
}
static inline float round_act(float v, int dtype) {
    switch (dtype) {
    case OR_F32: return v;
    case OR_BF16: return bf16_to_f32(f32_to_bf16(v));
    default: return f16_to_f32(f32_to_f16(v));
    }
}

/*
 * Deterministic expf shared bit-for-bit with the HIP kernels
 * (lvllm_amd/csrc/lkm_math.h: lkm_expf).  The reference calls CUDA expf
 * (topk_softmax_kernels.cu:428, activation_kernels.cu act fns), whose bits
 * no CPU can reproduce; fixing the operation sequence (Cody-Waite reduction +
 * degree-7 polynomial, explicit fmaf, no contraction) makes routing weights
 * reproducible between this oracle and the GPU.  |rel err| <~ 1 ulp on
 * [-104, 88] (checked against libm in tests/test_oracle.py).
 */
LKM_OR_API float lkm_or_expf(float x) {
    if (!(x == x)) return x;
    if (x > 88.72283f) return INFINITY;
    if (x < -103.97f) return 0.0f;
    const float LOG2E = 1.44269504088896341f;
    const float LN2_HI = 0.693145751953125f;       /* 0x3f317200 */
    const float LN2_LO = 1.42860682030941723e-6f;  /* ln2 - LN2_HI */
    float n = nearbyintf(x * LOG2E);
    float r = fmaf(-n, LN2_HI, x);
    r = fmaf(-n, LN2_LO, r);
    float p = 1.9841270e-4f;                       /* 1/5040 */
    p = fmaf(p, r, 1.3888889e-3f);                 /* 1/720 */
    p = fmaf(p, r, 8.3333338e-3f);                 /* 1/120 */
    p = fmaf(p, r, 4.1666668e-2f);                 /* 1/24  */
    p = fmaf(p, r, 1.6666667e-1f);                 /* 1/6   */
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    /* scale by 2^n in two exact steps so subnormal results round once */
    int ni = (int)n;
    int n1 = ni / 2, n2 = ni - n1;
    float s1 = u32_as_f32((uint32_t)(n1 + 127) << 23);
    float s2 = u32_as_f32((uint32_t)(n2 + 127) << 23);
    return (p * s1) * s2;
}

/* ------------------------------------------------------------------ routing */

/*
 * a1: fused softmax/sigmoid top-k.
 * Follows csrc/libtorch_stable/moe/topk_softmax_kernels.cu:408-592:
 *   softmax: max -> expf(x-max) -> sum -> * (1/sum)        (:408-451)
 *   sigmoid: 1/(1+exp(-x))                                  (:452-458)
 *   NaN/Inf scores -> 0                                     (:466-471)
 *   choice = score + bias                                   (:478-493)
 *   k rounds of arg-max on choice, strict '>' scanning ascending index,
 *   ties -> lowest index                                    (:500-543)
 *   weight = unbiased score; renorm: scale = rsf / (sum>0?sum:1) (:581-592)
 * Reduction order is pinned to the HIP kernel's: expert e lives in lane e%64,
 * lanes sum their own elements in ascending e, then a 32,16,..,1 xor butterfly.
 * ids are int32; pinned by tests/kernels/moe/test_fused_topk.py:18-91.
 */
LKM_OR_API void lkm_or_topk_softmax(const void* logits, int in_dtype, const float* bias,
                                    int M, int E, int K, int scoring, int renormalize,
                                    float routed_scaling, float* out_w, int32_t* out_ids) {
    float* sc = (float*)malloc(sizeof(float) * (size_t)E);
    float* ch = (float*)malloc(sizeof(float) * (size_t)E);
    for (int m = 0; m < M; ++m) {
        for (int e = 0; e < E; ++e) sc[e] = load_act(logits, in_dtype, (size_t)m * E + e);
        if (scoring == 0) {
            float mx = sc[0];
            for (int e = 1; e < E; ++e) mx = fmaxf(mx, sc[e]); /* max ignores NaN like v_max_f32 */
            float lane[64];
            for (int l = 0; l < 64; ++l) lane[l] = 0.0f;
            for (int e = 0; e < E; ++e) {
                sc[e] = lkm_or_expf(sc[e] - mx);
                lane[e & 63] += sc[e];
            }
            for (int mask = 32; mask > 0; mask >>= 1) {
                float t[64];
                for (int l = 0; l < 64; ++l) t[l] = lane[l] + lane[l ^ mask];
                memcpy(lane, t, sizeof(t));
            }
            float rinv = 1.0f / lane[0];
            for (int e = 0; e < E; ++e) sc[e] = sc[e] * rinv;
        } else {
            for (int e = 0; e < E; ++e) sc[e] = 1.0f / (1.0f + lkm_or_expf(-sc[e]));
        }
        for (int e = 0; e < E; ++e) {
            if (isnan(sc[e]) || isinf(sc[e])) sc[e] = 0.0f;
            ch[e] = bias ? sc[e] + bias[e] : sc[e];
        }
        float sel_sum = 0.0f;
        for (int k = 0; k < K; ++k) {
            int best = 0;
            float bv = ch[0];
            for (int e = 1; e < E; ++e)
                if (ch[e] > bv) { bv = ch[e]; best = e; }
            out_w[(size_t)m * K + k] = sc[best];
            out_ids[(size_t)m * K + k] = best;
            if (renormalize) sel_sum += sc[best];
            ch[best] = -INFINITY;
        }
        float scale = routed_scaling;
        if (renormalize) scale /= (sel_sum > 0.0f ? sel_sum : 1.0f);
        for (int k = 0; k < K; ++k) out_w[(size_t)m * K + k] *= scale;
    }
    free(sc); free(ch);
}

/*
 * a2: group-limited top-k (DeepSeek-V3 / GLM-4.5).
 * Follows vllm/model_executor/layers/fused_moe/router/grouped_topk_router.py:112-161:
 *   scores = softmax|sigmoid(logits) in fp32; choice = scores + bias
 *   group score = sum of top-2 choice per group (bias) | max per group (no bias)
 *   keep topk_group groups (ties -> lowest group index), mask the rest to -inf
 *   top-k over masked choice (ties -> lowest index), descending order
 *   weights = unbiased scores; renorm: w / sum(w); then * routed_scaling if != 1
 * torch.topk(sorted=False) leaves the output ORDER unspecified; this restatement
 * (and the HIP kernel) emit descending-choice order, tests compare as sets.
 */
LKM_OR_API void lkm_or_grouped_topk(const void* logits, int in_dtype, const float* bias,
                                    int M, int E, int K, int n_group, int topk_group,
                                    int scoring, int renormalize, float routed_scaling,
                                    float* out_w, int32_t* out_ids) {
    float* sc = (float*)malloc(sizeof(float) * (size_t)E);
    float* ch = (float*)malloc(sizeof(float) * (size_t)E);
    float* gs = (float*)malloc(sizeof(float) * (size_t)n_group);
    char* keep = (char*)malloc((size_t)n_group);
    const int gsz = E / n_group;
    for (int m = 0; m < M; ++m) {
        for (int e = 0; e < E; ++e) sc[e] = load_act(logits, in_dtype, (size_t)m * E + e);
        if (scoring == 0) {
            float mx = sc[0];
            for (int e = 1; e < E; ++e) mx = fmaxf(mx, sc[e]);
            float lane[64];
            for (int l = 0; l < 64; ++l) lane[l] = 0.0f;
            for (int e = 0; e < E; ++e) { sc[e] = lkm_or_expf(sc[e] - mx); lane[e & 63] += sc[e]; }
            for (int mask = 32; mask > 0; mask >>= 1) {
                float t[64];
                for (int l = 0; l < 64; ++l) t[l] = lane[l] + lane[l ^ mask];
                memcpy(lane, t, sizeof(t));
            }
            float rinv = 1.0f / lane[0];
            for (int e = 0; e < E; ++e) sc[e] = sc[e] * rinv;
        } else {
            for (int e = 0; e < E; ++e) sc[e] = 1.0f / (1.0f + lkm_or_expf(-sc[e]));
        }
        for (int e = 0; e < E; ++e) ch[e] = bias ? sc[e] + bias[e] : sc[e];
        for (int g = 0; g < n_group; ++g) {
            const float* c = ch + (size_t)g * gsz;
            if (bias) {
                float a = -INFINITY, b = -INFINITY;     /* top-2 */
                for (int i = 0; i < gsz; ++i) {
                    if (c[i] > a) { b = a; a = c[i]; }
                    else if (c[i] > b) b = c[i];
                }
                gs[g] = (gsz > 1) ? a + b : a;
            } else {
                float a = c[0];
                for (int i = 1; i < gsz; ++i) a = (c[i] > a) ? c[i] : a;
                gs[g] = a;
            }
            keep[g] = 0;
        }
        for (int t = 0; t < topk_group; ++t) {
            int best = -1; float bv = 0.0f;
            for (int g = 0; g < n_group; ++g) {
                if (keep[g]) continue;
                if (best < 0 || gs[g] > bv) { best = g; bv = gs[g]; }
            }
            keep[best] = 1;
        }
        for (int e = 0; e < E; ++e) if (!keep[e / gsz]) ch[e] = -INFINITY;
        float sum = 0.0f;
        for (int k = 0; k < K; ++k) {
            int best = 0; float bv = ch[0];
            for (int e = 1; e < E; ++e) if (ch[e] > bv) { bv = ch[e]; best = e; }
            out_w[(size_t)m * K + k] = sc[best];
            out_ids[(size_t)m * K + k] = best;
            sum += sc[best];
            ch[best] = -INFINITY;
        }
        for (int k = 0; k < K; ++k) {
            float w = out_w[(size_t)m * K + k];
            if (renormalize) w = w / sum;
            if (routed_scaling != 1.0f) w = w * routed_scaling;
            out_w[(size_t)m * K + k] = w;
        }
    }
    free(sc); free(ch); free(gs); free(keep);
}

/*
 * EP expert placement: vllm/model_executor/layers/fused_moe/expert_map_manager.py:62-92.
 * strategy 0 = linear, 1 = round_robin.  Returns the local expert count.
 * (The code gives the remainder to the FIRST ranks, whatever its docstring says.)
 */
LKM_OR_API int lkm_or_expert_map(int ep_size, int ep_rank, int E, int strategy, int32_t* map) {
    int base = E / ep_size, rem = E % ep_size;
    int local = base + (ep_rank < rem ? 1 : 0);
    for (int e = 0; e < E; ++e) map[e] = -1;
    if (strategy == 0) {
        int start = ep_rank * base + (ep_rank < rem ? ep_rank : rem);
        for (int i = 0; i < local; ++i) map[start + i] = i;
    } else {
        int i = 0;
        for (int e = ep_rank; e < E; e += ep_size) map[e] = i++;
    }
    return local;
}

/* a3: RoutedExperts.global_to_local_expert_ids, routed_experts.py:1332-1342 */
LKM_OR_API void lkm_or_map_ids(const int32_t* ids, int64_t n, const int32_t* map, int E,
                               int32_t* out) {
    for (int64_t i = 0; i < n; ++i) {
        int32_t id = ids[i];
        int32_t c = id < 0 ? 0 : (id > E - 1 ? E - 1 : id);
        out[i] = id < 0 ? -1 : map[c];
    }
}

/*
 * Token -> expert scatter (stable counting sort of the M*K slots by expert).
 * Follows csrc/cpu/cpu_fused_moe.cpp:200-227 (count, exclusive prefix, scatter in
 * ascending flat-slot order) == torch.sort(stable=True) of
 * tests/kernels/moe/test_moe_permute_unpermute.py:52-55.  ids < 0 or >= E are
 * skipped (lk_moe contract: -1 = non-local / padding).
 */
LKM_OR_API void lkm_or_sort(const int32_t* ids, int n_slots, int E, int32_t* counts,
                            int32_t* offsets, int32_t* sorted_slot, int32_t* pos_of_slot) {
    for (int e = 0; e < E; ++e) counts[e] = 0;
    for (int i = 0; i < n_slots; ++i)
        if (ids[i] >= 0 && ids[i] < E) counts[ids[i]]++;
    offsets[0] = 0;
    for (int e = 0; e < E; ++e) offsets[e + 1] = offsets[e] + counts[e];
    int32_t* cur = (int32_t*)malloc(sizeof(int32_t) * (size_t)(E > 0 ? E : 1));
    for (int e = 0; e < E; ++e) cur[e] = offsets[e];
    for (int i = 0; i < n_slots; ++i) {
        int id = ids[i];
        if (id >= 0 && id < E) {
            int p = cur[id]++;
            sorted_slot[p] = i;
            pos_of_slot[i] = p;
        } else {
            pos_of_slot[i] = -1;
        }
    }
    for (int p = offsets[E]; p < n_slots; ++p) sorted_slot[p] = -1;
    free(cur);
}

/* ------------------------------------------------------------------ experts */

typedef struct {
    int32_t E, H, I;            /* local experts, hidden, intermediate (per partition) */
    int32_t has_gate;           /* 1: w13 = [E,2I,H]; 0: w13 = [E,I,H] */
    int32_t activation;         /* OR_ACT_* */
    float swiglu_alpha, swiglu_limit;
    int32_t act_dtype;          /* OR_BF16 | OR_F16 : dtype of hidden states + intermediate */
    int32_t wfmt;               /* OR_W_* */
    int32_t groupN, groupK;     /* quant block shape; int4: groupN=1 */
    int32_t round_gemm1;        /* 1: round GEMM1 output to act dtype before the activation
                                   (in-tree GPU operator, fused_moe.py:1735-1819);
                                   0: keep fp32 (CPU path, test_cpu_fused_moe.py:86-91) */
    int32_t w8a8;               /* fp8 only: 1 = dynamic 1xgroupK activation quant (W8A8,
                                   tests/kernels/moe/test_block_fp8.py:107-137) */
    const float* gs13;          /* NVFP4 only: per-expert f32 multipliers [E] (the reference passes
                                   1/global_scale when need_reciprocal_global_scale,            */
    const float* gs2;           /*   routed_experts.py:1686-1688); NULL = 1.0 */
    int32_t int4_unrounded;     /* int4 only, 1 = weight (q-8)*s kept in fp32 (exact) instead of rounded to the act
                                   dtype: the result of applying the group scale to fp32 partial sums -- NOT the
                                   reference's semantics (0), a checker for a candidate kernel mode that stays
                                   inside the reference's tolerance (tests/test_oracle_golden.py) */
} OrMoeDesc;

/* FP4 E2M1 magnitudes (tests/kernels/quantization/nvfp4_utils.py:11-13 kE2M1ToFloat;
   tests/quantization/reference_mxfp4.py:39-88): code = sign<<3 | index */
static const float kE2M1[8] = {0.0f, 0.5f, 1.0f, 1.5f, 2.0f, 3.0f, 4.0f, 6.0f};
static inline float fp4_to_f32(unsigned c) { return (c & 8u) ? -kE2M1[c & 7u] : kE2M1[c & 7u]; }

/* dequantise one weight row [K] to f32.  rows are [N][K] (K contiguous). */
static void dequant_row(const OrMoeDesc* d, const void* w, const void* scale, const float* gs,
                        int64_t e, int64_t N, int64_t K, int64_t n, float* out, int apply_scale) {
    switch (d->wfmt) {
    case OR_W_MXFP4: {
        /* OCP MXFP4 (routed_experts.py:1747-1813; reference_mxfp4.py:91-117 dq_mxfp4_torch):
           bytes [E,N,K/2] low nibble = even k, E2M1; scales uint8 E8M0 [E,N,K/32];
           w = T(fp4) * T(2^(s-127)) computed in T -- exact unless the product leaves T's range */
        const uint8_t* p = (const uint8_t*)w + ((size_t)e * N + n) * (K / 2);
        const uint8_t* sc = (const uint8_t*)scale + ((size_t)e * N + n) * (K / 32);
        for (int64_t k = 0; k < K; ++k) {
            float v = fp4_to_f32((p[k >> 1] >> ((k & 1) * 4)) & 0xfu);
            float sf = round_act(ldexpf(1.0f, (int)sc[k / 32] - 127), d->act_dtype);
            out[k] = round_act(v * sf, d->act_dtype);
        }
    } break;
    case OR_W_NVFP4: {
        /* NVFP4 (routed_experts.py:1673-1745; nvfp4_utils.py:39-66 dequantize_nvfp4_to_dtype with
           the scale factors in linear layout): bytes [E,N,K/2] E2M1, block scales fp8 e4m3fn
           [E,N,K/16], per-expert f32 multiplier gs[e]:  w = T(fp4 * (f32(sf) * gs)) */
        const uint8_t* p = (const uint8_t*)w + ((size_t)e * N + n) * (K / 2);
        const uint8_t* sc = (const uint8_t*)scale + ((size_t)e * N + n) * (K / 16);
        const float g = gs ? gs[e] : 1.0f;
        for (int64_t k = 0; k < K; ++k) {
            float v = fp4_to_f32((p[k >> 1] >> ((k & 1) * 4)) & 0xfu);
            float sf = fp8e4m3_to_f32(sc[k / 16]) * g;
            out[k] = round_act(v * sf, d->act_dtype);
        }
    } break;
    case OR_W_BF16: {
        const uint16_t* p = (const uint16_t*)w + ((size_t)e * N + n) * K;
        for (int64_t k = 0; k < K; ++k) out[k] = bf16_to_f32(p[k]);
    } break;
    case OR_W_F16: {
        const uint16_t* p = (const uint16_t*)w + ((size_t)e * N + n) * K;
        for (int64_t k = 0; k < K; ++k) out[k] = f16_to_f32(p[k]);
    } break;
    case OR_W_FP8_E4M3: {
        /* fp8.py:570-652 layout: w [E,N,K] e4m3fn, scale fp32 [E, ceil(N/gN), ceil(K/gK)];
           dequant w_f32 = fp8 * scale (SURVEY 8c "fp8 W8A16") */
        const uint8_t* p = (const uint8_t*)w + ((size_t)e * N + n) * K;
        int64_t gN = d->groupN, gK = d->groupK;
        int64_t nb = (N + gN - 1) / gN, kb = (K + gK - 1) / gK;
        const float* s = (const float*)scale + ((size_t)e * nb + n / gN) * kb;
        for (int64_t k = 0; k < K; ++k) {
            float v = fp8e4m3_to_f32(p[k]);
            out[k] = apply_scale ? v * s[k / gK] : v;
        }
    } break;
    default: {
        /* uint4b8 (compressed_tensors_wNa16.py:38-46; routed_experts.py:1461-1479):
           bytes [E,N,K/2], low nibble = even k, stored value = q+8;
           scales act-dtype [E,N,K/g]; dequant = T((nib-8) * scale_f32)
           (fused_moe.py:207-208,237-276; quant_utils.py:706 w_ref) */
        const uint8_t* p = (const uint8_t*)w + ((size_t)e * N + n) * (K / 2);
        /* groupK >= K: one group per row (channel-wise scales [E,N,1]; the reference passes max(groupK of w13, w2),
           routed_experts.py:1440-1453) */
        int64_t gK = d->groupK, kb = (K + gK - 1) / gK;
        const uint16_t* s = (const uint16_t*)scale + ((size_t)e * N + n) * kb;
        for (int64_t k = 0; k < K; ++k) {
            int q = (p[k >> 1] >> ((k & 1) * 4)) & 0xf;
            float sf = d->act_dtype == OR_BF16 ? bf16_to_f32(s[k / gK]) : f16_to_f32(s[k / gK]);
            out[k] = d->int4_unrounded ? (float)(q - 8) * sf : round_act((float)(q - 8) * sf, d->act_dtype);
        }
    } break;
    }
}

/* dot of one f32 weight row against R f32 activation rows.  16-lane partial sums per row, rows
   processed 8 at a time so that the 8 accumulator chains are independent (the summation order per
   row is fixed -> deterministic; vectorises without -ffast-math). */
__attribute__((target_clones("avx512f", "default")))
static void dot_rows(const float* w, const float* x, int64_t ldx, int R, int64_t K,
                     float* out) {
    for (int r0 = 0; r0 < R; r0 += 8) {
        const int rn = R - r0 < 8 ? R - r0 : 8;
        float acc[8][16];
        for (int r = 0; r < 8; ++r)
            for (int j = 0; j < 16; ++j) acc[r][j] = 0.0f;
        int64_t k = 0;
        if (rn == 8) {
            for (; k + 16 <= K; k += 16)
                for (int r = 0; r < 8; ++r) {
                    const float* xr = x + (size_t)(r0 + r) * ldx + k;
                    for (int j = 0; j < 16; ++j) acc[r][j] += w[k + j] * xr[j];
                }
        } else {
            for (; k + 16 <= K; k += 16)
                for (int r = 0; r < rn; ++r) {
                    const float* xr = x + (size_t)(r0 + r) * ldx + k;
                    for (int j = 0; j < 16; ++j) acc[r][j] += w[k + j] * xr[j];
                }
        }
        for (int r = 0; r < rn; ++r) {
            const float* xr = x + (size_t)(r0 + r) * ldx;
            float t = 0.0f;
            for (int64_t kk = k; kk < K; ++kk) t += w[kk] * xr[kk];
            for (int s = 8; s > 0; s >>= 1)
                for (int j = 0; j < s; ++j) acc[r][j] += acc[r][j + s];
            out[r0 + r] = acc[r][0] + t;
        }
    }
}

static inline float act_silu(float g) { return g / (1.0f + lkm_or_expf(-g)); }

/* dynamic per-token-group fp8 quant of one row segment, tests/kernels/quant_utils.py:157-180:
   amax.clamp(1e-10)/448 -> scale; q = clamp(x/scale, +-448).to(e4m3fn); returns scale,
   writes the de-scaled fp8 VALUES (as f32) to q */
static float quant_group_fp8(const float* x, int64_t n, float* q) {
    float amax = 0.0f;
    for (int64_t i = 0; i < n; ++i) { float a = fabsf(x[i]); amax = a > amax ? a : amax; }
    if (amax < 1e-10f) amax = 1e-10f;
    float s = amax / 448.0f;
    for (int64_t i = 0; i < n; ++i) {
        float v = x[i] / s;
        v = v > 448.0f ? 448.0f : (v < -448.0f ? -448.0f : v);
        q[i] = fp8e4m3_to_f32(f32_to_fp8e4m3(v));
    }
    return s;
}

/*
 * a6/a7/a8/a9: routed experts  y[m] = sum_k w[m,k] * W2[e] * act(W13[e] * x[m]).
 * Structure follows csrc/cpu/cpu_fused_moe.cpp:200-635 (count-sort -> w13 GEMM + act with an
 * act-dtype-rounded intermediate -> w2 GEMM in fp32 -> fp32 weighted sum), rounding points of
 * tests/kernels/moe/test_cpu_fused_moe.py:46-107 (round_gemm1=0) or of the in-tree GPU operator
 * tests/kernels/utils.py:855-1021 torch_experts (round_gemm1=1: GEMM1 and GEMM2 outputs rounded
 * to the act dtype, fp32 weighted sum :989-993).  SiLU-mul rounding: activation_kernels.cu:57-75,
 * 157-160 (T(silu_f32(g)) * u when round_gemm1, else fp32 product rounded once).
 * out is fp32 [M,H] (cpu_decode / cpu_prefill contract, routed_experts.py:1840-1882).
 * ids: int32 local ids, <0 = skip.  All rows of out are written (zeros if nothing routed).
 * Weight layouts: SURVEY 8(a5).
 */
LKM_OR_API int lkm_or_moe(const OrMoeDesc* d, const void* w13, const void* w2, const void* s13,
                          const void* s2, const void* x, const int32_t* ids, const float* tw,
                          int M, int K, float* out) {
    const int64_t E = d->E, H = d->H, I = d->I;
    const int64_t N1 = d->has_gate ? 2 * I : I;
    const int n_slots = M * K;
    int32_t* counts = (int32_t*)malloc(sizeof(int32_t) * (size_t)(E + 1));
    int32_t* offs = (int32_t*)malloc(sizeof(int32_t) * (size_t)(E + 2));
    int32_t* sorted = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_slots + 1));
    int32_t* pos = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_slots + 1));
    lkm_or_sort(ids, n_slots, (int)E, counts, offs, sorted, pos);
    const int n_rows = offs[E];

    /* gather activations as f32, expert-sorted rows */
    float* xs = (float*)malloc(sizeof(float) * (size_t)(n_rows > 0 ? n_rows : 1) * H);
    float* xscale = NULL;                       /* w8a8: per row per k-block scales */
    const int64_t gK = d->groupK > 0 ? d->groupK : 1;
    const int64_t kb1 = (H + gK - 1) / gK, kb2 = (I + gK - 1) / gK;
    if (d->w8a8) xscale = (float*)malloc(sizeof(float) * (size_t)(n_rows > 0 ? n_rows : 1) * kb1);
#pragma omp parallel for schedule(static)
    for (int r = 0; r < n_rows; ++r) {
        int tok = sorted[r] / K;
        float* dst = xs + (size_t)r * H;
        for (int64_t k = 0; k < H; ++k) dst[k] = load_act(x, d->act_dtype, (size_t)tok * H + k);
        if (d->w8a8)
            for (int64_t b = 0; b < kb1; ++b) {
                int64_t n = (b + 1) * gK <= H ? gK : H - b * gK;
                xscale[(size_t)r * kb1 + b] = quant_group_fp8(dst + b * gK, n, dst + b * gK);
            }
    }

    float* g1 = (float*)malloc(sizeof(float) * (size_t)(n_rows > 0 ? n_rows : 1) * N1);
    float* act = (float*)malloc(sizeof(float) * (size_t)(n_rows > 0 ? n_rows : 1) * I);
    float* ascale = NULL;
    if (d->w8a8) ascale = (float*)malloc(sizeof(float) * (size_t)(n_rows > 0 ? n_rows : 1) * kb2);
    float* y = (float*)malloc(sizeof(float) * (size_t)(n_rows > 0 ? n_rows : 1) * H);

    /* ---- GEMM1: tasks = (expert, weight row) */
#pragma omp parallel
    {
        float* wrow = (float*)malloc(sizeof(float) * (size_t)(H > I ? H : I));
        float* tmp = (float*)malloc(sizeof(float) * 4096);
#pragma omp for schedule(runtime)
        for (int64_t task = 0; task < E * N1; ++task) {
            int64_t e = task / N1, n = task % N1;
            int R = counts[e];
            if (R == 0) continue;
            const float* xe = xs + (size_t)offs[e] * H;
            float* ge = g1 + (size_t)offs[e] * N1;
            dequant_row(d, w13, s13, d->gs13, e, N1, H, n, wrow, !d->w8a8);
            for (int r0 = 0; r0 < R; r0 += 4096) {
                int rr = R - r0 < 4096 ? R - r0 : 4096;
                if (!d->w8a8) {
                    dot_rows(wrow, xe + (size_t)r0 * H, H, rr, H, tmp);
                } else {
                    /* native_w8a8_block_matmul, tests/kernels/quant_utils.py:91-154:
                       C += (A_blk . B_blk) * (As * Bs) per k-block */
                    const int64_t gN = d->groupN, nb = (N1 + gN - 1) / gN;
                    const float* ws = (const float*)s13 + ((size_t)e * nb + n / gN) * kb1;
                    for (int r = 0; r < rr; ++r) {
                        float c = 0.0f;
                        for (int64_t b = 0; b < kb1; ++b) {
                            int64_t k0 = b * gK, kn = k0 + gK <= H ? gK : H - k0;
                            float p;
                            dot_rows(wrow + k0, xe + (size_t)(r0 + r) * H + k0, H, 1, kn, &p);
                            c += p * (xscale[(size_t)(offs[e] + r0 + r) * kb1 + b] * ws[b]);
                        }
                        tmp[r] = c;
                    }
                }
                for (int r = 0; r < rr; ++r) {
                    float v = tmp[r];
                    if (d->round_gemm1) v = round_act(v, d->act_dtype);
                    ge[(size_t)(r0 + r) * N1 + n] = v;
                }
            }
        }
        free(wrow); free(tmp);
    }

    /* ---- activation */
#pragma omp parallel for schedule(static)
    for (int r = 0; r < n_rows; ++r) {
        const float* g = g1 + (size_t)r * N1;
        float* a = act + (size_t)r * I;
        for (int64_t i = 0; i < I; ++i) {
            float v;
            if (d->activation == OR_ACT_SILU && d->has_gate) {
                /* gate = first half, up = second half (activation_kernels.cu:95-150) */
                if (d->round_gemm1) v = round_act(act_silu(g[i]), d->act_dtype) * g[I + i];
                else v = act_silu(g[i]) * g[I + i];
            } else if (d->activation == OR_ACT_SWIGLUOAI) {
                /* interleaved gate/up, activation_kernels.cu:401-408 */
                float gg = fminf(g[2 * i], d->swiglu_limit);
                float uu = fmaxf(fminf(g[2 * i + 1], d->swiglu_limit), -d->swiglu_limit);
                v = (uu + 1.0f) * gg / (1.0f + lkm_or_expf(-gg * d->swiglu_alpha));
            } else {
                /* relu2, no gate: activation.py:208-210 */
                float t = g[i] > 0.0f ? g[i] : 0.0f;
                v = t * t;
            }
            a[i] = round_act(v, d->act_dtype);
        }
        if (d->w8a8)
            for (int64_t b = 0; b < kb2; ++b) {
                int64_t n = (b + 1) * gK <= I ? gK : I - b * gK;
                ascale[(size_t)r * kb2 + b] = quant_group_fp8(a + b * gK, n, a + b * gK);
            }
    }

    /* ---- GEMM2: tasks = (expert, output row h) */
#pragma omp parallel
    {
        float* wrow = (float*)malloc(sizeof(float) * (size_t)(H > I ? H : I));
        float* tmp = (float*)malloc(sizeof(float) * 4096);
#pragma omp for schedule(runtime)
        for (int64_t task = 0; task < E * H; ++task) {
            int64_t e = task / H, h = task % H;
            int R = counts[e];
            if (R == 0) continue;
            const float* ae = act + (size_t)offs[e] * I;
            dequant_row(d, w2, s2, d->gs2, e, H, I, h, wrow, !d->w8a8);
            for (int r0 = 0; r0 < R; r0 += 4096) {
                int rr = R - r0 < 4096 ? R - r0 : 4096;
                if (!d->w8a8) {
                    dot_rows(wrow, ae + (size_t)r0 * I, I, rr, I, tmp);
                } else {
                    const int64_t gN = d->groupN, nb = (H + gN - 1) / gN;
                    const float* ws = (const float*)s2 + ((size_t)e * nb + h / gN) * kb2;
                    for (int r = 0; r < rr; ++r) {
                        float c = 0.0f;
                        for (int64_t b = 0; b < kb2; ++b) {
                            int64_t k0 = b * gK, kn = k0 + gK <= I ? gK : I - k0;
                            float p;
                            dot_rows(wrow + k0, ae + (size_t)(r0 + r) * I + k0, I, 1, kn, &p);
                            c += p * (ascale[(size_t)(offs[e] + r0 + r) * kb2 + b] * ws[b]);
                        }
                        tmp[r] = c;
                    }
                }
                for (int r = 0; r < rr; ++r) {
                    float v = tmp[r];
                    if (d->round_gemm1) v = round_act(v, d->act_dtype);
                    y[(size_t)(offs[e] + r0 + r) * H + h] = v;
                }
            }
        }
        free(wrow); free(tmp);
    }

    /* ---- combine: fp32 weighted sum over the K slots, ascending k
       (cpu_fused_moe.cpp:524-635; finalizeMoeRoutingKernel .inl:91-143) */
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        float* o = out + (size_t)m * H;
        for (int64_t h = 0; h < H; ++h) o[h] = 0.0f;
        for (int k = 0; k < K; ++k) {
            int p = pos[m * K + k];
            if (p < 0) continue;
            float wk = tw[m * K + k];
            const float* yr = y + (size_t)p * H;
            for (int64_t h = 0; h < H; ++h) o[h] += wk * yr[h];
        }
    }

    free(counts); free(offs); free(sorted); free(pos); free(xs); free(g1); free(act); free(y);
    if (xscale) free(xscale);
    if (ascale) free(ascale);
    return 0;
}

/* ------------------------------------------------------------------ quantisers (test inputs) */

/*
 * Symmetric uint4b8 group quantisation of w [N,K] (act dtype) along K.
 * Follows quant_utils.py:642-738 quantize_weights(w.T, uint4b8, g) as used by
 * tests/kernels/moe/test_moe.py:634-641: s = max(|max/7|, |min/-8|) per group,
 * q = clamp(round(w/s), -8, 7) + 8, packed low nibble = even k.
 * Arithmetic is done in the act dtype like torch does on bf16 tensors.
 */
LKM_OR_API void lkm_or_quant_int4(const uint16_t* w, int act_dtype, int64_t N, int64_t K, int g,
                                  uint8_t* packed, uint16_t* scales) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        for (int64_t b = 0; b < K / g; ++b) {
            float mx = -INFINITY, mn = INFINITY;
            for (int j = 0; j < g; ++j) {
                float v = load_act(w, act_dtype, (size_t)n * K + b * g + j);
                mx = v > mx ? v : mx; mn = v < mn ? v : mn;
            }
            float s1 = fabsf(round_act(mx / 7.0f, act_dtype));
            float s2 = fabsf(round_act(mn / -8.0f, act_dtype));
            float s = s1 > s2 ? s1 : s2;
            scales[(size_t)n * (K / g) + b] =
                act_dtype == OR_BF16 ? f32_to_bf16(s) : f32_to_f16(s);
            for (int j = 0; j < g; ++j) {
                int64_t k = b * g + j;
                float v = load_act(w, act_dtype, (size_t)n * K + k);
                float qf = nearbyintf(round_act(v / s, act_dtype));
                int q = (int)qf;
                q = q < -8 ? -8 : (q > 7 ? 7 : q);
                q += 8;
                uint8_t* p = packed + (size_t)n * (K / 2) + (k >> 1);
                if (k & 1) *p = (uint8_t)((*p & 0x0f) | (q << 4));
                else *p = (uint8_t)((*p & 0xf0) | q);
            }
        }
    }
}

/* 128x128-style block fp8 quantisation: scale = amax/448 per block, q = (w/scale).to(e4m3fn)
   (SURVEY 8d synthetic inputs; per_block_cast_to_fp8 in tests/kernels/quant_utils.py) */
LKM_OR_API void lkm_or_quant_fp8_block(const float* w, int64_t N, int64_t K, int gN, int gK,
                                       uint8_t* q, float* scales) {
    int64_t nb = (N + gN - 1) / gN, kb = (K + gK - 1) / gK;
#pragma omp parallel for schedule(static)
    for (int64_t bi = 0; bi < nb * kb; ++bi) {
        int64_t bn = bi / kb, bk = bi % kb;
        float amax = 0.0f;
        for (int64_t n = bn * gN; n < (bn + 1) * gN && n < N; ++n)
            for (int64_t k = bk * gK; k < (bk + 1) * gK && k < K; ++k) {
                float a = fabsf(w[(size_t)n * K + k]);
                amax = a > amax ? a : amax;
            }
        if (amax < 1e-4f) amax = 1e-4f;
        float s = amax / 448.0f;
        scales[bi] = s;
        for (int64_t n = bn * gN; n < (bn + 1) * gN && n < N; ++n)
            for (int64_t k = bk * gK; k < (bk + 1) * gK && k < K; ++k)
                q[(size_t)n * K + k] = f32_to_fp8e4m3(w[(size_t)n * K + k] / s);
    }
}

LKM_OR_API void lkm_or_cvt_f32_to(const float* src, int dtype, int64_t n, void* dst) {
    for (int64_t i = 0; i < n; ++i) {
        if (dtype == OR_BF16) ((uint16_t*)dst)[i] = f32_to_bf16(src[i]);
        else if (dtype == OR_F16) ((uint16_t*)dst)[i] = f32_to_f16(src[i]);
        else ((float*)dst)[i] = src[i];
    }
}
LKM_OR_API void lkm_or_cvt_to_f32(const void* src, int dtype, int64_t n, float* dst) {
    for (int64_t i = 0; i < n; ++i) dst[i] = load_act(src, dtype, (size_t)i);
}
LKM_OR_API void lkm_or_fp8_to_f32(const uint8_t* src, int64_t n, float* dst) {
    for (int64_t i = 0; i < n; ++i) dst[i] = fp8e4m3_to_f32(src[i]);
}
LKM_OR_API void lkm_or_f32_to_fp8(const float* src, int64_t n, uint8_t* dst) {
    for (int64_t i = 0; i < n; ++i) dst[i] = f32_to_fp8e4m3(src[i]);
}

/* threads > 0 sets the OpenMP team size; the GEMM task loops use schedule(runtime) = dynamic,64 */
/* router GEMM: logits[M,E] = x[M,H] . w[E,H]^T (+ bias), the reference's F.linear in fp32
   (tests/kernels/test_fp32_router_gemm.py:35-37); accumulated in double so that the oracle is the
   exact value rounded once; optional rounding to the gate's output dtype. */
LKM_OR_API void lkm_or_router_logits(const void* x, int x_dtype, const void* w, int w_dtype,
                                     const float* bias, int M, int H, int E, int round_dtype,
                                     float* out) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m)
        for (int e = 0; e < E; ++e) {
            double acc = 0.0;
            for (int k = 0; k < H; ++k)
                acc += (double)load_act(x, x_dtype, (size_t)m * H + k) * (double)load_act(w, w_dtype, (size_t)e * H + k);
            float v = (float)acc;
            if (bias) v = v + bias[e];
            if (round_dtype != OR_F32) v = round_act(v, round_dtype);
            out[(size_t)m * E + e] = v;
        }
}

/* dequantise E x N rows of a packed 4-bit format to act-dtype bit patterns (golden pinning) */
LKM_OR_API void lkm_or_dequant_rows(int wfmt, int act_dtype, const void* w, const void* scale,
                                    const float* gs, int64_t E, int64_t N, int64_t K, int groupK,
                                    uint16_t* out) {
    OrMoeDesc d;
    memset(&d, 0, sizeof d);
    d.wfmt = wfmt;
    d.act_dtype = act_dtype;
    d.groupN = 1;
    d.groupK = groupK;
    float* row = (float*)malloc(sizeof(float) * (size_t)K);
    for (int64_t e = 0; e < E; ++e)
        for (int64_t n = 0; n < N; ++n) {
            dequant_row(&d, w, scale, gs, e, N, K, n, row, 1);
            uint16_t* o = out + ((size_t)e * N + n) * K;
            for (int64_t k = 0; k < K; ++k)
                o[k] = act_dtype == OR_BF16 ? f32_to_bf16(row[k]) : f32_to_f16(row[k]);
        }
    free(row);
}

LKM_OR_API void lkm_or_configure(int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    omp_set_schedule(omp_sched_dynamic, 64);
#else
    (void)threads;
#endif
}

LKM_OR_API int lkm_or_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
