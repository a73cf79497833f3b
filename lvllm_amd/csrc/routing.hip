// routing.hip -- router kernels for gfx950 (wave64): fused softmax/sigmoid top-k, group-limited
// top-k, global->local expert id map.
//
// What they compute follows the reference (paths relative to the reference tree):
//   topk_softmax   csrc/libtorch_stable/moe/topk_softmax_kernels.cu:408-592
//   grouped_topk   vllm/model_executor/layers/fused_moe/router/grouped_topk_router.py:112-161
//   id map         vllm/model_executor/layers/fused_moe/routed_experts.py:1332-1342
// How: one 64-lane wavefront per token row, expert e lives in lane e%64 (register slot e/64),
// reductions are 64-wide xor butterflies.  The arithmetic sequence (lkm_expf, per-lane sums in
// ascending e then butterfly 32..1, 1/sum, score*rinv) is restated verbatim in
// oracle/lkm_oracle.c so ids AND weights are bit-reproducible; contraction is off for that reason.
#include "lkm_common.h"

namespace lkm {

constexpr int kMaxSlots = 8;  // E <= 512

__device__ __forceinline__ float load_logit(const void* p, int dt, size_t i) {
    if (dt == LKM_DT_F32) return ((const float*)p)[i];
    unsigned short h = ((const unsigned short*)p)[i];
    return dt == LKM_DT_BF16 ? bf16_bits_to_f32(h) : f16_bits_to_f32(h);
}

// scores for one row, in registers: sc[s] = score of expert s*64+lane (0 for e >= E)
__device__ __forceinline__ void row_scores(const void* logits, int dt, int row, int E, int lane,
                                           int scoring, float (&sc)[kMaxSlots]) {
#pragma clang fp contract(off)
    float v[kMaxSlots];
#pragma unroll
    for (int s = 0; s < kMaxSlots; ++s) {
        int e = s * 64 + lane;
        v[s] = (e < E) ? load_logit(logits, dt, (size_t)row * E + e) : -__builtin_inff();
    }
    if (scoring == 0) {
        float mx = v[0];
#pragma unroll
        for (int s = 1; s < kMaxSlots; ++s) mx = fmaxf(mx, v[s]);
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
        float sum = 0.0f;
#pragma unroll
        for (int s = 0; s < kMaxSlots; ++s) {
            int e = s * 64 + lane;
            if (e < E) {
                v[s] = lkm_expf(v[s] - mx);
                sum += v[s];
            } else {
                v[s] = 0.0f;
            }
        }
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) sum = sum + __shfl_xor(sum, m, 64);
        float rinv = 1.0f / sum;
#pragma unroll
        for (int s = 0; s < kMaxSlots; ++s) sc[s] = v[s] * rinv;
    } else {
#pragma unroll
        for (int s = 0; s < kMaxSlots; ++s) {
            int e = s * 64 + lane;
            sc[s] = (e < E) ? 1.0f / (1.0f + lkm_expf(-v[s])) : 0.0f;
        }
    }
}

// wave-wide arg-max of (value, index) with "lowest index wins ties"; also carries a payload.
__device__ __forceinline__ void wave_argmax(float& bv, int& be, float& bp) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        float ov = __shfl_xor(bv, m, 64);
        int oe = __shfl_xor(be, m, 64);
        float op = __shfl_xor(bp, m, 64);
        if (ov > bv || (ov == bv && oe < be)) {
            bv = ov;
            be = oe;
            bp = op;
        }
    }
}

__global__ __launch_bounds__(256) void topk_softmax_kernel(
    const void* __restrict__ logits, int dt, const float* __restrict__ bias, int M, int E, int K,
    int scoring, int renorm, float rsf, float* __restrict__ out_w, int32_t* __restrict__ out_ids) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float sc[kMaxSlots], ch[kMaxSlots];
    row_scores(logits, dt, row, E, lane, scoring, sc);
#pragma unroll
    for (int s = 0; s < kMaxSlots; ++s) {
        int e = s * 64 + lane;
        if (__builtin_isnan(sc[s]) || __builtin_isinf(sc[s])) sc[s] = 0.0f;  // :466-471
        if (e < E)
            ch[s] = bias ? sc[s] + bias[e] : sc[s];
        else
            ch[s] = -__builtin_inff();
    }
    float sel_sum = 0.0f;
    for (int k = 0; k < K; ++k) {
        float bv = ch[0], bp = sc[0];
        int be = lane;
#pragma unroll
        for (int s = 1; s < kMaxSlots; ++s) {
            if (ch[s] > bv) {
                bv = ch[s];
                bp = sc[s];
                be = s * 64 + lane;
            }
        }
        wave_argmax(bv, be, bp);
        if (lane == 0) {
            out_w[(size_t)row * K + k] = bp;
            out_ids[(size_t)row * K + k] = be;
            if (renorm) sel_sum += bp;
        }
#pragma unroll
        for (int s = 0; s < kMaxSlots; ++s)
            if (be == s * 64 + lane) ch[s] = -__builtin_inff();
    }
    if (lane == 0) {
        float scale = rsf;
        if (renorm) scale /= (sel_sum > 0.0f ? sel_sum : 1.0f);  // :581-592
        for (int k = 0; k < K; ++k) out_w[(size_t)row * K + k] *= scale;
    }
}

__global__ __launch_bounds__(256) void grouped_topk_kernel(
    const void* __restrict__ logits, int dt, const float* __restrict__ bias, int M, int E, int K,
    int n_group, int topk_group, int scoring, int renorm, float rsf, float* __restrict__ out_w,
    int32_t* __restrict__ out_ids) {
#pragma clang fp contract(off)
    __shared__ float lds_ch[4][kMaxSlots * 64];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wv;
    const bool active = row < M;  // keep the wave alive for __syncthreads
    const int gsz = E / n_group;
    float sc[kMaxSlots], ch[kMaxSlots];
    if (active) {
        row_scores(logits, dt, row, E, lane, scoring, sc);
#pragma unroll
        for (int s = 0; s < kMaxSlots; ++s) {
            int e = s * 64 + lane;
            ch[s] = (e < E) ? (bias ? sc[s] + bias[e] : sc[s]) : -__builtin_inff();
            lds_ch[wv][s * 64 + lane] = ch[s];
        }
    }
    __syncthreads();
    if (!active) return;
    // lane g scores group g in ascending expert order (same order as the oracle)
    float gs = -__builtin_inff();
    if (lane < n_group) {
        const float* c = &lds_ch[wv][lane * gsz];
        if (bias) {
            float a = -__builtin_inff(), b = -__builtin_inff();
            for (int i = 0; i < gsz; ++i) {
                float x = c[i];
                if (x > a) {
                    b = a;
                    a = x;
                } else if (x > b) {
                    b = x;
                }
            }
            gs = (gsz > 1) ? a + b : a;
        } else {
            float a = c[0];
            for (int i = 1; i < gsz; ++i) a = (c[i] > a) ? c[i] : a;
            gs = a;
        }
    }
    unsigned long long keep = 0ull;
    bool taken = !(lane < n_group);
    for (int t = 0; t < topk_group; ++t) {
        // lanes already taken / out of range must never win, not even on ties with -inf values
        float bv = gs, bp = 0.0f;
        int be = taken ? (1 << 20) + lane : lane;
        if (taken) bv = -__builtin_inff();
        wave_argmax(bv, be, bp);
        keep |= 1ull << (be & 63);
        if (lane == be) taken = true;
    }
#pragma unroll
    for (int s = 0; s < kMaxSlots; ++s) {
        int e = s * 64 + lane;
        if (e < E && !((keep >> (e / gsz)) & 1ull)) ch[s] = -__builtin_inff();
    }
    float sum = 0.0f;
    for (int k = 0; k < K; ++k) {
        float bv = ch[0], bp = sc[0];
        int be = lane;
#pragma unroll
        for (int s = 1; s < kMaxSlots; ++s) {
            if (ch[s] > bv) {
                bv = ch[s];
                bp = sc[s];
                be = s * 64 + lane;
            }
        }
        wave_argmax(bv, be, bp);
        if (lane == 0) {
            out_w[(size_t)row * K + k] = bp;
            out_ids[(size_t)row * K + k] = be;
            sum += bp;
        }
#pragma unroll
        for (int s = 0; s < kMaxSlots; ++s)
            if (be == s * 64 + lane) ch[s] = -__builtin_inff();
    }
    if (lane == 0) {
        for (int k = 0; k < K; ++k) {
            float w = out_w[(size_t)row * K + k];
            if (renorm) w = w / sum;
            if (rsf != 1.0f) w = w * rsf;
            out_w[(size_t)row * K + k] = w;
        }
    }
}

__global__ void map_ids_kernel(const int32_t* __restrict__ ids, int64_t n,
                               const int32_t* __restrict__ map, int E, int32_t* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t id = ids[i];
    int32_t c = id < 0 ? 0 : (id > E - 1 ? E - 1 : id);
    out[i] = id < 0 ? -1 : map[c];
}

// Expert-parallel dispatch pack (fixed capacity, no host synchronisation): for every destination
// rank r and every slot s = m*K+k, send_ids[r][s] = local expert id at r (or -1 if the slot is not
// routed to r), send_w[r][s] = routing weight, send_x[r][s] = hidden[m] (copied only for routed
// slots; unrouted rows are never read by the receiving engine).  Linear expert placement
// (expert_map_manager.py:62-79): the first `rem` ranks own base+1 experts.
__global__ __launch_bounds__(256) void ep_pack_kernel(const unsigned short* __restrict__ hidden,
                                                      const int32_t* __restrict__ ids,
                                                      const float* __restrict__ tw, int M, int K, int H,
                                                      int E, int ep, unsigned short* __restrict__ send_x,
                                                      int32_t* __restrict__ send_ids,
                                                      float* __restrict__ send_w) {
    const int s = blockIdx.x, r = blockIdx.y;
    const int n_slots = M * K;
    const int id = ids[s];
    const int base = E / ep, rem = E % ep, cut = rem * (base + 1);
    int owner = -1, first = 0;
    if (id >= 0 && id < E) {
        owner = id < cut ? id / (base + 1) : rem + (id - cut) / (base > 0 ? base : 1);
        first = owner * base + (owner < rem ? owner : rem);
    }
    const bool mine = owner == r;
    if (threadIdx.x == 0) {
        send_ids[(size_t)r * n_slots + s] = mine ? id - first : -1;
        send_w[(size_t)r * n_slots + s] = mine ? tw[s] : 0.0f;
    }
    if (!mine) return;
    const u32x4* src = (const u32x4*)(hidden + (size_t)(s / K) * H);
    u32x4* dst = (u32x4*)(send_x + ((size_t)r * n_slots + s) * H);
    for (int i = threadIdx.x; i < H / 8; i += 256) dst[i] = src[i];
}

}  // namespace lkm

using namespace lkm;

extern "C" int lkm_ep_pack(void* stream, const void* hidden, const int32_t* topk_ids,
                           const float* topk_weights, int32_t M, int32_t K, int32_t H,
                           int32_t num_experts, int32_t ep_size, void* send_x, int32_t* send_ids,
                           float* send_w) {
    LKM_REQUIRE(M >= 0 && K > 0 && H > 0 && H % 8 == 0 && num_experts > 0 && ep_size > 0, "ep_pack: bad sizes");
    if (M == 0) return LKM_OK;
    hipLaunchKernelGGL(ep_pack_kernel, dim3(M * K, ep_size), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)hidden, topk_ids, topk_weights, M, K, H, num_experts,
                       ep_size, (unsigned short*)send_x, send_ids, send_w);
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}

extern "C" int lkm_topk_softmax(void* stream, const void* logits, int32_t logits_dtype,
                                const float* bias, int32_t M, int32_t E, int32_t K,
                                int32_t scoring, int32_t renormalize, float routed_scaling,
                                float* out_weights, int32_t* out_ids) {
    LKM_REQUIRE(M >= 0 && E > 0 && K > 0 && K <= E, "topk_softmax: bad shape M=%d E=%d K=%d", M, E, K);
    LKM_REQUIRE(E <= kMaxSlots * 64, "topk_softmax: E=%d > %d unsupported", E, kMaxSlots * 64);
    LKM_REQUIRE(logits_dtype >= LKM_DT_F32 && logits_dtype <= LKM_DT_F16, "topk_softmax: bad dtype");
    LKM_REQUIRE(scoring == 0 || scoring == 1, "topk_softmax: scoring must be 0 (softmax) or 1 (sigmoid)");
    if (M == 0) return LKM_OK;
    hipLaunchKernelGGL(topk_softmax_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, (hipStream_t)stream,
                       logits, logits_dtype, bias, M, E, K, scoring, renormalize, routed_scaling,
                       out_weights, out_ids);
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}

extern "C" int lkm_grouped_topk(void* stream, const void* logits, int32_t logits_dtype,
                                const float* bias, int32_t M, int32_t E, int32_t K,
                                int32_t n_group, int32_t topk_group, int32_t scoring,
                                int32_t renormalize, float routed_scaling, float* out_weights,
                                int32_t* out_ids) {
    LKM_REQUIRE(M >= 0 && E > 0 && K > 0, "grouped_topk: bad shape");
    LKM_REQUIRE(E <= kMaxSlots * 64, "grouped_topk: E=%d > %d unsupported", E, kMaxSlots * 64);
    LKM_REQUIRE(n_group > 0 && n_group <= 64 && E % n_group == 0, "grouped_topk: n_group=%d must divide E=%d and be <= 64", n_group, E);
    LKM_REQUIRE(topk_group > 0 && topk_group <= n_group, "grouped_topk: bad topk_group=%d", topk_group);
    LKM_REQUIRE(K <= topk_group * (E / n_group), "grouped_topk: K=%d exceeds kept experts", K);
    LKM_REQUIRE(logits_dtype >= LKM_DT_F32 && logits_dtype <= LKM_DT_F16, "grouped_topk: bad dtype");
    LKM_REQUIRE(scoring == 0 || scoring == 1, "grouped_topk: scoring must be 0 or 1");
    if (M == 0) return LKM_OK;
    hipLaunchKernelGGL(grouped_topk_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, (hipStream_t)stream,
                       logits, logits_dtype, bias, M, E, K, n_group, topk_group, scoring,
                       renormalize, routed_scaling, out_weights, out_ids);
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}

extern "C" int lkm_map_expert_ids(void* stream, const int32_t* ids, int64_t n,
                                  const int32_t* expert_map, int32_t E, int32_t* out) {
    LKM_REQUIRE(n >= 0 && E > 0, "map_expert_ids: bad sizes");
    if (n == 0) return LKM_OK;
    hipLaunchKernelGGL(map_ids_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, ids, n, expert_map, E, out);
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}
