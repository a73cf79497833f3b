// ep.hip -- expert-parallel exchange kernels: the pack in front of the dispatch all-to-all and the sum
// behind the return all-to-all (include/lkm.h: lkm_ep_pack_tokens / lkm_ep_combine).
//
// Stands where the reference's all-to-all prepare/finalize backends stand
// (vllm/model_executor/layers/fused_moe/modular_kernel.py:257-418, prepare_finalize/*, device side
// vllm/distributed/device_communicators/all2all.py:101-150): those move one row per (token, expert slot).
// Here a token travels to a rank at most once, as one record [activations | top_k ids | top_k weights],
// the owner's engine forms the weighted sum over ITS experts of that token (the grouped GEMMs' own top-k
// combine) and ONE row per (token, rank) comes back: with group-limited routing (DeepSeek-V3: 8 experts in
// <= 4 groups = ranks) that is <= half the rows of the slot-granular form, and the worst case per
// destination is exactly num_tokens records -- a fixed capacity without any overflow, i.e. equal-split
// collectives with no size exchange and no host synchronisation, capturable in a hipGraph.
// Byte work, HBM/latency-bound: 16-byte coalesced copies, one workgroup per (token chunk, destination).
#include "lkm_kernels.h"

namespace lkm {

constexpr int kEpThreads = 256;
constexpr int kEpTokPerWg = 16;      // tokens copied by one workgroup (4 for decode-sized steps)
constexpr int kEpMaxTokens = 8192;   // LDS: one int per token

__host__ __device__ inline int64_t ep_row_bytes(int H, int K) { return ((int64_t)H * 2 + (int64_t)K * 8 + 15) / 16 * 16; }

// rank that owns global expert `id` under linear placement (expert_map_manager.py:62-79: the first
// E % ep ranks hold one expert more) and that rank's first expert
__device__ __forceinline__ int ep_owner(int id, int base, int rem, int cut, int* first) {
    const int o = id < cut ? id / (base + 1) : rem + (id - cut) / (base > 0 ? base : 1);
    *first = o * base + (o < rem ? o : rem);
    return o;
}

// grid = (token chunks, ep).  Every workgroup of destination p first ranks ALL tokens that have an expert on
// p (ascending token index -> deterministic record order; M*K ids, a few KB), then copies its own chunk.
__global__ __launch_bounds__(kEpThreads) void ep_pack_tokens_kernel(
    const unsigned short* __restrict__ hidden, const int32_t* __restrict__ ids, const float* __restrict__ tw,
    int M, int K, int H, int E, int ep, int cap, int global_ids, int tok_per_wg, unsigned char* __restrict__ send,
    int32_t* __restrict__ slot_of, int32_t* __restrict__ overflow) {
    __shared__ int32_t s_slot[kEpMaxTokens];
    __shared__ int32_t s_wsum[kEpThreads / 64];
    const int p = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int base = E / ep, rem = E % ep, cut = rem * (base + 1);
    const int64_t rowb = ep_row_bytes(H, K);
    int carry = 0;
    for (int m0 = 0; m0 < M; m0 += kEpThreads) {
        const int m = m0 + tid;
        bool f = false;
        if (m < M) {
            for (int k = 0; k < K; ++k) {
                const int id = ids[(size_t)m * K + k];
                int first;
                if (id >= 0 && id < E && ep_owner(id, base, rem, cut, &first) == p) f = true;
            }
        }
        const unsigned long long b = __ballot(f);
        const int before = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) s_wsum[wv] = __popcll(b);
        __syncthreads();
        int wbase = 0, tot = 0;
        for (int w = 0; w < kEpThreads / 64; ++w) {
            if (w < wv) wbase += s_wsum[w];
            tot += s_wsum[w];
        }
        if (m < M) s_slot[m] = f ? carry + wbase + before : -1;
        carry += tot;
        __syncthreads();
    }
    const int n_rec = carry < cap ? carry : cap;
    unsigned char* blockp = send + (size_t)p * cap * rowb;
    if (blockIdx.x == 0) {
        // unused record slots: ids = -1 (the receiver's scatter skips them; their activations are never read)
        for (int i = tid; i < (cap - n_rec) * K; i += kEpThreads) {
            const int c = n_rec + i / K, k = i % K;
            ((int32_t*)(blockp + (size_t)c * rowb + (size_t)H * 2))[k] = -1;
        }
        if (tid == 0 && carry > cap) atomicAdd(overflow, carry - cap);
    }
    // my chunk of tokens: one wavefront per token, 16-byte copies, four of them in flight per lane (a copy loop that
    // waits for every load before its store took 15 us for 32 tokens of 8 KB: decode-sized steps now get one token per
    // wavefront, tok_per_wg = 4, and spread over M/4 workgroups)
    const int t0 = blockIdx.x * tok_per_wg;
    for (int t = wv; t < tok_per_wg; t += kEpThreads / 64) {
        const int m = t0 + t;
        if (m >= M) break;
        int c = s_slot[m];
        if (c >= cap) c = -1;
        if (lane == 0) slot_of[(size_t)p * M + m] = c;
        if (c < 0) continue;
        unsigned char* rec = blockp + (size_t)c * rowb;
        const u32x4* src = (const u32x4*)(hidden + (size_t)m * H);
        u32x4* dst = (u32x4*)rec;
        const int nv = H / 8;
        for (int i = lane; i < nv; i += 256) {
            u32x4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (i + q * 64 < nv) v[q] = src[i + q * 64];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (i + q * 64 < nv) dst[i + q * 64] = v[q];
        }
        int32_t* rid = (int32_t*)(rec + (size_t)H * 2);
        float* rw = (float*)(rec + (size_t)H * 2 + (size_t)K * 4);
        for (int k = lane; k < K; k += 64) {
            const int id = ids[(size_t)m * K + k];
            int first = 0, out = -1;
            if (id >= 0 && id < E && ep_owner(id, base, rem, cut, &first) == p) out = global_ids ? id : id - first;
            rid[k] = out;
            rw[k] = tw[(size_t)m * K + k];
        }
    }
}

// out[m][h] = sum_p back[p][slot_of[p][m]][h], fp32, p ascending
template <typename InT, typename OutT>
__global__ __launch_bounds__(256) void ep_combine_kernel(const InT* __restrict__ back,
                                                         const int32_t* __restrict__ slot_of, int M, int H,
                                                         int ep, int cap, OutT* __restrict__ out) {
#pragma clang fp contract(off)
    const int m = blockIdx.y;
    const int h = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (h >= H) return;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < ep; ++p) {
        const int c = slot_of[(size_t)p * M + m];
        if (c < 0) continue;
        acc += load4<InT>(back + ((size_t)p * cap + c) * H + h);
    }
    store4<OutT>(out + (size_t)m * H + h, acc);
}

template <typename InT>
static void launch_ep_combine_in(hipStream_t st, const void* back, const int32_t* slot_of, int M, int H, int ep,
                                 int cap, void* out, int out_dt) {
    dim3 grid(ceil_div(H, 1024), M), block(256);
    if (out_dt == LKM_DT_F32)
        hipLaunchKernelGGL((ep_combine_kernel<InT, float>), grid, block, 0, st, (const InT*)back, slot_of, M, H, ep,
                           cap, (float*)out);
    else if (out_dt == LKM_DT_BF16)
        hipLaunchKernelGGL((ep_combine_kernel<InT, bf16_out>), grid, block, 0, st, (const InT*)back, slot_of, M, H,
                           ep, cap, (bf16_out*)out);
    else
        hipLaunchKernelGGL((ep_combine_kernel<InT, f16_out>), grid, block, 0, st, (const InT*)back, slot_of, M, H,
                           ep, cap, (f16_out*)out);
}

}  // namespace lkm

using namespace lkm;

extern "C" int64_t lkm_ep_row_bytes(int32_t H, int32_t K) { return ep_row_bytes(H, K); }

extern "C" int lkm_ep_pack_tokens(void* stream, const void* hidden, const int32_t* topk_ids,
                                  const float* topk_weights, int32_t M, int32_t K, int32_t H,
                                  int32_t num_experts, int32_t ep_size, int32_t capacity, int32_t global_ids,
                                  void* send, int32_t* slot_of, int32_t* overflow) {
    LKM_REQUIRE(M >= 0 && K > 0 && H > 0 && H % 8 == 0 && num_experts > 0 && ep_size > 0, "ep_pack_tokens: bad sizes");
    LKM_REQUIRE(M <= kEpMaxTokens, "ep_pack_tokens: %d tokens > %d (use the ragged exchange for prefill sizes)", M, kEpMaxTokens);
    LKM_REQUIRE(capacity > 0, "ep_pack_tokens: capacity must be > 0");
    LKM_REQUIRE(send && slot_of && overflow && (M == 0 || (hidden && topk_ids && topk_weights)), "ep_pack_tokens: null pointer");
    // M == 0 (a rank without tokens this step) still has to mark its record slots empty
    // every workgroup ranks all tokens first (M*K ids): few tokens per workgroup while that is cheap
    const int tpw = M <= 256 ? 4 : kEpTokPerWg;
    hipLaunchKernelGGL(ep_pack_tokens_kernel, dim3(M > 0 ? ceil_div(M, tpw) : 1, ep_size), dim3(kEpThreads), 0,
                       (hipStream_t)stream, (const unsigned short*)hidden, topk_ids, topk_weights, M, K, H,
                       num_experts, ep_size, capacity, global_ids, tpw, (unsigned char*)send, slot_of, overflow);
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}

extern "C" int lkm_ep_combine(void* stream, const void* back, int32_t back_dtype, const int32_t* slot_of,
                              int32_t M, int32_t H, int32_t ep_size, int32_t capacity, void* out,
                              int32_t out_dtype) {
    LKM_REQUIRE(M >= 0 && H > 0 && H % 4 == 0 && ep_size > 0 && capacity > 0, "ep_combine: bad sizes");
    LKM_REQUIRE(back_dtype >= LKM_DT_F32 && back_dtype <= LKM_DT_F16 && out_dtype >= LKM_DT_F32 && out_dtype <= LKM_DT_F16, "ep_combine: bad dtype");
    if (M == 0) return LKM_OK;
    LKM_REQUIRE(back && slot_of && out, "ep_combine: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (back_dtype == LKM_DT_F32) launch_ep_combine_in<float>(st, back, slot_of, M, H, ep_size, capacity, out, out_dtype);
    else if (back_dtype == LKM_DT_BF16) launch_ep_combine_in<bf16_out>(st, back, slot_of, M, H, ep_size, capacity, out, out_dtype);
    else launch_ep_combine_in<f16_out>(st, back, slot_of, M, H, ep_size, capacity, out, out_dtype);
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}
