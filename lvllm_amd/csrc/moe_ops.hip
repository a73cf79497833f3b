// moe_ops.hip -- the scatter / gather step of the path as stand-alone operators (SURVEY 8 a9, round-4 verdict "missing" 4):
//   lkm_moe_align_block_size   vllm/model_executor/layers/fused_moe/moe_align_block_size.py:11-103
//   lkm_moe_permute            vllm/model_executor/layers/fused_moe/moe_permute_unpermute.py:105-242
//   lkm_moe_unpermute          vllm/model_executor/layers/fused_moe/moe_permute_unpermute.py:245-283
// The engine itself never materialises these forms (its GEMMs gather through sorted_slot and its combine reads
// pos_of_slot); a GPU-resident layer's caller that wants the reference's operator outputs gets them from here.  All three
// sit on the engine's counting sort (dispatch.hip launch_sort: stable, so rows of an expert keep token order -- the order
// the reference's golden implementations produce, tests/kernels/moe/test_moe_align_block_size.py:96-172 and
// test_moe_permute_unpermute.py:37-123) plus one placement kernel each.  Integer outputs are exact; moe_unpermute sums in
// fp32 in slot order and rounds once.  Scratch comes from the caller (lkm_moe_ops_workspace_bytes): no allocation, no host
// synchronisation -- the operators are graph-capturable.
#include "lkm_kernels.h"

namespace lkm {

constexpr int kOpsMaxKeys = 512;      // dispatch.hip kMaxLocalExperts: the sort's expert range

struct OpsWs {
    int32_t *keys, *counts, *offsets, *sorted_slot, *pos_of_slot, *active, *meta, *hist;
    size_t hist_cap;
};
static size_t ops_ws_ints(int n_slots, int E) {
    const size_t hist = n_slots > 4096 ? ((size_t)n_slots / 1024 + 1) * (size_t)E : 0;
    return (size_t)n_slots * 3 + (size_t)E * 3 + 1 + kMetaInts + hist + 16;
}
static OpsWs ops_ws_carve(void* ws, int n_slots, int E) {
    OpsWs w;
    int32_t* p = (int32_t*)ws;
    w.keys = p;            p += n_slots;
    w.sorted_slot = p;     p += n_slots;
    w.pos_of_slot = p;     p += n_slots;
    w.counts = p;          p += E;
    w.offsets = p;         p += E + 1;
    w.active = p;          p += E;
    w.meta = p;            p += kMetaInts;
    w.hist_cap = n_slots > 4096 ? ((size_t)n_slots / 1024 + 1) * (size_t)E : 0;
    w.hist = w.hist_cap ? p : nullptr;
    return w;
}

// key of a slot for the sort.  No expert_map: the id itself.  With one, align (compact == 0): the MAPPED (local) id, ids
// whose map entry is -1 dropped -- the reference counts and ranks by get_local_expert_id (moe_align_sum_kernels.cu:86-100),
// so its blocks come in local-id order for ANY map (permuted / EPLB-style placements too), and expert_ids holds the local
// id; permute (compact > 0 = n_local): local experts by their local id first, then the others by n_local +
// global id (test_moe_permute_unpermute.py:49-55: "topk_ids + n_expert" -- any order-preserving offset sorts the same).
__global__ __launch_bounds__(256) void ops_keys_kernel(const int32_t* __restrict__ ids, int n, int n_expert,
                                                       const int32_t* __restrict__ expert_map, int n_local, int n_keys,
                                                       int32_t* __restrict__ keys) {
    // permute with a map: the non-local experts sort behind the local ones BY GLOBAL ID.  Their keys are n_local + (rank
    // among the non-local global ids), not n_local + id: the key range stays n_expert (<= 512, the sort's range) for a
    // 512-expert model under expert parallelism too.  Every block rebuilds the <= 512-entry rank table in LDS.
    __shared__ int s_rank[kOpsMaxKeys];
    const int tid = threadIdx.x;
    if (expert_map && n_local > 0) {
        for (int j = tid; j < kOpsMaxKeys; j += 256) s_rank[j] = (j < n_expert && expert_map[j] < 0) ? 1 : 0;
        __syncthreads();
        for (int d = 1; d < kOpsMaxKeys; d <<= 1) {          // inclusive scan (Hillis-Steele), two entries per thread
            int v[2];
            for (int q = 0; q < 2; ++q) {
                const int j = tid + q * 256;
                v[q] = s_rank[j] + (j >= d ? s_rank[j - d] : 0);
            }
            __syncthreads();
            for (int q = 0; q < 2; ++q) s_rank[tid + q * 256] = v[q];
            __syncthreads();
        }
    }
    const int i = blockIdx.x * 256 + tid;
    if (i >= n) return;
    const int id = ids[i];
    int key = -1;
    if (id >= 0 && id < n_expert) {
        if (!expert_map) key = id;
        else {
            const int l = expert_map[id];
            if (n_local > 0) {
                key = (l >= 0 && l < n_local) ? l : n_local + s_rank[id] - 1;     // (inclusive rank of a non-local id >= 1)
                if (key >= n_keys) key = n_keys - 1;                              // malformed map: stay inside the sort's range
            } else {
                key = (l >= 0 && l < n_expert) ? l : -1;
            }
        }
    }
    keys[i] = key;
}

// ---- moe_align_block_size: block e places expert e's rows at the block-padded offset; the last block fills the tails
__global__ __launch_bounds__(256) void align_place_kernel(const int32_t* __restrict__ counts, const int32_t* __restrict__ offsets,
                                                          const int32_t* __restrict__ sorted_slot, int E, int block_size,
                                                          int n_slots, const int32_t* __restrict__ expert_map,
                                                          int32_t* __restrict__ sorted_ids, int sorted_cap,
                                                          int32_t* __restrict__ expert_ids, int blocks_cap,
                                                          int32_t* __restrict__ num_post_pad) {
    __shared__ int s_red[256];
    const int e = blockIdx.x, tid = threadIdx.x;
    // padded rows of the experts before mine (e == E: of all of them)
    int part = 0;
    for (int j = tid; j < e; j += 256) part += (counts[j] + block_size - 1) / block_size * block_size;
    s_red[tid] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) s_red[tid] += s_red[tid + s];
        __syncthreads();
    }
    const int poff = s_red[0];
    if (e == E) {     // tails: unused ids = n_slots (the padding value), unused blocks = -1, and the padded total
        for (int i = poff + tid; i < sorted_cap; i += 256) sorted_ids[i] = n_slots;
        for (int b = poff / block_size + tid; b < blocks_cap; b += 256) expert_ids[b] = -1;
        if (tid == 0) num_post_pad[0] = poff;
        return;
    }
    const int cnt = counts[e], off = offsets[e];
    const int padded = (cnt + block_size - 1) / block_size * block_size;
    for (int i = tid; i < padded; i += 256)
        if (poff + i < sorted_cap) sorted_ids[poff + i] = i < cnt ? sorted_slot[off + i] : n_slots;
    const int eid = e;      // keys are local ids when an expert_map is given (ops_keys_kernel)
    for (int b = tid; b < padded / block_size; b += 256)
        if (poff / block_size + b < blocks_cap) expert_ids[poff / block_size + b] = eid;
}

// ---- moe_permute: index maps + the gathered rows of the local experts
__global__ __launch_bounds__(256) void permute_index_kernel(const int32_t* __restrict__ offsets, const int32_t* __restrict__ sorted_slot,
                                                            const int32_t* __restrict__ pos_of_slot, int n, int n_local,
                                                            int64_t* __restrict__ first_token_offset,
                                                            int32_t* __restrict__ inv_permuted_idx, int32_t* __restrict__ permuted_idx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i <= n_local) first_token_offset[i] = offsets[i];
    if (i >= n) return;
    const int valid = offsets[n_local];
    inv_permuted_idx[i] = pos_of_slot[i];
    permuted_idx[i] = i < valid ? sorted_slot[i] : n;
}
__global__ __launch_bounds__(256) void permute_rows_kernel(const u32x4* __restrict__ src, const int32_t* __restrict__ offsets,
                                                           const int32_t* __restrict__ sorted_slot, int n_local, int topk,
                                                           int vec_per_row, u32x4* __restrict__ dst) {
    const int row = blockIdx.x;
    if (row >= offsets[n_local]) return;        // (rows of non-local experts are never read: test ...:187-192 compares valid rows)
    const u32x4* s = src + (size_t)(sorted_slot[row] / topk) * vec_per_row;
    u32x4* d = dst + (size_t)row * vec_per_row;
    for (int v = threadIdx.x; v < vec_per_row; v += 256) d[v] = s[v];       // (a token row is read top_k times: keep it cached)
}

// ---- moe_unpermute: out[t] = T(sum_k w[t][k] * rows[inv[t][k]]) over the valid rows, fp32, slot order
template <int DT>
__device__ __forceinline__ float ops_to_f32(unsigned short h) {
    if (DT == LKM_DT_BF16) return __uint_as_float((unsigned)h << 16);
    return (float)__builtin_bit_cast(_Float16, h);
}
template <int DT>
__device__ __forceinline__ unsigned short ops_from_f32(float f) {
    if (DT == LKM_DT_BF16) {
        unsigned u = __float_as_uint(f);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);      // NaN stays NaN
        u += 0x7fffu + ((u >> 16) & 1u);
        return (unsigned short)(u >> 16);
    }
    return __builtin_bit_cast(unsigned short, (_Float16)f);
}
template <int DT>
__global__ __launch_bounds__(256) void unpermute_kernel(const unsigned short* __restrict__ rows, const float* __restrict__ tw,
                                                        const int32_t* __restrict__ inv, const int64_t* __restrict__ first_token_offset,
                                                        int n_local, int topk, int H, unsigned short* __restrict__ out) {
    const int t = blockIdx.x;
    const long long valid = first_token_offset ? first_token_offset[n_local] : 0x7fffffffLL;
    for (int c0 = threadIdx.x * 8; c0 < H; c0 += 256 * 8) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < topk; ++k) {
            const int r = inv[(size_t)t * topk + k];
            if (r < 0 || r >= valid) continue;
            const float w = tw[(size_t)t * topk + k];
            const u32x4 v = *(const u32x4*)(rows + (size_t)r * H + c0);
            const unsigned d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[2 * j] += w * ops_to_f32<DT>((unsigned short)(d[j] & 0xffffu));
                acc[2 * j + 1] += w * ops_to_f32<DT>((unsigned short)(d[j] >> 16));
            }
        }
        u32x4 o;
        o.x = (unsigned)ops_from_f32<DT>(acc[0]) | ((unsigned)ops_from_f32<DT>(acc[1]) << 16);
        o.y = (unsigned)ops_from_f32<DT>(acc[2]) | ((unsigned)ops_from_f32<DT>(acc[3]) << 16);
        o.z = (unsigned)ops_from_f32<DT>(acc[4]) | ((unsigned)ops_from_f32<DT>(acc[5]) << 16);
        o.w = (unsigned)ops_from_f32<DT>(acc[6]) | ((unsigned)ops_from_f32<DT>(acc[7]) << 16);
        *(u32x4*)(out + (size_t)t * H + c0) = o;
    }
}

static int ops_sort(hipStream_t st, const OpsWs& w, int n, int E) {
    return launch_sort(st, w.keys, 1, 1, 0, n, E, w.counts, w.offsets, w.sorted_slot, w.pos_of_slot, w.active, w.meta, 0, 0,
                       nullptr, nullptr, w.hist, w.hist_cap);
}

}  // namespace lkm

using namespace lkm;

extern "C" int64_t lkm_moe_ops_workspace_bytes(int32_t n_slots, int32_t n_keys) {
    if (n_slots < 0 || n_keys <= 0) return 0;
    return (int64_t)ops_ws_ints(n_slots, n_keys) * 4;
}

extern "C" int lkm_moe_align_block_size(void* stream, const int32_t* topk_ids, int32_t n_slots, int32_t num_experts,
                                        int32_t block_size, const int32_t* expert_map, int32_t* sorted_ids,
                                        int32_t sorted_cap, int32_t* expert_ids, int32_t blocks_cap,
                                        int32_t* num_tokens_post_pad, void* workspace) {
    LKM_REQUIRE(n_slots >= 0 && block_size > 0 && sorted_ids && expert_ids && num_tokens_post_pad && workspace, "moe_align_block_size: bad arguments");
    LKM_REQUIRE(num_experts > 0 && num_experts <= kOpsMaxKeys, "moe_align_block_size: num_experts=%d out of range (1..%d)", num_experts, kOpsMaxKeys);
    LKM_REQUIRE(n_slots == 0 || topk_ids, "moe_align_block_size: null ids");
    hipStream_t st = (hipStream_t)stream;
    const OpsWs w = ops_ws_carve(workspace, n_slots, num_experts);
    if (n_slots > 0) {
        hipLaunchKernelGGL(ops_keys_kernel, dim3((unsigned)ceil_div(n_slots, 256)), dim3(256), 0, st, topk_ids, n_slots, num_experts,
                           expert_map, 0, num_experts, w.keys);
        int rc = ops_sort(st, w, n_slots, num_experts);
        if (rc != LKM_OK) return rc;
    } else {
        LKM_HIP_CHECK(hipMemsetAsync(w.counts, 0, sizeof(int32_t) * (2 * (size_t)num_experts + 1), st));   // counts + offsets
    }
    hipLaunchKernelGGL(align_place_kernel, dim3((unsigned)num_experts + 1), dim3(256), 0, st, w.counts, w.offsets, w.sorted_slot,
                       num_experts, block_size, n_slots, expert_map, sorted_ids, sorted_cap, expert_ids, blocks_cap, num_tokens_post_pad);
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}

extern "C" int lkm_moe_permute(void* stream, const void* hidden, int32_t row_bytes, int32_t n_token, const int32_t* topk_ids,
                               int32_t topk, const int32_t* expert_map, int32_t n_expert, int32_t n_local_expert,
                               void* permuted_hidden, int64_t* expert_first_token_offset, int32_t* inv_permuted_idx,
                               int32_t* permuted_idx, void* workspace) {
    LKM_REQUIRE(n_token >= 0 && topk > 0 && n_expert > 0 && n_local_expert > 0 && n_local_expert <= n_expert, "moe_permute: bad sizes");
    LKM_REQUIRE(row_bytes > 0 && row_bytes % 16 == 0, "moe_permute: hidden rows must be a multiple of 16 bytes (got %d)", row_bytes);   // (the reference's assert, :136-138)
    LKM_REQUIRE(expert_first_token_offset && inv_permuted_idx && permuted_idx && permuted_hidden && workspace, "moe_permute: null output");
    LKM_REQUIRE(expert_map || n_local_expert == n_expert, "moe_permute: n_local_expert < n_expert needs an expert_map");
    // keys: local ids, then the non-local experts ranked by global id -- n_expert keys for a well-formed map (n_local of its
    // entries >= 0); the workspace contract stays n_local_expert + n_expert (lkm.h), capped at the sort's range
    LKM_REQUIRE(n_expert <= kOpsMaxKeys, "moe_permute: n_expert=%d exceeds the sort's %d keys", n_expert, kOpsMaxKeys);
    const int n_keys = expert_map ? (n_local_expert + n_expert <= kOpsMaxKeys ? n_local_expert + n_expert : kOpsMaxKeys) : n_expert;
    LKM_REQUIRE(((uintptr_t)hidden & 15) == 0 && ((uintptr_t)permuted_hidden & 15) == 0, "moe_permute: rows must be 16-byte aligned");
    const long long n = (long long)n_token * topk;
    LKM_REQUIRE(n < (1LL << 31), "moe_permute: %lld slots", n);
    hipStream_t st = (hipStream_t)stream;
    const OpsWs w = ops_ws_carve(workspace, (int)n, n_keys);
    if (n == 0) {
        LKM_HIP_CHECK(hipMemsetAsync(expert_first_token_offset, 0, sizeof(int64_t) * ((size_t)n_local_expert + 1), st));
        return LKM_OK;
    }
    LKM_REQUIRE(hidden && topk_ids, "moe_permute: null input");
    hipLaunchKernelGGL(ops_keys_kernel, dim3((unsigned)ceil_div((int)n, 256)), dim3(256), 0, st, topk_ids, (int)n, n_expert, expert_map,
                       expert_map ? n_local_expert : 0, n_keys, w.keys);
    int rc = ops_sort(st, w, (int)n, n_keys);
    if (rc != LKM_OK) return rc;
    const int cover = (int)n > n_local_expert + 1 ? (int)n : n_local_expert + 1;
    hipLaunchKernelGGL(permute_index_kernel, dim3((unsigned)ceil_div(cover, 256)), dim3(256), 0, st, w.offsets, w.sorted_slot,
                       w.pos_of_slot, (int)n, n_local_expert, expert_first_token_offset, inv_permuted_idx, permuted_idx);
    hipLaunchKernelGGL(permute_rows_kernel, dim3((unsigned)n), dim3(256), 0, st, (const u32x4*)hidden, w.offsets, w.sorted_slot,
                       n_local_expert, topk, row_bytes / 16, (u32x4*)permuted_hidden);
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}

extern "C" int lkm_moe_unpermute(void* stream, const void* permuted_hidden, int32_t dtype, const float* topk_weights,
                                 const int32_t* inv_permuted_idx, const int64_t* expert_first_token_offset,
                                 int32_t n_local_expert, int32_t n_token, int32_t topk, int32_t n_hidden, void* out) {
    LKM_REQUIRE(n_token >= 0 && topk > 0 && n_hidden > 0, "moe_unpermute: bad sizes");
    LKM_REQUIRE(dtype == LKM_DT_BF16 || dtype == LKM_DT_F16, "moe_unpermute: 16-bit rows only (dtype %d)", dtype);
    LKM_REQUIRE(n_hidden % 8 == 0, "moe_unpermute: hidden rows must be a multiple of 16 bytes");                     // (:266-268)
    if (n_token == 0) return LKM_OK;
    LKM_REQUIRE(permuted_hidden && topk_weights && inv_permuted_idx && out, "moe_unpermute: null pointer");
    LKM_REQUIRE(!expert_first_token_offset || n_local_expert > 0, "moe_unpermute: n_local_expert must accompany the offsets");
    LKM_REQUIRE(((uintptr_t)permuted_hidden & 15) == 0 && ((uintptr_t)out & 15) == 0, "moe_unpermute: rows must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == LKM_DT_BF16)
        hipLaunchKernelGGL(unpermute_kernel<LKM_DT_BF16>, dim3((unsigned)n_token), dim3(256), 0, st, (const unsigned short*)permuted_hidden,
                           topk_weights, inv_permuted_idx, expert_first_token_offset, n_local_expert, topk, n_hidden, (unsigned short*)out);
    else
        hipLaunchKernelGGL(unpermute_kernel<LKM_DT_F16>, dim3((unsigned)n_token), dim3(256), 0, st, (const unsigned short*)permuted_hidden,
                           topk_weights, inv_permuted_idx, expert_first_token_offset, n_local_expert, topk, n_hidden, (unsigned short*)out);
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}
