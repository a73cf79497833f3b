// eplb.hip -- logical -> physical expert ids + expert-load recording for gfx950 (include/lkm_eplb.h).
//
// What it computes follows the reference's kernel for this step
// (vllm/model_executor/layers/fused_moe/router/base_router.py:24-97): per routed slot i of token t = i / top_k
//   replica = ((t * 2654435769) mod 2^32) mod max(replica_count[id], 1)
//   phys    = logical_to_physical[id][replica]          (-1 for ids outside [0, num_logical))
//   load[phys] += 1   when recording is on, the slot belongs to an unpadded token and 0 <= phys < load_size
// Integer work: the tests require bit-exact ids and counters against a CPU restatement.
// How: one lane per slot, 256 lanes per workgroup; the load counters of a workgroup are accumulated in an
// LDS histogram (ds atomics) and flushed with ONE global atomic per touched expert, so a hot expert costs
// one global atomic per workgroup instead of one per routed row.  Integer adds commute: any order gives
// the same counters.
#include "lkm_common.h"
#include "../../include/lkm_eplb.h"

namespace lkm {

constexpr int kEplbBlock = 256;
constexpr int kEplbHistMax = 2048;   // physical experts per layer that fit the LDS histogram (8 KiB)

template <bool LDS_HIST>
__global__ __launch_bounds__(kEplbBlock) void eplb_map_record_kernel(
    const int32_t* ids, int64_t numel, int top_k, const int32_t* __restrict__ log2phy,
    const int32_t* __restrict__ logcnt, int num_logical, int map_slots, int32_t* load, int load_size,
    const int32_t* __restrict__ record_enabled, const int32_t* __restrict__ num_unpadded, int32_t* out) {
    __shared__ int32_t hist[LDS_HIST ? kEplbHistMax : 1];
    const int tid = threadIdx.x;
    // workgroup-uniform: the switch is a device scalar so that a captured graph keeps honouring it
    const bool rec = load != nullptr && (record_enabled == nullptr || *record_enabled != 0);
    if (LDS_HIST && rec) {
        for (int j = tid; j < load_size; j += kEplbBlock) hist[j] = 0;
        __syncthreads();
    }
    const int64_t i = (int64_t)blockIdx.x * kEplbBlock + tid;
    if (i < numel) {
        const int32_t id = ids[i];
        int32_t phys = -1;
        if (id >= 0 && id < num_logical) {
            int32_t cnt = logcnt[id];
            cnt = cnt < 1 ? 1 : (cnt > map_slots ? map_slots : cnt);   // > map_slots only for inconsistent maps: stay in bounds
            const uint32_t hashed = (uint32_t)(i / top_k) * 2654435769u;   // low 32 bits of t * floor(2^32 / phi)
            phys = log2phy[(int64_t)id * map_slots + (int32_t)(hashed % (uint32_t)cnt)];
        }
        out[i] = phys;
        if (rec && phys >= 0 && phys < load_size &&
            (num_unpadded == nullptr || i < (int64_t)(*num_unpadded) * top_k)) {
            if (LDS_HIST) atomicAdd(&hist[phys], 1);
            else atomicAdd(&load[phys], 1);
        }
    }
    if (LDS_HIST && rec) {
        __syncthreads();
        for (int j = tid; j < load_size; j += kEplbBlock) {
            const int32_t v = hist[j];
            if (v != 0) atomicAdd(&load[j], v);
        }
    }
}

}  // namespace lkm

using namespace lkm;

extern "C" int lkm_eplb_map_record(void* stream, const int32_t* topk_ids, int64_t numel, int32_t top_k,
                                   const int32_t* log2phy, const int32_t* logcnt, int32_t num_logical,
                                   int32_t map_slots, int32_t* load, int32_t load_size,
                                   const int32_t* record_enabled, const int32_t* num_unpadded,
                                   int32_t* out_ids) {
    LKM_REQUIRE(numel >= 0 && top_k > 0, "eplb_map_record: bad sizes (numel=%lld top_k=%d)", (long long)numel, top_k);
    if (numel == 0) return LKM_OK;
    LKM_REQUIRE(topk_ids && out_ids && log2phy && logcnt, "eplb_map_record: null pointer");
    LKM_REQUIRE(num_logical > 0 && map_slots > 0, "eplb_map_record: bad map shape (%d logical x %d slots)", num_logical, map_slots);
    LKM_REQUIRE(load == nullptr || load_size > 0, "eplb_map_record: load counters given with load_size=%d", load_size);
    const dim3 grid((unsigned)((numel + kEplbBlock - 1) / kEplbBlock)), block(kEplbBlock);
    if (load != nullptr && load_size <= kEplbHistMax)
        hipLaunchKernelGGL(eplb_map_record_kernel<true>, grid, block, 0, (hipStream_t)stream, topk_ids, numel,
                           top_k, log2phy, logcnt, num_logical, map_slots, load, load_size, record_enabled,
                           num_unpadded, out_ids);
    else
        hipLaunchKernelGGL(eplb_map_record_kernel<false>, grid, block, 0, (hipStream_t)stream, topk_ids, numel,
                           top_k, log2phy, logcnt, num_logical, map_slots, load, load_size, record_enabled,
                           num_unpadded, out_ids);
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}
