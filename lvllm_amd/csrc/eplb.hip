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

#include "eplb_kernel.inc"

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
