/* lkm_eplb.h -- placement maintenance for the HBM-resident expert tier: the part of
 * the C ABI that expert-parallel load balancing needs.  Extension of include/lkm.h (same library,
 * same error convention: 0 = LKM_OK, negative = LKM_E_*, text via lkm_last_error()).
 *
 * Reference interfaces replaced (paths relative to the reference tree, guqiong96/Lvllm):
 *   lkm_eplb_map_record   eplb_map_to_physical_and_record
 *                         (vllm/model_executor/layers/fused_moe/router/base_router.py:24-128, called from
 *                          BaseRouter._apply_eplb_mapping :204-223 right after top-k)
 *   lkm_expert_bytes / lkm_export_expert / lkm_import_expert
 *                         the per-parameter expert rows `w[src]` / `b[dst]` that
 *                         vllm/distributed/eplb/rebalance_execute.py:172-425 (move_to_buffer /
 *                         move_from_buffer) sends, receives and copies: here ONE packed image per expert
 *                         (the engine's pre-shuffled MFMA layout, all slabs of an expert back to back).
 */
#ifndef LKM_EPLB_H
#define LKM_EPLB_H

#include "lkm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Logical -> physical expert ids after routing, and load recording, in one pass over the M*top_k slots.
 *   out_ids[i] = (0 <= id < num_logical) ? log2phy[id * map_slots + hash(i / top_k) % max(cnt[id], 1)] : -1
 *   hash(t)    = (t * 2654435769) mod 2^32          (Knuth multiplicative hash of the TOKEN index: every
 *                                                     slot of a token picks the same replica rank)
 *   if (*record_enabled != 0) and (num_unpadded == NULL or i < *num_unpadded * top_k)
 *      and 0 <= out_ids[i] < load_size:   atomically load[out_ids[i]] += 1
 * All pointers are device memory; int32 throughout (ids, maps, counters).  `record_enabled` and
 * `num_unpadded` are device scalars so that a captured hipGraph keeps honouring them.  `load` may be NULL
 * (map only).  out_ids may alias topk_ids.  Asynchronous on `stream`; capturable. */
int lkm_eplb_map_record(void* stream, const int32_t* topk_ids, int64_t numel, int32_t top_k,
                        const int32_t* log2phy, const int32_t* logcnt, int32_t num_logical,
                        int32_t map_slots, int32_t* load, int32_t load_size,
                        const int32_t* record_enabled, const int32_t* num_unpadded, int32_t* out_ids);

/* Bytes of one expert's packed image (weights + scales + per-expert multipliers, each slab padded to
 * 16 bytes).  The same for every expert of the engine and for every engine built from the same
 * configuration, on any rank.  < 0 on error. */
int64_t lkm_expert_bytes(LkmHandle h);

/* Copy local expert `expert`'s slabs into `dst` (device memory, lkm_expert_bytes() bytes) / overwrite them
 * from `src`.  Device-to-device, asynchronous on `stream`, ordered with the engine's kernels on that
 * stream.  The image is opaque: valid only for engines of the same configuration (format, shapes, group
 * sizes); it is already in the kernels' operand layout, nothing is re-shuffled. */
int lkm_export_expert(LkmHandle h, void* stream, int32_t expert, void* dst);
int lkm_import_expert(LkmHandle h, void* stream, int32_t expert, const void* src);

#ifdef __cplusplus
}
#endif
#endif /* LKM_EPLB_H */
