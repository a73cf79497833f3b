// CPU emulation build of lvllm_amd/csrc/eplb_kernel.inc (the SAME source the product compiles with hipcc), with the
// variant selection of lkm_eplb_map_record (eplb.hip) restated.  TEST INFRASTRUCTURE, see hip_cpu_emu.h.
#include "hip_cpu_emu.h"

#include "../../lvllm_amd/csrc/eplb_kernel.inc"

extern "C" int emu_eplb_map_record(const int32_t* ids, int64_t numel, int32_t top_k, const int32_t* log2phy,
                                   const int32_t* logcnt, int32_t num_logical, int32_t map_slots, int32_t* load,
                                   int32_t load_size, const int32_t* record_enabled, const int32_t* num_unpadded,
                                   int32_t* out, int32_t force_variant /* -1 auto, 0 global atomics, 1 LDS histogram */) {
    using namespace lkm;
    if (numel <= 0) return 0;
    const unsigned grid = (unsigned)((numel + kEplbBlock - 1) / kEplbBlock);
    bool hist = load != nullptr && load_size <= kEplbHistMax;
    if (force_variant >= 0) hist = force_variant == 1;
    if (hist)
        emu_launch(eplb_map_record_kernel<true>, grid, kEplbBlock, ids, numel, (int)top_k, log2phy, logcnt, (int)num_logical,
                   (int)map_slots, load, (int)load_size, record_enabled, num_unpadded, out);
    else
        emu_launch(eplb_map_record_kernel<false>, grid, kEplbBlock, ids, numel, (int)top_k, log2phy, logcnt, (int)num_logical,
                   (int)map_slots, load, (int)load_size, record_enabled, num_unpadded, out);
    return 0;
}
