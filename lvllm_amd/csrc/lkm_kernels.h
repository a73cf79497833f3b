// lkm_kernels.h -- parameter blocks and host launchers shared between the translation units of
// liblkm.so (kernels live in routing.hip / dispatch.hip / repack.hip / gemm_skinny.hip).
#pragma once
#include "lkm_common.h"
#include "routing_dev.h"

namespace lkm {

// ---- lkm_api.hip: which GEMM kernels a step launched (lkm_last_kernels: bench.py's roofline.kernel and the FETCH
// passes of tools/update_hbm_traffic.py name the kernel from the launch, not from a literal).  Every GEMM launcher goes
// through LKM_LAUNCH_GEMM; the engine reads the host-side stubs noted between its gemm1 / gemm2 marks.
void note_gemm_launch(const void* host_fn);
#define LKM_LAUNCH_GEMM(kern, grid, block, lds, st, ...)            \
    do {                                                            \
        ::lkm::note_gemm_launch((const void*)(kern));               \
        hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__); \
    } while (0)

// ---- repack.hip
struct RepackDims {
    int E, n_half, halves, interleaved;  // source rows N = n_half*halves
    int K;                               // source K (elements)
    int T_half, U;                       // dest tiles per half, units
    int a8;                              // fp8 only: k mapping for fp8 activations (W8A8)
};
int launch_repack_w(hipStream_t st, int wf, const void* src, void* dst, const RepackDims& d);
int launch_repack_s_int4(hipStream_t st, const void* src, void* dst, const RepackDims& d, int group,
                         int spu);
int launch_repack_s_int4ps(hipStream_t st, const void* src, void* dst, const RepackDims& d, int group, int adt);
int launch_repack_s_int4zp(hipStream_t st, const void* src, void* dst, const RepackDims& d, int group, int spu, int which, int adt);
int launch_wna16_expand(hipStream_t st, const void* q, const void* scales, const void* zp, void* out, int64_t rows,
                        int K, int group, int bits, int adt);
int launch_repack_s_fp8(hipStream_t st, const void* src, void* dst, const RepackDims& d, int gN,
                        int gK);
int launch_repack_s_fp4(hipStream_t st, const void* src, void* dst, const RepackDims& d, int group,
                        int pad);

// ---- dispatch.hip
// Work list of the round-3 fp8 x fp8 prefill kernel: for each of the two GEMMs (rg1 / rg2 row groups per token tile),
// every (token tile, row group) item as {expert, offsets[e] + first row, rows, row group}, ordered inside each part of the
// tile list (one XCD's run, or the whole list) expert by expert, and inside an expert row group by row group with the
// expert's token tiles adjacent.  max_tiles sizes the launch; the lists hold max_tiles * rg records each.
int launch_build_items(hipStream_t st, const int32_t* tile_e, const int32_t* tile_r0, const int32_t* counts,
                       const int32_t* offsets, const int32_t* meta, int xcd_parts, int rg1, int rg2, int max_tiles,
                       int32_t* items1, int32_t* items2);

// slot i = column i % top_k of token i / top_k in an [M][ids_ld] array; id_offset is subtracted from ids >= 0
int launch_sort(hipStream_t st, const int32_t* ids, int top_k, int ids_ld, int id_offset, int n_slots, int E,
                int32_t* counts, int32_t* offsets, int32_t* sorted_slot, int32_t* pos_of_slot, int32_t* active,
                int32_t* meta, int tile_rows, int tile_min, int32_t* tile_e, int32_t* tile_r0,
                int32_t* hist, size_t hist_cap, int xcd_cap = 0);
// router + sort in one launch for decode batches (RouteArgs: routing_dev.h); same outputs as the router kernel
// followed by launch_sort
bool launch_route_sort_ok(int M, int K, int E_router, int n_group, int E_local);
int launch_route_sort(hipStream_t st, const RouteArgs& ra, int id_offset, int E, int32_t* counts, int32_t* offsets,
                      int32_t* sorted_slot, int32_t* pos_of_slot, int32_t* active, int32_t* meta, int tile_rows,
                      int tile_min, int32_t* tile_e, int32_t* tile_r0, int xcd_cap);
// tile_rows of the two sort launchers may carry a granule in its high half: pack_tile_rows(256, 32) = 256-row tiles,
// an expert's rows dealt to its tiles as evenly as 32-row granules allow (dispatch.hip tile_first_row)
inline int pack_tile_rows(int rows, int gran) { return rows | (gran << 16); }
// tile_min of the two single-workgroup sort launchers may instead carry MIXED tile heights (negative): experts with more than
// big_min rows get tiles of big_rows rows, the others tiles of tile_rows; the big tiles are the first meta[4] list entries
inline int pack_mixed_tiles(int big_rows, int big_min) { return -((big_rows << 16) | (big_min & 0xffff)); }
constexpr int kMetaInts = 32;   // meta[0..3]: see dispatch.hip; meta[8..16]: per-XCD runs of the tile list
int launch_quant_fp8_rows(hipStream_t st, const void* src, int ld_src, int adt, int R, int K, void* dst,
                          float* scales);
// int4 fast mode: sums[r][kb] = fp32 sum of the 128 elements of group kb of row r (16-bit rows, row stride ld_src)
int launch_rowsum128_rows(hipStream_t st, const void* src, int ld_src, int adt, int R, int K, float* sums);
int launch_combine(hipStream_t st, const void* y, int y_dt, int SK, size_t sk_stride,
                   const int32_t* pos_of_slot, const float* tw, int tw_ld, int M, int K, int H, void* out,
                   int out_dt);

int launch_read_probe(hipStream_t st, const void* src, size_t bytes, int n_blocks, int unroll,
                      unsigned* sink);

// ---- gemm_skinny.hip
struct GemmParams {
    // weights (pre-shuffled), scales (pre-shuffled or null)
    const void* w;
    const void* s;
    // weight addressing, in 16-byte vectors per lane group of 64: vector of (expert e, tile t, unit u, load l) =
    // e * w_estride + t * w_tstride + u * w_ustride + l * 64 + lane.  The image is tile-major (a tile's units contiguous;
    // the unit-major alternative of round 2 -- measured no faster, profiles/r02_weight_layout_ab.log -- was removed in
    // round 4, the kernels keep addressing through the three strides)
    long long w_estride, w_tstride, w_ustride;
    int spu;     // int4: scales per 128-k unit (1,2,4)
    const float* gs;   // NVFP4: per-expert f32 multiplier [E] (NULL = 1)
    int T_half;  // tiles per half (gate / up); w2: tiles total
    int halves;  // 2 gated w13, 1 otherwise
    int U;       // K units
    int Kreal;   // real K (elements) for the token-operand bounds check
    int n_real;  // real rows per half (I for GEMM1, H for GEMM2)
    // token operand
    const void* x;  // GEMM1: hidden [M][H] ; GEMM2: act [rows][ldx]
    int ldx;        // row stride of x in elements
    int top_k;      // GEMM1: slot -> token = slot / top_k
    float rcp_top_k;  // 1 / top_k (the fp8 prefill kernel divides through it: gemm_prefill_a8w.h store_table)
    const float* xscale;  // W8A8: per (token row, 128-k block) activation scales, row stride ld_xscale
    int ld_xscale;
    int round_gemm1;      // 1: round GEMM1 outputs to the act dtype before the activation (in-tree GPU
                          //    operator, block-fp8 semantics), 0: keep fp32 (CPU operator)
    // routing metadata (device)
    const int32_t* counts;
    const int32_t* offsets;
    const int32_t* active;
    const int32_t* meta;
    const int32_t* sorted_slot;
    const int32_t* tile_e;   // tiled kernels: work list of (expert, first row) token tiles
    const int32_t* tile_r0;
    const int32_t* items;    // fp8 x fp8 prefill kernel (gemm_prefill_a8w.h): its own work list, one 16-byte record
                             // {expert, first output row, rows, row group} per (token tile, row group) item in launch
                             // order (dispatch.hip build_items_kernel)
    int max_rows;            // skinny kernels: skip experts with more rows than this (0 = no limit);
                             // they are handled by the tiled kernels of the same step (hybrid dispatch)
    int stream_nt;           // 1: weights are read once (decode) -> nontemporal loads
    // outputs
    void* out;  // GEMM1: act [rows][ldo] act dtype ; GEMM2: y [SK][sk_stride] fp32
    int ldo;
    // gated GEMM1 of the fp8 prefill kernel with the intermediate's 1 x 128 quantisation fused in (else null): e4m3 bytes
    // [rows][ldo] and scales [rows][ld_qs] instead of `out`
    unsigned char* out_q;
    float* out_qs;
    int ld_qs;
    size_t sk_stride;  // GEMM2: elements between split-K slabs
    int SK;            // GEMM2: number of K splits
    int groups;        // tile groups per expert = T_half / NT
    // single-token decode (M == 1): the K slots are K distinct experts, so the scatter is the identity --
    // the GEMM kernels read the router's ids / weights themselves and the sort + combine launches go away
    const int32_t* direct_ids;   // [K] (non-null: direct mode; id < 0 = not local)
    const float* direct_w;       // [K]
    int direct_out_dt;           // LKM_DT_* of `out` in the direct GEMM2
    int direct_E, direct_id_off; // direct mode: local expert count and the offset subtracted from ids >= 0
                                 // (an id outside [0, E) after that is not local, like -1)
    // single-token decode through lkm_forward_routed: the direct GEMM1 routes the one row itself (every workgroup, on
    // its first wavefront: ~2 us under the start of the weight stream instead of a router launch and its gap);
    // workgroup (0, 0) also stores the ids / weights for GEMM2 and the caller (direct_ids / direct_w point at them)
    int route_on;
    RouteArgs route;
    long long x_rows;  // rows of the activation matrix behind `x` (bounds of the LDS-DMA buffer window)
    int tile_lo_meta, tile_hi_meta;   // gemm_tiled_kernel on a PART of the tile list (mixed tile heights): item = blockIdx.y +
                       // meta[tile_lo_meta] (0: the list's start), valid below meta[tile_hi_meta] (0: meta[3], the whole list)
    int xcd_map;       // tiled kernels: != 0 -> XCD-aware 1-D work mapping.  Host side: the longest run of tiles one
                       // XCD may get (launch_sort's xcd_cap); the launcher replaces it by the row-group count
    int y_dt;          // GEMM2: dtype of the per-row partials `out` (LKM_DT_F32, or the activation dtype where the
                       // reference rounds the GEMM2 output itself: block-fp8 W8A8, native_w8a8_block_matmul output_dtype)
    int dbg;           // development ablations of the prefill kernels (tuning key "dbg"; results are wrong when set)
    int tile_uniform_scale;   // fp8: every 16-row weight tile has one block scale per K unit (groupN % 16 == 0)
    // activation
    int act_type;
    float alpha, limit;
};
inline void set_w_layout(GemmParams& p, int T_all, int U, int loads) {
    p.w_estride = (long long)T_all * U * loads * 64;
    p.w_tstride = (long long)U * loads * 64;
    p.w_ustride = (long long)loads * 64;
}
struct LaunchCfg {
    int nt, tb, kw, sk;
    int tiled;   // 0: skinny streamer (token operand straight from L2);
                 // else token-tile rows (64 / 128): token operand staged through LDS
    int waves;   // tiled: waves per workgroup (4 / 8)
    int pd;      // tiled: weight register stages (2 / 4; prefetch distance pd-1 K units)
    int pf;      // 256-row tiles: 16-bit weights 8 = the LDS-DMA prefill kernel (gemm_prefill.h, opt-in), fp8 x fp8 9 = gemm_prefill_a8w.h;
                 // 4-bit formats at 32/64-row tiles: 5 / 6 = the 32x32-MFMA kernels (gemm_w4x.h / gemm_w4e.h)
};
int launch_gemm1(hipStream_t st, int wf, int adt, const LaunchCfg& cfg, const GemmParams& p,
                 bool gated, int max_active);
int launch_gemm2(hipStream_t st, int wf, int adt, const LaunchCfg& cfg, const GemmParams& p,
                 int max_active);
// single-token decode: GEMM2 of all K slots + weighted sum in one launch (cfg.nt, cfg.sk; K*sk <= 16)
int launch_gemm2_direct(hipStream_t st, int wf, int adt, const LaunchCfg& cfg, const GemmParams& p, int K);
// tiled variants: grid.y = max_tiles (upper bound of the device-side work list)
int launch_gemm1_tiled(hipStream_t st, int wf, int adt, const LaunchCfg& cfg, const GemmParams& p,
                       bool gated, int max_tiles);
int launch_gemm2_tiled(hipStream_t st, int wf, int adt, const LaunchCfg& cfg, const GemmParams& p,
                       int max_tiles);

}  // namespace lkm
