// gemm_prefill_a8.h -- per-expert grouped GEMMs for the prefill regime, fp8 weights x fp8 activations (W8A8,
// the in-tree block-fp8 semantics: fused_moe.py:1764-1819, native_w8a8_block_matmul
// tests/kernels/quant_utils.py:91-154): 256 weight rows x 256 tokens per workgroup on the MX-scaled fp8 MFMA.
//
// v_mfma_scale_f32_16x16x128_f8f6f4 multiplies a 16 x 128 by a 128 x 16 fp8 tile in ONE instruction at twice the
// rate of the legacy fp8 / bf16 MFMAs -- and 128 k is exactly the block over which the reference accumulates before
// it applies  weight-block scale x token-group scale.  So: E8M0 scales of 1.0 (0x7f) in the instruction, C = 0,
// and the fp32 result of every instruction is one block partial sum:  acc += (ws[row block, k block] *
// xs[token, k block]) * partial  -- two v_pk_fma_f32 per MFMA, issued in the shadow of the next MFMAs.
//
// Data movement (same byte geometry as the bf16 kernel of gemm_prefill.h: a K unit is 128 BYTES of every row):
//   * both operands and both scale vectors reach LDS by LDS-DMA (buffer_load ... lds), nothing is staged through
//     VGPRs and no ordinary vector-memory load exists in the K loop (one would make every wait a vmcnt(0));
//       weights: the pre-shuffled W8A8 layout (lkm_common.h) makes the two 16-byte halves of a lane's 16 x 128
//                A operand two contiguous KiB per tile: ds_read_b128 at lane*16 (+1024), conflict-free;
//       tokens : 128-byte row pieces gathered per lane, XOR-swizzled on the SOURCE side (as gemm_tiled.h);
//       scales : 16 tiles x 64 B of weight scales (one DMA), 256 x 4 B of token scales (four 4-byte DMAs);
//   * 8 waves as 2 (weight-row halves) x 4 (64-token quarters): 8 tiles x 4 token blocks = 32 accumulators;
//   * two LDS buffers of 66 KiB; a unit's B operands (4 x 8 VGPRs) live in registers for the unit, its A operands
//     stream through a 4-deep register ring three tiles ahead of their MFMAs.  ONE barrier per unit, at tile 5 of 8:
//     by then every A operand of the unit is in registers (or in flight from LDS, waited for), so the buffer is
//     handed back to the DMA for unit u+2 while tiles 5..7 are multiplied and the operands of unit u+1 (landed:
//     its DMA was issued one whole unit earlier) start streaming in.
// Gated GEMM1 pairs gate tile t with up tile t in the same wave, so the activation epilogue is lane-local.
#pragma once
#include "gemm_tiled.h"

namespace lkm {

typedef __attribute__((ext_vector_type(8))) int i32x8;

constexpr int kA8WBytes = 16 * 2048;                  // 16 weight tiles x (2 x 1 KiB)
constexpr int kA8XBytes = 256 * 128;                  // 256 token rows x 128 fp8
constexpr int kA8WsOff = kA8WBytes + kA8XBytes;       // 16 tiles x 16 fp32 weight scales
constexpr int kA8XsOff = kA8WsOff + 1024;             // 256 fp32 token scales
constexpr int kA8BufBytes = kA8XsOff + 1024;          // 67 584
constexpr int kA8LdsBytes = 2 * kA8BufBytes;

template <int ADT, bool GATED, bool IS_G1>
__global__ __launch_bounds__(512) void gemm_prefill_a8_kernel(GemmParams p) {
    static_assert(!GATED || IS_G1, "only GEMM1 is gated");
#if defined(__HIP_DEVICE_COMPILE__)   // buffer resources / LDS-DMA builtins exist in the device pass only
    typedef __attribute__((address_space(3))) void* LdsPtr;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int ti = blockIdx.y, bx = blockIdx.x;
    if (p.xcd_map) {     // XCD-aware 1-D mapping: see gemm_tiled_kernel and dispatch.hip (xcd_start)
        const int RG = p.xcd_map;
        const int L = blockIdx.x, c = L & 7, sidx = L >> 3;
        const int first = p.meta[8 + c], n_c = p.meta[9 + c] - first;
        if (sidx >= n_c * RG) return;
        ti = first + sidx / RG;
        bx = sidx % RG;
    }
    if (ti >= p.meta[3]) return;
    const int e = p.tile_e[ti], r0 = p.tile_r0[ti];
    const int m_e = p.counts[e], off_e = p.offsets[e];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // an SGPR: LDS-DMA bases (M0) and the role branches are scalar
    const int g = lane >> 4, j = lane & 15;
    const int wr = wave >> 2, wc = wave & 3;
    const int T_all = p.T_half * p.halves;
    constexpr int TPH = GATED ? 8 : 16;                       // tiles per half taken by one workgroup
    const int tbase = bx * TPH;
    const int U = p.U;
    // ragged last tile of an expert (GLM-4.5-Air: 512 +- 22 rows over 256-row tiles): only the 64-token quarters that
    // hold rows are fetched, and a wave whose quarter is empty keeps the DMA / barrier cadence but multiplies nothing
    const int nq = ((m_e - r0 < 256 ? m_e - r0 : 256) + 63) >> 6;

    // global tile of local tile tl (0..15): gated = 8 gate tiles then the 8 up tiles of the same rows
    auto gtile = [&](int tl) __attribute__((always_inline)) {
        const int half = GATED ? tl >> 3 : 0, idx = GATED ? tl & 7 : tl;
        const int t = tbase + idx;
        return half * p.T_half + (t < p.T_half ? t : 0);                 // clamped: padded tile counts
    };

    // ---- LDS-DMA streams (wave-instructions per unit: 4 weight KiB + 4 token KiB per wave, + 1 scale piece)
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.w + (size_t)e * T_all * U * 2048), 0, (int)((size_t)T_all * U * 2048), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.s + (size_t)e * T_all * U * 64), 0, (int)((size_t)T_all * U * 64), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_xs = __builtin_amdgcn_make_buffer_rsrc((void*)p.xscale, 0, 0x7fffffff, 0x00020000);
    int asoff[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) asoff[q] = __builtin_amdgcn_readfirstlane((int)(gtile(2 * wave + q) * p.w_tstride * 16));
    const int wub = (int)(p.w_ustride * 16);       // bytes between consecutive K units of a tile (GemmParams::w_ustride)
    const int alane = lane * 16;
    int bvoff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int pc = q * 512 + tid;
        const int row = pc >> 3, pslot = pc & 7;
        const int lslot = pslot ^ x_swizzle<128>(row);
        const int r = r0 + row;
        const int rr = r < m_e ? r : r0;
        const int src_row = IS_G1 ? p.sorted_slot[off_e + rr] / p.top_k : off_e + rr;
        bvoff[q] = src_row * p.ldx + lslot * 16;              // fp8: bytes == elements; < 2 GiB (launcher)
    }
    // scales: wave 0 fetches the 16 x 64 B of weight scales, waves 1..4 the 4 x 64 token scales
    int svoff = 0;
    if (wave == 0) {
        svoff = gtile(lane >> 2) * U * 64 + (lane & 3) * 16;
    } else if (wave <= 4) {
        const int r = r0 + (wave - 1) * 64 + lane;
        const int rr = r < m_e ? r : r0;
        const int src_row = IS_G1 ? p.sorted_slot[off_e + rr] / p.top_k : off_e + rr;
        svoff = src_row * p.ld_xscale * 4;
    }
    auto dma_unit_aux = [&](int u, auto BUF, auto AUX) __attribute__((always_inline)) {
        constexpr int buf = decltype(BUF)::v, aux = decltype(AUX)::v;
        char* base = lds + buf * kA8BufBytes;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int ld = 0; ld < 2; ++ld)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (LdsPtr)(base + ((2 * wave + q) * 2 + ld) * 1024), 16,
                                                         alane, asoff[q] + u * wub + ld * 1024, 0, aux);
        if (!(p.dbg & 32))       // (ablation: weights only)
#pragma unroll
        for (int q = 0; q < 4; ++q)     // instruction q moves token rows [64 q, 64 q + 64)
            if (q < nq)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (LdsPtr)(base + kA8WBytes + (q * 512 + wave * 64) * 16), 16,
                                                         bvoff[q], u * 128, 0, aux);
        if (p.dbg & 64) return;  // (ablation: no scale vectors)
        if (wave == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_s, (LdsPtr)(base + kA8WsOff), 16, svoff, u * 64, 0, 0);
        else if (wave <= nq)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_xs, (LdsPtr)(base + kA8XsOff + (wave - 1) * 256), 4, svoff,
                                                     u * 4, 0, 0);
    };
    auto dma_unit = [&](int u, auto BUF) __attribute__((always_inline)) { dma_unit_aux(u, BUF, IC<0>{}); };
    auto sync_all = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- operands in registers
    struct ATile {
        i32x8 a;          // .lo: k = g*16 .. +15, .hi: k = 64 + g*16 .. +15 of this lane's weight row (the two DMA'd KiB)
        float ws;         // the tile's weight-block scale for the unit
    };
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    ATile ring[4];
    i32x8 fb[4];
    float xsv[4];
    f32x4 acc[8][4];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[t][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // LDS addresses: a wave's local tile of step t is a compile-time function of t plus wr
    const int a_lane = (GATED ? wr * 4 : wr * 8) * 2048 + lane * 16;      // + imm(t) (+ 1024)
    const int ws_lane = kA8WsOff + (GATED ? wr * 4 : wr * 8) * 64;        // + imm(t): row 0 of the tile
    const int brow = (wc * 64 + j) * 128;                                 // + b * 2048
    const int bsw0 = kA8WBytes + brow + ((g ^ x_swizzle<128>(j)) * 16);
    const int bsw1 = kA8WBytes + brow + (((4 + g) ^ x_swizzle<128>(j)) * 16);
    const int xs_lane = kA8XsOff + (wc * 64 + j) * 4;                     // + b * 64

    auto load_a = [&](ATile& r, auto BUF, auto TC) __attribute__((always_inline)) {
        constexpr int buf = decltype(BUF)::v, t = decltype(TC)::v;
        constexpr int tl = GATED ? (t < 4 ? t : 8 + (t - 4)) : t;        // relative to the wave's first tile
        const char* base = lds + buf * kA8BufBytes;
        r.a.lo = *(const i32x4*)(base + a_lane + tl * 2048);
        r.a.hi = *(const i32x4*)(base + a_lane + tl * 2048 + 1024);
        r.ws = *(const float*)(base + ws_lane + tl * 64);
    };
    auto load_b = [&](auto BUF, auto BC) __attribute__((always_inline)) {
        constexpr int buf = decltype(BUF)::v, b = decltype(BC)::v;
        const char* base = lds + buf * kA8BufBytes;
        fb[b].lo = *(const i32x4*)(base + bsw0 + b * 2048);
        fb[b].hi = *(const i32x4*)(base + bsw1 + b * 2048);
        xsv[b] = *(const float*)(base + xs_lane + b * 64);
    };

    // The accumulator update of a tile (scale product, four fused multiply-adds per MFMA) is issued one tile LATE, one
    // MFMA's worth between each pair of the next tile's MFMAs: a wave then covers its own VALU work with its own
    // matrix work.  (Issued straight behind their MFMAs, the updates of the two waves of a SIMD -- released together
    // by the barrier and in phase ever after -- queue on the VALU while the matrix pipe idles, and vice versa:
    // measured 32 % MFMA-busy, 42 % VALU-busy, in sequence.)
    f32x4 part[2][4];
    f32x2 fprev[4];     // {f, f} pairs behind an empty asm: packed fp32 with default operand selects only (the op_sel
                        // hazard of lkm_common.h splat2_opaque; tools/scan_pk_swizzle.py checks the generated code)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        part[0][b] = part[1][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        fprev[b] = splat2_opaque(0.f);
    }
    // unit u lives in buffer BUF; on entry ring[0..2] = its tiles 0..2, fb / xsv = its token operands
    auto unit = [&](int u, auto BUF) __attribute__((always_inline)) {
        constexpr int buf = decltype(BUF)::v;
        static_for<8>([&](auto TC) __attribute__((always_inline)) {
            constexpr int t = decltype(TC)::v;
            constexpr int tp = (t + 7) & 7;                  // the tile whose update is issued in this step
            if constexpr (t == 5) {
                // every A operand of this unit is in registers or in flight from LDS: after the waits nobody reads
                // BUF any more, and the DMA of unit u+1 (issued a unit ago) has landed in the other buffer
                sync_all();
                if (u + 2 < U && !(p.dbg & 1)) dma_unit(u + 2, IC<buf>{});
            }
            if constexpr (t + 3 < 8) load_a(ring[(t + 3) & 3], IC<buf>{}, IC<t + 3>{});
            else load_a(ring[(t + 3) & 3], IC<buf ^ 1>{}, IC<t + 3 - 8>{});
            const ATile& r = ring[t & 3];
            f32x2 fnow[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) fnow[b] = splat2_opaque(r.ws * xsv[b]);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                part[t & 1][b] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                    r.a, fb[b], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                const f32x4 pp = part[(t & 1) ^ 1][b];
                const f32x2 f = fprev[b];
                f32x4& c = acc[tp][b];
                const f32x2 lo = __builtin_elementwise_fma(f32x2{pp.x, pp.y}, f, f32x2{c.x, c.y});
                const f32x2 hi = __builtin_elementwise_fma(f32x2{pp.z, pp.w}, f, f32x2{c.z, c.w});
                c = f32x4{lo.x, lo.y, hi.x, hi.y};
                // the update is complete HERE: without this the optimiser sinks the (memory-free) MFMAs of tiles
                // 0..4 below the barrier of tile 5, every operand stays live across it and 200 VGPRs spill
                asm volatile("" : "+v"(c));
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) fprev[b] = fnow[b];
            if constexpr (t == 7) {
                // the unit's token operands and token scales are done with: the next unit's replace them
                static_for<4>([&](auto BC) __attribute__((always_inline)) { load_b(IC<buf ^ 1>{}, BC); });
            }
            // issue order inside the step: the three operand reads of tile t+3 and the four scale products first, then
            // MFMA b / the four multiply-adds of the previous tile's block b, alternating
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
            // keep the operand reads of tile t+3 and the MFMAs of tile t in THIS step: hoisting every read of the
            // unit to its top would need all eight A operands live at once (64 VGPRs more than the file holds)
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // ---- prologue: unit 0 lands, unit 1 is in flight while the first operands are read
    dma_unit(0, IC<0>{});
    sync_all();
    if (U > 1) dma_unit(1, IC<1>{});
    static_for<4>([&](auto BC) __attribute__((always_inline)) { load_b(IC<0>{}, BC); });
    load_a(ring[0], IC<0>{}, IC<0>{});
    load_a(ring[1], IC<0>{}, IC<1>{});
    load_a(ring[2], IC<0>{}, IC<2>{});
    if ((p.dbg & 2) || wc >= nq) {   // a wave without token rows (or the ablation): DMA / barrier cadence only
        for (int u = 0; u < U; ++u) {
            sync_all();
            if (u + 2 < U) {
                if (u & 1) dma_unit(u + 2, IC<1>{});
                else dma_unit(u + 2, IC<0>{});
            }
        }
    } else
    for (int u = 0; u < U; u += 2) {
        unit(u, IC<0>{});
        if (u + 1 < U) unit(u + 1, IC<1>{});
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // trailing operand reads
    {   // the update of the very last tile
        constexpr int lastp = 1;                              // tile 7 wrote part[7 & 1]
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const f32x4 pp = part[lastp][b];
            const f32x2 f = fprev[b];
            f32x4& c = acc[7][b];
            const f32x2 lo = __builtin_elementwise_fma(f32x2{pp.x, pp.y}, f, f32x2{c.x, c.y});
            const f32x2 hi = __builtin_elementwise_fma(f32x2{pp.z, pp.w}, f, f32x2{c.z, c.w});
            c = f32x4{lo.x, lo.y, hi.x, hi.y};
        }
    }

    // ---- epilogue (D layout lane (g,j): rows tile*16 + g*4 + r, token column j of block b)
    static_for<4>([&](auto BC) __attribute__((always_inline)) {
        constexpr int b = decltype(BC)::v;
        const int r_tok = r0 + (wc * 4 + b) * 16 + j;
        if (r_tok < m_e) {
            static_for<GATED ? 4 : 8>([&](auto TC) __attribute__((always_inline)) {
                constexpr int t = decltype(TC)::v;
                const int tl = GATED ? wr * 4 + t : wr * 8 + t;          // tile index inside the half
                const int n = (tbase + tl) * 16 + g * 4;
                if (tbase + tl < p.T_half && n < p.n_real) {
                    if constexpr (IS_G1) store_gemm1_frag<ADT, GATED>(p, acc[t][b], acc[GATED ? 4 + t : t][b], (size_t)(off_e + r_tok), n);
                    else if (p.y_dt == LKM_DT_F32) store_gemm2_frag(p, acc[t][b], 0, (size_t)(off_e + r_tok), n);
                    else {          // the reference's block-fp8 GEMM rounds its output to the activation dtype
                        unsigned short* o = (unsigned short*)p.out + (size_t)(off_e + r_tok) * p.ldo + n;
                        if (n + 4 <= p.n_real) {
                            *(u32x2*)o = u32x2{ActT<ADT>::pack2(acc[t][b].x, acc[t][b].y), ActT<ADT>::pack2(acc[t][b].z, acc[t][b].w)};
                        } else {
                            const f32x4 v = acc[t][b];
                            if (n + 0 < p.n_real) o[0] = ActT<ADT>::from_f32(v.x);
                            if (n + 1 < p.n_real) o[1] = ActT<ADT>::from_f32(v.y);
                            if (n + 2 < p.n_real) o[2] = ActT<ADT>::from_f32(v.z);
                            if (n + 3 < p.n_real) o[3] = ActT<ADT>::from_f32(v.w);
                        }
                    }
                }
            });
        }
    });
#else
    (void)p;
#endif
}

// usable when K is a whole number of 128-byte units, every 16-row weight tile has ONE block scale (groupN a
// multiple of 16) and the operand matrices fit 2 GiB buffer windows; otherwise the plan stays on gemm_tiled_kernel
inline bool prefill_a8_ok(const GemmParams& p) {
    return p.Kreal % 128 == 0 && p.tile_uniform_scale &&
           (size_t)p.x_rows * (size_t)p.ldx < (size_t)0x7fffffff &&
           (size_t)p.x_rows * (size_t)p.ld_xscale * 4 < (size_t)0x7fffffff &&
           (size_t)p.T_half * p.halves * p.U * 2048 < (size_t)0x7fffffff;
}

template <int ADT, bool GATED, bool IS_G1>
static int launch_prefill_a8_t(hipStream_t st, const GemmParams& p, int max_tiles) {
    constexpr size_t lds = (size_t)kA8LdsBytes;
    const int TPH = GATED ? 8 : 16;
    const int RG = ceil_div(p.T_half, TPH);
    dim3 grid(RG, max_tiles), block(512);
    GemmParams pp = p;
    if (p.xcd_map) {
        pp.xcd_map = RG;
        grid = dim3(8 * p.xcd_map * RG, 1);      // p.xcd_map = upper bound of the tiles in one XCD's run (host)
    }
    auto kern = gemm_prefill_a8_kernel<ADT, GATED, IS_G1>;
    LKM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, grid, block, lds, st, pp);
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}

template <typename ADTC>
static bool launch_prefill_a8_if(hipStream_t st, const LaunchCfg& cfg, const GemmParams& p, bool gated, bool is_g1,
                                 int max_tiles, int* rc, ADTC) {
    constexpr int ADT = ADTC::v;
    if (cfg.tiled != 256 || cfg.pf != 8) return false;
    if (!prefill_a8_ok(p)) {
        set_error("fp8 W8A8 prefill kernel: shape not eligible (K %% 128, scale granularity or 2 GiB windows)");
        *rc = LKM_E_INVALID;
        return true;
    }
    if (is_g1) *rc = gated ? launch_prefill_a8_t<ADT, true, true>(st, p, max_tiles) : launch_prefill_a8_t<ADT, false, true>(st, p, max_tiles);
    else *rc = launch_prefill_a8_t<ADT, false, false>(st, p, max_tiles);
    return true;
}

}  // namespace lkm
