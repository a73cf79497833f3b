// gemm_prefill.h -- per-expert grouped GEMMs for the prefill regime (hundreds of rows per expert,
// MFMA-bound): 16-bit weights, 256 weight rows x 256 tokens per workgroup.
//
// Same math as gemm_skinny.h / gemm_tiled.h; what changes is how the operands reach the MFMAs:
//   * BOTH operands go through LDS, filled by LDS-DMA (buffer_load_dwordx4 ... lds: no VGPR staging,
//     one 32-bit lane offset per stream, the K advance in an SGPR):
//       weights: the pre-shuffled layout (lkm_common.h) makes a 16x32 A fragment one contiguous KiB,
//                so one wave-wide DMA drops a ready-to-read fragment image into LDS and ds_read_b128
//                at lane*16 fetches it conflict-free;
//       tokens : 128-byte row pieces gathered per lane, XOR-swizzled on the SOURCE side so the linear
//                LDS image is conflict-free for the B-fragment reads (as in gemm_tiled.h).
//   * 8 waves as 2 (weight-row halves) x 4 (64-token quarters): a wave owns 8 weight tiles x 4 token
//     blocks = 32 accumulators (128 registers); a weight fragment is shared by the 4 waves of a row
//     half, a token fragment by the 2 waves of a quarter: 12 ds_read_b128 per 32 MFMAs.
//   * K unit = 64 (two k-steps), two LDS buffers, fragments replaced on the fly:
//         phase A (k-step 0 of unit u): MFMAs; the fragments of (u, k-step 1) stream in behind them
//           -> vmcnt(0) [the DMA of unit u+1, issued a full unit ago] + lgkmcnt(0) + s_barrier
//         phase B (k-step 1 of unit u): issue the DMA of unit u+2 into the buffer everybody just
//           finished reading; MFMAs; the fragments of (u+1, k-step 0) stream in.
//     ONE barrier per 64 MFMAs per wave, no wait for a load younger than a full unit, no branch in
//     the steady loop.
// Gated GEMM1 pairs gate tile t with up tile t in the same wave (tiles 0..3 / 4..7 of its 8), so the
// activation epilogue is lane-local, as in the other kernels.
#pragma once
#include "gemm_tiled.h"

namespace lkm {

constexpr int kPfTokens = 256;          // token tile
constexpr int kPfTiles = 16;            // weight tiles per workgroup (gated: 8 gate + 8 up)
constexpr int kPfABytes = kPfTiles * 2 * 1024;                       // 16 tiles x 2 k-steps x 1 KiB
constexpr int kPfBufBytes = kPfABytes + kPfTokens * 128;             // + token rows, one unit

// WAVES = 8: 2 x 4 waves, 8 tiles x 4 token blocks each, two waves per SIMD (256 registers per lane).
// Measured and dropped (profiles/r01_prefill_pmc.md): WAVES = 4 (2 x 2 waves, one per SIMD, 256 AGPR
// accumulators): 2551 vs 1793 us on GLM prefill GEMM1; a variant with the tokens in a 3-buffer DMA ring and
// the weights in a 3-stage register ring (128 KiB in flight per CU instead of 64): 1769 us, no gain.
template <int ADT, bool GATED, bool IS_G1, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void gemm_prefill_kernel(GemmParams p) {
    static_assert(!GATED || IS_G1, "only GEMM1 is gated");
    static_assert(WAVES == 4 || WAVES == 8, "2 x 2 or 2 x 4 waves");
#if defined(__HIP_DEVICE_COMPILE__)   // buffer resources / LDS-DMA builtins exist in the device pass only
    constexpr int kPfThreads = WAVES * 64;
    constexpr int WC = WAVES / 2;                  // waves along the tokens
    constexpr int NBW = 16 / WC;                   // 16-token blocks per wave
    constexpr int ATW = 16 / WAVES;                // weight tiles each wave stages per unit
    constexpr int BPT = 2048 / kPfThreads;         // 16-byte token pieces each thread stages per unit
    typedef __attribute__((address_space(3))) void* LdsPtr;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int ti = blockIdx.y, bx = blockIdx.x;
    if (p.xcd_map) {     // XCD-aware 1-D mapping, see gemm_tiled_kernel
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
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, j = lane & 15;
    const int wr = wave / WC, wc = wave % WC;
    const int T_all = p.T_half * p.halves;
    constexpr int TPH = GATED ? 8 : 16;                       // tiles per half taken by one workgroup
    const int tbase = bx * TPH;
    const int U = p.U;

    // ---- LDS-DMA streams: 64 wave-instructions per unit, 8 per wave: 4 weight fragments + 4 token pieces.
    // Weight stream: buffer = this expert's matrix; wave `w` stages local tiles 2w, 2w+1 (SGPR offsets),
    // lane offset lane*16.  Token stream: buffer = the activation matrix; lane offset = my row piece.
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.w + (size_t)e * T_all * U * 2048), 0, (int)((size_t)T_all * U * 2048), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0x7fffffff, 0x00020000);
    int asoff[ATW];                                           // byte offset of my tiles (wave-uniform)
#pragma unroll
    for (int q = 0; q < ATW; ++q) {
        const int tl = ATW * wave + q;
        const int half = GATED ? tl / 8 : 0, idx = GATED ? tl % 8 : tl;
        const int t = tbase + idx;
        const int gt = half * p.T_half + (t < p.T_half ? t : 0);          // clamped: padded tile counts
        asoff[q] = __builtin_amdgcn_readfirstlane((int)(gt * p.w_tstride * 16));
    }
    const int alane = lane * 16;
    int bvoff[BPT];
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
        const int pc = q * kPfThreads + tid;
        const int row = pc >> 3, pslot = pc & 7;
        const int lslot = pslot ^ x_swizzle<128>(row);
        const int r = r0 + row;
        const int rr = r < m_e ? r : r0;
        const int src_row = IS_G1 ? p.sorted_slot[off_e + rr] / p.top_k : off_e + rr;
        bvoff[q] = src_row * p.ldx * 2 + lslot * 16;          // < 2 GiB (checked by the launcher)
    }
    auto dma_unit = [&](int u, auto BUF) __attribute__((always_inline)) {
        constexpr int buf = decltype(BUF)::v;
#pragma unroll
        for (int q = 0; q < ATW; ++q)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    rs_w, (LdsPtr)(lds + buf * kPfBufBytes + ((ATW * wave + q) * 2 + ks) * 1024), 16, alane,
                    asoff[q] + u * (int)(p.w_ustride * 16) + ks * 1024, 0, 0);
#pragma unroll
        for (int q = 0; q < BPT; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rs_x, (LdsPtr)(lds + buf * kPfBufBytes + kPfABytes + (q * kPfThreads + wave * 64) * 16), 16,
                bvoff[q], u * 128, 0, 0);
    };
    auto sync_unit = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    const bool has_rows = r0 + wc * NBW * 16 < m_e;           // wave-uniform: my token share holds rows
    if (!has_rows) {
        // nothing to multiply: keep the DMA / barrier cadence of the workgroup
        dma_unit(0, IC<0>{});
        dma_unit(1, IC<1>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int u = 0; u < U; u += 2) {
            sync_unit();
            dma_unit(u + 2 < U ? u + 2 : U - 1, IC<0>{});
            sync_unit();
            dma_unit(u + 3 < U ? u + 3 : U - 1, IC<1>{});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ---- fragments.  Register budget (2 waves per SIMD -> 256 per lane): 128 accumulators + 8 weight
    // fragments + two sets of 4 token fragments.  The next phase's weight fragment t is read into the
    // registers of the current one right after its four MFMAs were issued (an MFMA reads its operands at
    // issue).  LDS addresses are two lane registers + immediates: the XOR swizzle of a token row only
    // depends on j (blocks are 16 rows apart), tiles are 2 KiB apart.
    u32x4 fa[8], fb0[NBW], fb1[NBW];
    const int brow = (wc * NBW * 16 + j) * 128;                                   // + b*2048
    const int bsw0 = brow + ((g ^ x_swizzle<128>(j)) * 16), bsw1 = brow + (((4 + g) ^ x_swizzle<128>(j)) * 16);
    const int abyte = (GATED ? wr * 4 : wr * 8) * 2048 + lane * 16;               // + imm(t) (+ ks*1024)
    f32x4 acc[8][NBW];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int b = 0; b < NBW; ++b) acc[t][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one phase = 8 * NBW MFMAs on (fa, bc); when NEXT, the fragments of (buffer NBUF, k-step NKS) replace fa
    // in place and fill bn
    auto phase = [&](const u32x4 (&bc)[NBW], u32x4 (&bn)[NBW], auto NEXT, auto NBUF, auto NKS) __attribute__((always_inline)) {
        constexpr bool next = decltype(NEXT)::value;
        constexpr int nbuf = decltype(NBUF)::v, nks = decltype(NKS)::v;
        const char* nbase = lds + nbuf * kPfBufBytes;
        if constexpr (next) {
#pragma unroll
            for (int b = 0; b < NBW; ++b) bn[b] = *(const u32x4*)(nbase + kPfABytes + (nks ? bsw1 : bsw0) + b * 2048);
        }
        static_for<8>([&](auto TC) __attribute__((always_inline)) {
            constexpr int t = decltype(TC)::v;
#pragma unroll
            for (int b = 0; b < NBW; ++b) acc[t][b] = ActT<ADT>::mfma(fa[t], bc[b], acc[t][b]);
            if constexpr (next) {
                constexpr int imm = (GATED ? (t < 4 ? t : 8 + (t - 4)) : t) * 2048 + nks * 1024;
                fa[t] = *(const u32x4*)(nbase + abyte + imm);
            }
            __builtin_amdgcn_sched_barrier(0);                // keep the read behind ITS MFMAs
        });
    };
    // unit u in buffer BUF: phase A, mid barrier, DMA of unit u+2 into BUF, phase B.  No conditionals: past
    // the end the DMA re-fetches the last unit into a buffer nobody reads any more and the fragments that
    // stream in are never multiplied (U is even: the launcher checks), so the loop is one straight body.
    auto unit = [&](int u, auto BUF) __attribute__((always_inline)) {
        constexpr int buf = decltype(BUF)::v;
        phase(fb0, fb1, std::true_type{}, IC<buf>{}, IC<1>{});
        sync_unit();
        dma_unit(u + 2 < U ? u + 2 : U - 1, IC<buf>{});
        phase(fb1, fb0, std::true_type{}, IC<buf ^ 1>{}, IC<0>{});
        __builtin_amdgcn_sched_barrier(0);
    };

    dma_unit(0, IC<0>{});
    dma_unit(1, IC<1>{});
    if constexpr (WAVES == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // unit 0 landed, unit 1 in flight
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int b = 0; b < NBW; ++b) fb0[b] = *(const u32x4*)(lds + kPfABytes + bsw0 + b * 2048);
#pragma unroll
    for (int t = 0; t < 8; ++t) fa[t] = *(const u32x4*)(lds + abyte + (GATED ? (t < 4 ? t : 8 + (t - 4)) : t) * 2048);
    for (int u = 0; u < U; u += 2) {
        unit(u, IC<0>{});
        unit(u + 1, IC<1>{});
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the trailing DMA / fragment reads

    // ---- epilogue (D layout lane (g,j): rows tile*16 + g*4 + r, token column j of block b)
    static_for<NBW>([&](auto BC) __attribute__((always_inline)) {
        constexpr int b = decltype(BC)::v;
        const int r_tok = r0 + (wc * NBW + b) * 16 + j;
        if (r_tok < m_e) {
            static_for<GATED ? 4 : 8>([&](auto TC) __attribute__((always_inline)) {
                constexpr int t = decltype(TC)::v;
                const int tl = GATED ? wr * 4 + t : wr * 8 + t;          // tile index inside the half
                const int n = (tbase + tl) * 16 + g * 4;
                if (tbase + tl < p.T_half && n < p.n_real) {
                    if constexpr (IS_G1) store_gemm1_frag<ADT, GATED>(p, acc[t][b], acc[GATED ? 4 + t : t][b], (size_t)(off_e + r_tok), n);
                    else store_gemm2_frag(p, acc[t][b], 0, (size_t)(off_e + r_tok), n);
                }
            });
        }
    });
#else
    (void)p;
#endif
}

// usable when the K range is an even number of whole 64-element units (no ragged tail) and the activation
// matrix fits a 2 GiB buffer window; otherwise the caller stays on gemm_tiled_kernel
inline bool prefill_kernel_ok(const GemmParams& p, size_t x_rows) {
    return p.Kreal % 128 == 0 && p.U % 2 == 0 && x_rows * (size_t)p.ldx * 2 < (size_t)0x7fffffff &&
           (size_t)p.T_half * p.halves * p.U * 2048 < (size_t)0x7fffffff;
}

template <int ADT, bool GATED, bool IS_G1, int WAVES>
static int launch_prefill_t(hipStream_t st, const GemmParams& p, int max_tiles) {
    constexpr size_t lds = 2 * (size_t)kPfBufBytes;
    const int TPH = GATED ? 8 : 16;
    const int RG = ceil_div(p.T_half, TPH);
    dim3 grid(RG, max_tiles), block(WAVES * 64);
    GemmParams pp = p;
    if (p.xcd_map) {
        pp.xcd_map = RG;
        grid = dim3(8 * p.xcd_map * RG, 1);
    }
    auto kern = gemm_prefill_kernel<ADT, GATED, IS_G1, WAVES>;
    LKM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, grid, block, lds, st, pp);
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}

template <typename ADTC>
static bool launch_prefill_if(hipStream_t st, const LaunchCfg& cfg, const GemmParams& p, bool gated, bool is_g1,
                              int max_tiles, int* rc, ADTC) {
    constexpr int ADT = ADTC::v;
    if (cfg.tiled != 256 || cfg.pf != 8) return false;
    if (!prefill_kernel_ok(p, p.x_rows)) return false;
    if (is_g1) *rc = gated ? launch_prefill_t<ADT, true, true, 8>(st, p, max_tiles) : launch_prefill_t<ADT, false, true, 8>(st, p, max_tiles);
    else *rc = launch_prefill_t<ADT, false, false, 8>(st, p, max_tiles);
    return true;
}

}  // namespace lkm
