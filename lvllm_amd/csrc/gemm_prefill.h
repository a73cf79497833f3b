// gemm_prefill.h -- per-expert grouped GEMMs for the prefill regime (64+ rows per expert, MFMA-bound) with 16-bit
// ACTIVATIONS: bf16 / fp16 weights, fp8 e4m3 weights (W8A16, see W8 below) and the 4-bit formats (W4 below), 256 weight
// rows x 256 tokens per workgroup, round 4 structure.  What the reference's gpu_prefill runs (MOE_BF16 / MOE_FP8 /
// MOE_WNA16 ...: routed_experts.py:1884-1899).
//
// Same math as gemm_skinny.h / gemm_tiled.h (fp32 accumulation over k, the epilogues are theirs); what
// changes is how the operands reach the matrix pipe:
//   * BOTH operands go through LDS, filled by LDS-DMA (buffer_load_dwordx4 ... lds: no VGPR staging,
//     one 32-bit lane offset per stream, the K advance in an SGPR):
//       weights: the pre-shuffled image (lkm_common.h) is copied as it is, 2 KiB per (16-row tile, 64 k);
//       tokens : 128-byte row pieces gathered per lane, XOR-swizzled on the SOURCE side so that the linear
//                LDS image is conflict-free for the B-fragment reads (as in gemm_tiled.h);
//   * v_mfma_f32_32x32x16: 8 waves as 2 (weight-row halves) x 4 (64-token quarters), a wave owns 4 row
//     groups of 32 rows x 2 token groups of 32 = 8 accumulators of 16 registers.  A row group is two 16-row
//     tiles of the image read side by side (lanes 0-15 / 16-31; gated GEMM1: the gate tile and its up tile,
//     so D registers i and 8 + i of a lane are the gate and up value of one (row, token) and the activation
//     is lane-local); k-step kk of a unit takes k = 16 kk + 8 (lane / 32) .. + 7 of both operands (an MFMA
//     sums over its k, any A/B-consistent assignment is valid);
//   * the K loop runs in PHASES of one output quadrant each (2 row groups x 1 token group x 64 k = 8 MFMAs):
//         ph1 (A0,B0)   ph2 (A0,B1)   ph3 (A1,B1)   ph4 (A1,B0)        A0/A1, B0/B1 = quarters of a K tile
//     a phase = [fragment reads of the quarter it needs + one quarter of LDS-DMA for a later tile] s_barrier
//     [8 MFMAs] s_barrier, and the two wave halves run ONE barrier interval apart: while the waves of row
//     half 0 multiply, their SIMD partners of row half 1 read fragments and issue DMA, and vice versa -- the
//     matrix pipe of a SIMD always has one of its two waves in a multiply section;
//   * a quarter (16 KiB: the sub-rows every wave reads in the same phase) is re-filled two phases after its
//     last read, for the tile after next -- four quarters (64 KiB per CU) are in flight at any time in a
//     2 x 64 KiB ring, waited for with a counted vmcnt one phase before the first read (the wait, then a
//     barrier every wave passes, then the read: the only ordering an LDS-DMA has).
// Schedule of tile t (buffer t & 1), DMA issued / quarter whose landing is waited for (vmcnt(8), fp8 weights vmcnt(6)
// = all but the four youngest quarters):
//     ph1: B1(t+1) / B1(t)      ph2: A1(t+1) / A1(t)      ph3: A0(t+2) / -      ph4: B0(t+2) / A0, B0(t+1)
#pragma once
#include "gemm_tiled.h"

namespace lkm {

constexpr int kPfTokens = 256;                  // token tile
constexpr int kPfQ = 16 * 1024;                 // one quarter of a K tile: 128 rows x 64 k x 2 B
constexpr int kPfBufBytes = 4 * kPfQ;           // [A0][A1][B0][B1]
constexpr int kPfNarBytes = 3 * kPfQ;           // narrow tile (<= 128 tokens): [A0][A1][B0], three buffers
constexpr int kPfLdsBytes = 3 * kPfNarBytes;    // 144 KiB (>= the 2 x 64 KiB of a full tile)

// SiLU-mul epilogue of this kernel: the sigmoid on the transcendental unit (v_exp_f32 + v_rcp_f32, as gemm_prefill_a8w.h)
// instead of the polynomial exp and IEEE division the other kernels share with the CPU restatement -- at ~55 VALU per
// element those cost 16 % of GEMM1 here (every wave of the CU's one workgroup is in its epilogue at the same time, the
// matrix pipe idles).  Same rounding points (GemmParams::round_gemm1); the fp32 sigmoid differs in its last bits, i.e.
// one ulp of the activation dtype on ~1e-4 of the intermediate elements, far inside the operator's tolerance.
template <int ADT>
__device__ __forceinline__ void pf_silu_mul4(const GemmParams& p, const f32x4& gate, const f32x4& upv, float (&v)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float a = gate[r], up = upv[r];
        if (p.round_gemm1) {
            a = ActT<ADT>::to_f32(ActT<ADT>::from_f32(a));
            up = ActT<ADT>::to_f32(ActT<ADT>::from_f32(up));
        }
        const float sg = a * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a * -1.44269504088896341f));
        v[r] = p.round_gemm1 ? ActT<ADT>::to_f32(ActT<ADT>::from_f32(sg)) * up : sg * up;
    }
}

// the scale bytes the W4 mode loads by hand -> the decoders' Aux (gemm_skinny.h Dec<>)
template <int WF>
struct PfAux;
template <>
struct PfAux<LKM_W_INT4_B8> {
    template <typename A>
    static __device__ __forceinline__ void set(A& a, unsigned a1, u32x2 a2, int spu) {                         // 1 / 2 / 4 act-dtype scales
        a.raw = spu == 4 ? a2 : u32x2{spu == 1 ? (a1 & 0xffffu) : a1, 0u};
    }
};
template <>
struct PfAux<LKM_W_INT4_ZP> {
    template <typename A>
    static __device__ __forceinline__ void set(A& a, unsigned a1, u32x2 a2, int spu) {                         // 1 / 2 (scale, zero point) pairs
        a.raw = spu == 1 ? u32x4{a1, 0u, 0u, 0u} : u32x4{a2.x, a2.y, 0u, 0u};
    }
};
template <>
struct PfAux<LKM_W_MXFP4> {
    template <typename A>
    static __device__ __forceinline__ void set(A& a, unsigned a1, u32x2, int) { a.raw = a1; }                   // four E8M0
};
template <>
struct PfAux<LKM_W_NVFP4> {
    template <typename A>
    static __device__ __forceinline__ void set(A& a, unsigned, u32x2 a2, int) { a.raw = a2; }                   // eight e4m3
};

template <int N>
__device__ __forceinline__ void pf_wait_vmcnt() {
    static_assert(N == 5 || N == 6 || N == 7 || N == 8 || N == 10, "counts of the schedules");
    if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
}
// e4m3 -> activation dtype, exact (v_cvt_scalef32_pk_{bf16,f16}_fp8 with scale 1: two values per instruction)
template <int ADT>
__device__ __forceinline__ u32x4 pf_fp8_frag(unsigned d0, unsigned d1) {
    u32x4 o;
    if constexpr (ADT == LKM_DT_BF16) {
        o.x = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d0, 1.0f, false));
        o.y = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d0, 1.0f, true));
        o.z = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d1, 1.0f, false));
        o.w = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d1, 1.0f, true));
    } else {
        o.x = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(d0, 1.0f, false));
        o.y = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(d0, 1.0f, true));
        o.z = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(d1, 1.0f, false));
        o.w = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(d1, 1.0f, true));
    }
    return o;
}

// W8 = fp8 e4m3 weights with one fp32 scale per (16-row tile, 128 k) -- block-quantised W8A16, the "VRAM FP8" class of the
// reference (MOE_FP8.gpu_prefill): the image holds 16 rows x 128 k per 2 KiB, so a K tile of 64 k is ONE KiB per weight tile
// (half the DMA bytes of the 16-bit formats); a fragment read fetches 16 raw bytes = the A operands of k-steps j and j + 2,
// converted in registers (exact) behind the partner wave's MFMAs.  The block scales: the sum is  sum_u s_u P_u  with P_u
// the fp32 partial of unit u (gemm_skinny.h Dec<FP8>: UNIT_SCALE).  A second accumulator set for P_u does not fit, so the
// accumulators hold  (sum so far) / s_u  instead: entering unit u + 1 they are multiplied by s_u / s_{u+1} (one v_mul per
// register, the registers of the quadrant the phase is about to update, wave-uniform ratios: every row of a 16-row tile
// shares the scale -- the launcher requires block heights that are multiples of 16), leaving by s_last.  fp32 rounding
// of the ratios: <= 1e-7 relative per unit.
// (An fp8 x fp8 mode of this kernel -- 32x32x64 fp8 MFMA, weight and token scales both carried in the accumulators -- was
// built and measured in round 4 and not kept: equal to gemm_prefill_a8w.h on GEMM1, behind it on GEMM2;
// profiles/r04_prefill16_kernel.md, commit "fp8 x fp8 mode of the 16-bit prefill kernel ... experiment".)
//
// W4 = the 4-bit formats (uint4b8, MXFP4, NVFP4; MOE_WNA16 / MOE_MXFP4 / MOE_NVFP4
// .gpu_prefill): decoded in registers by every wave that needs a fragment they would cost 4 x 15-19 VALU per 8 weights and
// bind the kernel to the vector port (gemm_w4x.h's finding), so the weights are decoded ONCE per workgroup: in the place
// of a weight quarter's DMA each wave loads the raw 8 bytes + scale bytes of ITS 16-row tile (one lane = one (row, 8 k)
// piece, exactly the piece of the 16-bit image it then writes), converts them with the formats' bit-exact decoders
// (gemm_skinny.h Dec<>: T((q - 8) s) etc.) and stores the two 16-byte pieces into the image quarter the other waves read
// two or more phases later.  The raw loads are plain buffer loads issued by hand (the compiler must not see them: it
// would drain the token DMA with vmcnt(0) at their use) one K tile ahead, into the registers the decode just freed, and
// counted into the loop's vmcnt waits.  Everything after the image (fragment reads, MFMAs, epilogue) is the 16-bit kernel.
template <int WF, int ADT, bool GATED, bool IS_G1>
__global__ __launch_bounds__(512) void gemm_prefill_kernel(GemmParams p) {
    static_assert(!GATED || IS_G1, "only GEMM1 is gated");
    constexpr bool W8 = WF == LKM_W_FP8_E4M3;
    constexpr bool W4 = WF == LKM_W_INT4_B8 || WF == LKM_W_INT4_ZP || WF == LKM_W_MXFP4 || WF == LKM_W_NVFP4;
    constexpr int AG = W8 ? 1 : 2, BG = 2;                    // LDS-DMA instructions per wave and weight / token quarter (W4: two hand-issued loads)
    constexpr int UB = W4 ? 1024 : 2048;                      // bytes of a (16-row tile, K unit) in the weight image
#if defined(__HIP_DEVICE_COMPILE__)   // buffer resources / LDS-DMA builtins exist in the device pass only
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
        if (!(p.dbg & 8)) {
            // inside the run, the token tiles of ONE expert fastest, then the row group: the workgroups that stream the
            // same weight panel start together on neighbouring CUs of the XCD and stay in step (same K loop), so the
            // panel is fetched into the L2 once instead of once per token tile (tiles of an expert are adjacent in the
            // list, r0 ascending: dispatch.hip)
            const int ex = p.tile_e[ti];
            const int g0 = max(ti - p.tile_r0[ti] / kPfTokens, first);
            const int g1 = min(ti - p.tile_r0[ti] / kPfTokens + (p.counts[ex] + kPfTokens - 1) / kPfTokens, first + n_c);
            const int k = g1 - g0, local = sidx - (g0 - first) * RG;
            bx = local / k;
            ti = g0 + local % k;
        }
    }
    if (ti >= p.meta[3]) return;
    const int e = p.tile_e[ti], r0 = p.tile_r0[ti];
    const int m_e = p.counts[e], off_e = p.offsets[e];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef LKM_PF_ABL   // timing ablations (results are wrong): dbg & 16 no DMA in the K loop, & 32 every workgroup streams the same operands
    const bool abl_nodma = p.dbg & 16, abl_same = p.dbg & 32, abl_noepi = p.dbg & 64, abl_nost = p.dbg & 128;
#else
    constexpr bool abl_nodma = false, abl_same = false, abl_noepi = false, abl_nost = false;
#endif
    const int l32 = lane & 31, h = lane >> 5;
    const int wr = wave >> 2, wc = wave & 3;
    // A tile of <= 128 tokens (the stub an expert's row count leaves after its full tiles) runs the NARROW loop: one token
    // group of 32 per wave, two phases per K tile (A0 x B, A1 x B) -- half the matrix work of a full tile instead of all
    // of it with half of every wave's accumulators multiplying padding.
    const bool nar = !W4 && m_e - r0 <= 128;                  // (W4: the decode cadence is written for the two-buffer loop only)
    const int wtok = nar ? wc * 32 : wc * 64;                 // first token of this wave inside the tile
    const int T_all = p.T_half * p.halves;
    constexpr int TPH = GATED ? 8 : 16;                       // tiles per half taken by one workgroup
    const int tbase = bx * TPH;
    const int U = (W8 || W4) ? 2 * p.U : p.U;                // K tiles of 64 k

    // ---- LDS-DMA streams, two wave-instructions per wave and quarter.
    // Weight quarter s: the 8 tiles {row half wr', row group 2s + rgl of it, tile gu of the group}, LDS slot
    // wr'*4 + rgl*2 + gu = the staging wave's index.  Token quarter s: LDS row wc'*32 + i = token wc'*64 + s*32 + i.
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.w + (size_t)(abl_same ? 0 : e) * T_all * p.U * UB), 0, (int)((size_t)T_all * p.U * UB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0x7fffffff, 0x00020000);
    int asoff[2];                                             // byte offset of my tile of quarter s (wave-uniform)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int grp = (wave >> 2) * 4 + s * 2 + ((wave >> 1) & 1), gu = wave & 1;
        const int half = GATED ? gu : 0, t = tbase + (GATED ? grp : grp * 2 + gu);
        const int gt = half * p.T_half + (t < p.T_half ? t : 0);          // clamped: padded tile counts
        asoff[s] = __builtin_amdgcn_readfirstlane((int)((abl_same ? (wave & 1) : gt) * p.w_tstride * 16));
    }
    const int alane = lane * 16;
    int bvoff[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pc = q * 512 + tid;
            const int row = pc >> 3, pslot = pc & 7;
            const int lslot = pslot ^ x_swizzle<128>(row);
            const int r = nar ? r0 + row : r0 + (row >> 5) * 64 + s * 32 + (row & 31);
            const int rr = r < m_e ? r : r0;
            const int src_row = IS_G1 ? p.sorted_slot[off_e + rr] / p.top_k : off_e + rr;
            bvoff[s][q] = (abl_same ? row : src_row) * p.ldx * 2 + lslot * 16;   // < 2 GiB (checked by the launcher)
        }
    const int wustep = (int)(p.w_ustride * 16);
    // quarter ids: 0 = A0, 1 = A1, 2 = B0, 3 = B1 (= position inside a buffer)
    auto dma_at = [&](int u, auto OFF, auto QC) __attribute__((always_inline)) {
        constexpr int off = decltype(OFF)::v, qid = decltype(QC)::v;
        const int uc = u < U ? u : U - 1;                     // past the end: re-fetch the last unit into a quarter nobody reads
        if (abl_nodma && u >= 2) return;
        char* base = lds + off;
        if constexpr (qid < 2 && W4) {
            // (decoded into place: w4_decode)
        } else if constexpr (qid < 2 && W8) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (LdsPtr)(base + wave * 2048), 16, alane,
                                                     asoff[qid] + (uc >> 1) * wustep + (uc & 1) * 1024, 0, 0);
        } else if constexpr (qid < 2) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (LdsPtr)(base + wave * 2048 + ks * 1024), 16, alane,
                                                         asoff[qid] + uc * wustep + ks * 1024, 0, 0);
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (LdsPtr)(base + (q * 512 + wave * 64) * 16), 16,
                                                         bvoff[qid - 2][q], uc * 128, 0, 0);
        }
    };

    auto dma = [&](int u, auto BUF, auto QC) __attribute__((always_inline)) {
        dma_at(u, IC<decltype(BUF)::v * kPfBufBytes + decltype(QC)::v * kPfQ>{}, QC);
    };

    // ---- W4: the raw bytes of my tile of quarter s for one K tile (2 dwords: k-steps 2c, 2c + 1 of the unit) and the scale
    // bytes of my row for the unit; loaded by hand, laundered behind the counted wait that covers them
    typedef Dec<W4 ? WF : LKM_W_INT4_B8, ADT> D4;
    typedef int i32x4s __attribute__((ext_vector_type(4)));
    u32x2 w4raw[2] = {};
    unsigned w4a1[2] = {};                                     // scale bytes: one dword (uint4b8: one 16-bit scale; MXFP4: four E8M0) ...
    u32x2 w4a2[2] = {};                                        // ... or two (NVFP4: eight e4m3 block scales)
    i32x4s rs_wr = {0, 0, 0, 0}, rs_ar = {0, 0, 0, 0};
    int w4aoff[2] = {0, 0};
    const int w4dparam = WF == LKM_W_NVFP4 ? __builtin_bit_cast(int, p.gs ? p.gs[e] : 1.0f) : p.spu;
    if constexpr (W4) {
        const unsigned long long wb = (unsigned long long)((const char*)p.w + (size_t)e * T_all * p.U * UB), ab = (unsigned long long)p.s;
        rs_wr = i32x4s{__builtin_amdgcn_readfirstlane((int)wb), __builtin_amdgcn_readfirstlane((int)(wb >> 32) & 0xffff), 0x7fffffff, 0x00020000};
        rs_ar = i32x4s{__builtin_amdgcn_readfirstlane((int)ab), __builtin_amdgcn_readfirstlane((int)(ab >> 32) & 0xffff), 0x7fffffff, 0x00020000};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int grp = (wave >> 2) * 4 + s * 2 + ((wave >> 1) & 1), gu = wave & 1;
            const int half = GATED ? gu : 0, t = tbase + (GATED ? grp : grp * 2 + gu);
            const int gt = half * p.T_half + (t < p.T_half ? t : 0);
            w4aoff[s] = (int)(D4::aux_ptr(p.s, ((size_t)e * T_all + gt) * p.U, lane, p.spu) - (const char*)p.s);   // < 2 GiB: the launcher checks
        }
    }
    auto w4_load = [&](int t, auto SC) __attribute__((always_inline)) {       // raw + scale bytes of K tile t, quarter s
        constexpr int s = decltype(SC)::v;
        if constexpr (W4) {
            const int tc = t < U ? t : U - 1;
            const int wo = asoff[s] + (tc >> 1) * wustep + (tc & 1) * 8, ao = (tc >> 1) * D4::aux_step(p.spu);
            constexpr int wf = WF + 0 * s;                      // (dependent on the lambda's parameter: the untaken branches are discarded)
            asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(w4raw[s]) : "v"(alane), "s"(rs_wr), "s"(wo) : "memory");
            // (straight into the registers the decode reads: no instruction may touch them before the covering wait)
            const int& av = w4aoff[s];
            unsigned& a1 = w4a1[s];
            u32x2& a2 = w4a2[s];
            if constexpr (wf == LKM_W_INT4_B8) {
                // 1 / 2 / 4 scales per row and unit (groups of >= 128 / 64 / 32 k): 2 / 4 / 8 bytes, aligned to their size
                if (p.spu == 1) asm volatile("buffer_load_ushort %0, %1, %2, %3 offen" : "=v"(a1) : "v"(av), "s"(rs_ar), "s"(ao) : "memory");
                else if (p.spu == 2) asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(a1) : "v"(av), "s"(rs_ar), "s"(ao) : "memory");
                else asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(a2) : "v"(av), "s"(rs_ar), "s"(ao) : "memory");
            } else if constexpr (wf == LKM_W_INT4_ZP) {
                // 1 / 2 (scale, zero point) pairs per row and unit (groups of >= 128 / 64 k; 32-k groups keep the tile kernels)
                if (p.spu == 1) asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(a1) : "v"(av), "s"(rs_ar), "s"(ao) : "memory");
                else asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(a2) : "v"(av), "s"(rs_ar), "s"(ao) : "memory");
            } else if constexpr (wf == LKM_W_MXFP4)
                asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(a1) : "v"(av), "s"(rs_ar), "s"(ao) : "memory");
            else
                asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(a2) : "v"(av), "s"(rs_ar), "s"(ao) : "memory");
        }
    };
    // decode what w4_load fetched (K tile parity PC: k-steps 2 PC, 2 PC + 1 of its unit) into my tile of the image quarter at OFF
    auto w4_decode = [&](auto OFF, auto SC, auto PC) __attribute__((always_inline)) {
        constexpr int off = decltype(OFF)::v, s = decltype(SC)::v, pc = decltype(PC)::v;
        if constexpr (W4) {
            constexpr int wf0 = WF + 0 * s;
            asm volatile("" : "+v"(w4raw[s]));                   // landed: the counted wait before this call
            unsigned& a1 = w4a1[s];
            u32x2& a2 = w4a2[s];
            if constexpr (wf0 == LKM_W_NVFP4) asm volatile("" : "+v"(a2));
            else if constexpr (wf0 == LKM_W_INT4_B8 || wf0 == LKM_W_INT4_ZP) asm volatile("" : "+v"(a1), "+v"(a2));
            else asm volatile("" : "+v"(a1));
            u32x4 rawv[1];
            rawv[0] = u32x4{0u, 0u, 0u, 0u};
            rawv[0][2 * pc] = w4raw[s].x;
            rawv[0][2 * pc + 1] = w4raw[s].y;
            typename D4::Aux ax;
            PfAux<W4 ? WF : LKM_W_INT4_B8>::set(ax, w4a1[s], w4a2[s], p.spu);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                *(u32x4*)(lds + off + wave * 2048 + ks * 1024 + alane) = D4::frag(rawv, ax, 2 * pc + ks, w4dparam);
        }
    };

    // ---- fragments.  Registers (2 waves per SIMD -> 256 per lane): 128 accumulators, 8 weight fragments (one
    // quarter: 2 row groups x 4 k-steps), 2 x 4 token fragments (both quarters stay: B0 serves ph1 and ph4).
    const bool has_rows = r0 + wtok < m_e;                    // wave-uniform: my token share holds rows
    u32x4 fa[2][4], fb0[4], fb1[4];
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    // A fragment (quarter s, local row group rgl, k-step kk): tile slot wr*4 + rgl*2 + (lane & 16 ? 1 : 0), load kk / 2,
    // image lane ((kk & 1) * 2 + h) * 16 + (lane & 15)
    const int abyte = (wr * 4 + ((lane >> 4) & 1)) * 2048 + h * 256 + (lane & 15) * 16;
    int baddr[4];                                              // B fragment: row wc*32 + l32, 16-byte slot (2 kk + h) ^ swizzle
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) baddr[kk] = (wc * 32 + l32) * 128 + (((kk * 2 + h) ^ x_swizzle<128>(l32)) * 16);

    u32x4 raw[W8 ? 2 : 1][2];                                   // W8: the raw bytes of a weight quarter (k-steps j and j + 2 per read)
    auto read_a_at = [&](auto OFF) __attribute__((always_inline)) {
        constexpr int off = decltype(OFF)::v;
        if constexpr (W8) {
#pragma unroll
            for (int rgl = 0; rgl < 2; ++rgl)
#pragma unroll
                for (int j = 0; j < 2; ++j) raw[rgl][j] = *(const u32x4*)(lds + off + abyte + rgl * 4096 + j * 512);
        } else {
#pragma unroll
            for (int rgl = 0; rgl < 2; ++rgl)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    fa[rgl][kk] = *(const u32x4*)(lds + off + abyte + rgl * 4096 + (kk >> 1) * 1024 + (kk & 1) * 512);
        }
    };
    auto decode_a = [&]() __attribute__((always_inline)) {     // (after the reads have returned)
        if constexpr (W8) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int rgl = 0; rgl < 2; ++rgl)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    fa[rgl][j] = pf_fp8_frag<ADT>(raw[rgl][j].x, raw[rgl][j].y);
                    fa[rgl][j + 2] = pf_fp8_frag<ADT>(raw[rgl][j].z, raw[rgl][j].w);
                }
        }
    };
    // W8: the weight-scale streams of my four row groups x two tiles (gated: gate / up tile; else the two tiles of the
    // 32 rows).  ONE register per stream holds the ratio s_{u-1} / s_u of every unit, lane u = unit u (<= 64 units: the
    // launcher checks), formed before the first DMA from one vector load (so that every counted vmcnt of the loop covers
    // it; scalar loads inside the loop would put SMEM on the lgkm counter and turn the loop's lgkmcnt(0) waits into waits
    // for an L2 round trip).  A scale of 0 (all-zero block, contributes 0) counts as 1.
    // Per-CHANNEL scales (one per weight row, the same for every unit: RoutedExperts._process_fp8(False) -- the plan only
    // sends layouts here whose scales do not change along K unless a tile shares them): nothing to carry, the rows' scales
    // multiply the finished accumulators.
    const bool rowscale = W8 && !p.tile_uniform_scale;
    float svr[4][2], scur[4][2];
    auto scale_index = [&](int rg, int hf) __attribute__((always_inline)) {
        const int t = GATED ? tbase + wr * 4 + rg : tbase + (wr * 4 + rg) * 2 + hf;
        const int gt = (GATED ? hf * p.T_half : 0) + (t < p.T_half ? t : 0);
        return ((size_t)e * T_all + gt) * p.U * 16;
    };
    if constexpr (W8) {            // (the loads only: the ratios are formed behind the first DMAs, finish_scales)
        const float* sc = (const float*)p.s;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) svr[rg][hf] = lane < p.U ? sc[scale_index(rg, hf) + (size_t)lane * 16] : 1.0f;
    }
    auto finish_scales = [&]() __attribute__((always_inline)) {
        if constexpr (W8) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const float sl = svr[rg][hf] != 0.0f ? svr[rg][hf] : 1.0f;
                    const float prev = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane > 0 ? lane - 1 : 0) * 4, __builtin_bit_cast(int, sl)));
                    svr[rg][hf] = prev * __builtin_amdgcn_rcpf(sl);
                }
        }
    };
    // enter unit u: the ratios of row groups RG0, RG0 + 1 (wave-uniform)
    float ratio[4][2];
    auto enter_unit = [&](auto RG0, int u) __attribute__((always_inline)) {
        constexpr int rg0 = decltype(RG0)::v;
        if constexpr (W8) {
#pragma unroll
            for (int rg = rg0; rg < rg0 + 2; ++rg)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
                    ratio[rg][hf] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, svr[rg][hf]), u));
        }
    };
    // (plain v_mul_f32: the compiler would pack these into v_pk_mul_f32, which does not issue beside the partner wave's MFMAs)
    auto mul_reg_s = [&](float x, float r) __attribute__((always_inline)) { asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x) : "s"(r)); return x; };
    auto rescale = [&](auto RG0, auto TG) __attribute__((always_inline)) {
        constexpr int rg0 = decltype(RG0)::v, tg = decltype(TG)::v;
        if constexpr (W8) {
#pragma unroll
            for (int rg = rg0; rg < rg0 + 2; ++rg)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[rg][tg][i] = mul_reg_s(acc[rg][tg][i], ratio[rg][i >> 3]);
        }
    };
    auto read_b_at = [&](auto OFF, auto BSEL) __attribute__((always_inline)) {     // BSEL: token quarter 0 / 1 -> fb0 / fb1
        constexpr int off = decltype(OFF)::v, bs = decltype(BSEL)::v;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) (bs ? fb1 : fb0)[kk] = *(const u32x4*)(lds + off + baddr[kk]);
    };
    auto read_a = [&](auto BUF, auto SC) __attribute__((always_inline)) {
        read_a_at(IC<decltype(BUF)::v * kPfBufBytes + decltype(SC)::v * kPfQ>{});
    };
    auto read_b = [&](auto BUF, auto SC) __attribute__((always_inline)) {
        read_b_at(IC<decltype(BUF)::v * kPfBufBytes + (2 + decltype(SC)::v) * kPfQ>{}, SC);
    };
    auto mm = [&](auto RG0, auto TG, auto BSEL) __attribute__((always_inline)) {
        constexpr int rg0 = decltype(RG0)::v, tg = decltype(TG)::v, bs = decltype(BSEL)::v;
        // (measured with fp8 weights, whose load sections carry conversions and rescaling: without the priority GEMM1 +2 %)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int rgl = 0; rgl < 2; ++rgl)
                acc[rg0 + rgl][tg] = Mfma32<ADT>::run(fa[rgl][kk], (bs ? fb1 : fb0)[kk], acc[rg0 + rgl][tg]);
        __builtin_amdgcn_s_setprio(0);
    };
    auto bar = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // K tile t in buffer BUF; COMPUTE = false for a wave whose token quarter is empty (it keeps the DMA / barrier cadence)
    auto tile = [&](int t, auto BUF, auto COMPUTE) __attribute__((always_inline)) {
        constexpr int b = decltype(BUF)::v;
        constexpr bool comp = decltype(COMPUTE)::value;
        constexpr int VM = 2 * AG + 2 * BG;                   // all but the four youngest quarters
        const bool unit_in = W8 && comp && b == 0 && t > 0 && !rowscale;   // W8: this K tile opens a 128-k unit (buffer parity = tile parity)
        // ph1
        if constexpr (comp) {
            read_b(IC<b>{}, IC<0>{});
            __builtin_amdgcn_sched_barrier(0);
            read_a(IC<b>{}, IC<0>{});
        }
        dma(t + 1, IC<b ^ 1>{}, IC<3>{});
        if constexpr (comp) {
            if (unit_in) {
                enter_unit(IC<0>{}, t >> 1);
                rescale(IC<0>{}, IC<0>{});
            }
            decode_a();
        }
        pf_wait_vmcnt<VM>();
        bar();
        if constexpr (comp) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            mm(IC<0>{}, IC<0>{}, IC<0>{});
        }
        bar();
        // ph2
        if constexpr (comp) read_b(IC<b>{}, IC<1>{});
        dma(t + 1, IC<b ^ 1>{}, IC<1>{});
        if constexpr (W4) {                                   // A1(t + 1): decode what was fetched a K tile ago, fetch A1(t + 2)
            pf_wait_vmcnt<6>();                               // (younger: the raw A0 loads, B0(t+1), B1(t+1))
            w4_decode(IC<(b ^ 1) * kPfBufBytes + kPfQ>{}, IC<1>{}, IC<b ^ 1>{});
            w4_load(t + 2, IC<1>{});
            // (the image writes must be done before a barrier that precedes their readers: a multiplying wave waits lgkmcnt(0)
            // right behind this phase's first barrier anyway; a wave without rows has no other wait)
            if constexpr (!comp) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if constexpr (comp) {
            if (unit_in) rescale(IC<0>{}, IC<1>{});
        }
        if constexpr (!W4) pf_wait_vmcnt<VM>();
        bar();
        if constexpr (comp) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            mm(IC<0>{}, IC<1>{}, IC<1>{});
        }
        bar();
        // ph3
        if constexpr (comp) read_a(IC<b>{}, IC<1>{});
        dma(t + 2, IC<b>{}, IC<0>{});
        if constexpr (W4) {                                   // A0(t + 2) likewise (younger: B0(t+1), B1(t+1), the raw A1 loads), fetch A0(t + 3)
            pf_wait_vmcnt<6>();
            w4_decode(IC<b * kPfBufBytes>{}, IC<0>{}, IC<b>{});
            w4_load(t + 3, IC<0>{});
            if constexpr (!comp) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if constexpr (comp) {
            if (unit_in) {
                enter_unit(IC<2>{}, t >> 1);
                rescale(IC<2>{}, IC<1>{});
            }
            decode_a();
        }
        bar();
        if constexpr (comp) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            mm(IC<2>{}, IC<1>{}, IC<1>{});
        }
        bar();
        // ph4
        dma(t + 2, IC<b>{}, IC<2>{});
        if constexpr (comp) {
            if (unit_in) rescale(IC<2>{}, IC<0>{});
        }
        pf_wait_vmcnt<VM>();
        bar();
        if constexpr (comp) mm(IC<2>{}, IC<0>{}, IC<0>{});
        bar();
    };
    auto run = [&](auto COMPUTE) __attribute__((always_inline)) {
        if constexpr (W4) {
            // images of A0(0), A1(0), A0(1) first (one exposed load latency), then the issue order of the steady loop:
            // B0(0), B1(0), raw A1(1), raw A0(2), B0(1)
            w4_load(0, IC<0>{});
            w4_load(0, IC<1>{});
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            w4_decode(IC<0>{}, IC<0>{}, IC<0>{});
            w4_decode(IC<kPfQ>{}, IC<1>{}, IC<0>{});
            w4_load(1, IC<0>{});
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            w4_decode(IC<kPfBufBytes>{}, IC<0>{}, IC<1>{});
            dma(0, IC<0>{}, IC<2>{});
            dma(0, IC<0>{}, IC<3>{});
            w4_load(1, IC<1>{});
            w4_load(2, IC<0>{});
            dma(1, IC<1>{}, IC<2>{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            dma(0, IC<0>{}, IC<0>{});
            dma(0, IC<0>{}, IC<2>{});
            dma(0, IC<0>{}, IC<3>{});
            dma(0, IC<0>{}, IC<1>{});
            dma(1, IC<1>{}, IC<0>{});
            dma(1, IC<1>{}, IC<2>{});
        }
        if constexpr (decltype(COMPUTE)::value) finish_scales();
        pf_wait_vmcnt<2 * AG + 2 * BG>();                      // A0(0), B0(0) landed (and the older token-scale loads)
        bar();
        if (wr == 1) bar();                                   // row half 1 runs one barrier interval behind
        for (int t = 0; t < U; t += 2) {
            tile(t, IC<0>{}, COMPUTE);
            tile(t + 1, IC<1>{}, COMPUTE);                    // (U is even: the launcher checks)
        }
        if (wr == 0) bar();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the trailing DMA must land before the LDS is released
    };
    // NARROW tile t in buffer BUF of three ([A0][A1][B0], 48 KiB each).  Schedule, DMA issued / landing waited for:
    //     phA: A0, B0 (t+2) / A1(t)   [vmcnt(10): all but the five youngest quarters]      phB: A1(t+2) / A0, B0 (t+1)   [vmcnt(8)]
    // (a quarter is again re-filled two phases after its last read, for the tile after the next two)
    auto tile_n = [&](int t, auto BUF, auto COMPUTE) __attribute__((always_inline)) {
        constexpr int b = decltype(BUF)::v, b2 = (b + 2) % 3;
        constexpr bool comp = decltype(COMPUTE)::value;
        const bool unit_in = W8 && comp && (t & 1) == 0 && t > 0 && !rowscale;
        if constexpr (comp) {
            read_b_at(IC<b * kPfNarBytes + 2 * kPfQ>{}, IC<0>{});
            __builtin_amdgcn_sched_barrier(0);
            read_a_at(IC<b * kPfNarBytes>{});
        }
        dma_at(t + 2, IC<b2 * kPfNarBytes>{}, IC<0>{});
        dma_at(t + 2, IC<b2 * kPfNarBytes + 2 * kPfQ>{}, IC<2>{});
        if constexpr (comp) {
            if (unit_in) {
                enter_unit(IC<0>{}, t >> 1);
                rescale(IC<0>{}, IC<0>{});
            }
            decode_a();
        }
        pf_wait_vmcnt<3 * AG + 2 * BG>();                 // all but the five youngest quarters
        bar();
        if constexpr (comp) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            mm(IC<0>{}, IC<0>{}, IC<0>{});
        }
        bar();
        if constexpr (comp) read_a_at(IC<b * kPfNarBytes + kPfQ>{});
        dma_at(t + 2, IC<b2 * kPfNarBytes + kPfQ>{}, IC<1>{});
        if constexpr (comp) {
            if (unit_in) {
                enter_unit(IC<2>{}, t >> 1);
                rescale(IC<2>{}, IC<0>{});
            }
            decode_a();
        }
        pf_wait_vmcnt<3 * AG + BG>();
        bar();
        if constexpr (comp) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            mm(IC<2>{}, IC<0>{}, IC<0>{});
        }
        bar();
    };
    auto run_n = [&](auto COMPUTE) __attribute__((always_inline)) {
        dma_at(0, IC<0>{}, IC<0>{});
        dma_at(0, IC<2 * kPfQ>{}, IC<2>{});
        dma_at(0, IC<kPfQ>{}, IC<1>{});
        dma_at(1, IC<kPfNarBytes>{}, IC<0>{});
        dma_at(1, IC<kPfNarBytes + 2 * kPfQ>{}, IC<2>{});
        dma_at(1, IC<kPfNarBytes + kPfQ>{}, IC<1>{});
        if constexpr (decltype(COMPUTE)::value) finish_scales();
        pf_wait_vmcnt<3 * AG + BG>();                          // A0(0), B0(0) landed
        bar();
        if (wr == 1) bar();
        for (int t = 0;;) {
            tile_n(t, IC<0>{}, COMPUTE);
            if (++t >= U) break;
            tile_n(t, IC<1>{}, COMPUTE);
            if (++t >= U) break;
            tile_n(t, IC<2>{}, COMPUTE);
            if (++t >= U) break;
        }
        if (wr == 0) bar();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    if (!has_rows) {
        if (nar) run_n(std::false_type{});
        else run(std::false_type{});
        bar();                                                 // (the epilogue's barrier)
        return;
    }
    if (nar) run_n(std::true_type{});
    else run(std::true_type{});

    // ---- epilogue.  D layout: register i of lane (l32, h) = row 8 (i / 4) + 4 h + i % 4 of the 32-row group, token l32 --
    // a lane holds 4-element pieces of 32 different output rows, so stored directly every piece is its own 8 / 16-byte
    // write (measured: 25 % of GEMM2, the matrix pipe idle).  The K loop's LDS is free now: each wave transposes its
    // [64 tokens][64 / 128 features] block through its own 16 KiB (pieces XOR-swizzled by the token so that both the
    // piece writes and the row reads spread over the banks) and stores whole 128 - 512-byte row segments.
    char* my = lds + wave * 16384;
    const int nbase = (tbase + (GATED ? wr * 4 : wr * 8)) * 16;            // first output feature of this wave
    const size_t row0 = (size_t)(off_e + r0 + wtok);
    const int rows_here = min(m_e - (r0 + wtok), nar ? 32 : 64);           // > 0 (has_rows); a narrow tile only has token group 0
    // 16-bit outputs: ROWB bytes per token row, piece c = features 4c .. 4c+3 (8 bytes), 16-byte pairs XOR-ed with the token
    auto put16 = [&](auto ROWBC, int t, int c, const float (&v)[4]) __attribute__((always_inline)) {
        constexpr int ROWB = decltype(ROWBC)::v, P = ROWB / 16;
        *(u32x2*)(my + t * ROWB + (((c >> 1) ^ (t & (P - 1))) * 16) + (c & 1) * 8) =
            u32x2{ActT<ADT>::pack2(v[0], v[1]), ActT<ADT>::pack2(v[2], v[3])};
    };
    auto flush16 = [&](auto ROWBC) __attribute__((always_inline)) {
        constexpr int ROWB = decltype(ROWBC)::v, P = ROWB / 16, RPI = 64 / P;
        const int k = lane % P, ts = lane / P;
        unsigned short* o = (unsigned short*)p.out + row0 * p.ldo + nbase + k * 8;
        const bool whole = nbase + k * 8 + 8 <= p.n_real;
#pragma unroll
        for (int j = 0; j < 64 / RPI; ++j) {
            const int t = j * RPI + ts;
            const u32x4 d = *(const u32x4*)(my + t * ROWB + ((k ^ (t & (P - 1))) * 16));
            if (t < rows_here && !(abl_nost && d.x != 0x12345u)) {
                unsigned short* ot = o + (size_t)t * p.ldo;
                if (whole) {
                    *(u32x4*)ot = d;
                } else {
                    const unsigned dw[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (nbase + k * 8 + i < p.n_real) ot[i] = (unsigned short)(dw[i >> 1] >> ((i & 1) * 16));
                }
            }
        }
    };
    bar();                                                                  // every wave's DMA has landed, every fragment read is done
    if (W8 && rowscale) {
        const float* sc = (const float*)p.s;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const float* sr = sc + scale_index(rg, hf);                       // the 16 row scales of (tile, unit 0)
                const f32x4 s0 = *(const f32x4*)(sr + 4 * h), s1 = *(const f32x4*)(sr + 8 + 4 * h);    // rows 4h + j / 8 + 4h + j of the tile
#pragma unroll
                for (int tg = 0; tg < 2; ++tg)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[rg][tg][8 * hf + j] *= s0[j];
                        acc[rg][tg][8 * hf + 4 + j] *= s1[j];
                    }
            }
    } else if constexpr (W8) {            // leave the last unit's scale (the DMA ring is drained: plain loads again)
        const float* sc = (const float*)p.s;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const float sl = sc[scale_index(rg, hf) + (size_t)(p.U - 1) * 16];
                scur[rg][hf] = sl != 0.0f ? sl : 1.0f;
            }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int tg = 0; tg < 2; ++tg)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[rg][tg][i] *= scur[rg][i >> 3];
    }
    if constexpr (IS_G1) {
        constexpr int ROWB = GATED ? 128 : 256;
        const bool fast_silu = GATED && p.act_type == LKM_ACT_SILU;
        static_for<2>([&](auto TGC) __attribute__((always_inline)) {
            constexpr int tg = decltype(TGC)::v;
            static_for<4>([&](auto RGC) __attribute__((always_inline)) {
                constexpr int rg = decltype(RGC)::v;
                const f32x16& c = acc[rg][tg];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x4 lo = {c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]};
                    const f32x4 hi = {c[8 + 4 * q], c[9 + 4 * q], c[10 + 4 * q], c[11 + 4 * q]};
                    float v[4];
                    if constexpr (GATED) {
                        if (fast_silu) pf_silu_mul4<ADT>(p, lo, hi, v);
                        else gemm1_act4<ADT, true>(p, lo, hi, v);
                        put16(IC<ROWB>{}, tg * 32 + l32, rg * 4 + 2 * q + h, v);
                    } else {
                        gemm1_act4<ADT, false>(p, lo, lo, v);
                        put16(IC<ROWB>{}, tg * 32 + l32, rg * 8 + 2 * q + h, v);
                        gemm1_act4<ADT, false>(p, hi, hi, v);
                        put16(IC<ROWB>{}, tg * 32 + l32, rg * 8 + 4 + 2 * q + h, v);
                    }
                }
            });
        });
        if (!abl_noepi) flush16(IC<ROWB>{});
    } else if (p.y_dt != LKM_DT_F32) {
        // partial rows in the activation dtype (lkm_api.hip decides): 256 bytes per token
        static_for<2>([&](auto TGC) __attribute__((always_inline)) {
            constexpr int tg = decltype(TGC)::v;
            static_for<4>([&](auto RGC) __attribute__((always_inline)) {
                constexpr int rg = decltype(RGC)::v;
                const f32x16& c = acc[rg][tg];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v[4] = {c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]};
                    put16(IC<256>{}, tg * 32 + l32, rg * 8 + (q >> 1) * 4 + 2 * (q & 1) + h, v);
                }
            });
        });
        if (!abl_noepi) flush16(IC<256>{});
    } else {
        // fp32 partial rows: 512 bytes per token, one 32-token group per pass
        const int k = lane & 31, ts = lane >> 5;
        float* o = (float*)p.out + row0 * p.ldo + nbase + k * 4;
        const bool whole = nbase + k * 4 + 4 <= p.n_real;
        static_for<2>([&](auto TGC) __attribute__((always_inline)) {
            constexpr int tg = decltype(TGC)::v;
            static_for<4>([&](auto RGC) __attribute__((always_inline)) {
                constexpr int rg = decltype(RGC)::v;
                const f32x16& c = acc[rg][tg];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cc = rg * 8 + (q >> 1) * 4 + 2 * (q & 1) + h;   // 16-byte piece: features 4 cc .. 4 cc + 3
                    *(f32x4*)(my + l32 * 512 + ((cc ^ (l32 & 15)) * 16)) = f32x4{c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]};
                }
            });
            if (!abl_noepi) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int t = 2 * j + ts;
                    const f32x4 d = *(const f32x4*)(my + t * 512 + ((k ^ (t & 15)) * 16));
                    if (tg * 32 + t < rows_here && !(abl_nost && d[0] != 12345.f)) {
                        float* ot = o + (size_t)(tg * 32 + t) * p.ldo;
                        if (whole) {
                            *(f32x4*)ot = d;
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (nbase + k * 4 + i < p.n_real) ot[i] = d[i];
                        }
                    }
                }
            }
        });
    }
#else
    (void)p;
#endif
}

// usable when the K range is an even number of whole 64-element units (no ragged tail) and the operands fit the
// 2 GiB buffer windows; otherwise the caller stays on gemm_tiled_kernel
inline bool prefill_kernel_ok(const GemmParams& p, size_t x_rows, int wf) {
    const bool w8 = wf == LKM_W_FP8_E4M3, w4 = wf_is_4bit(wf);
    const size_t ub = w4 ? 1024 : 2048;
    return p.Kreal % 128 == 0 && (w8 ? p.U <= 64 : (w4 || p.U % 2 == 0)) && x_rows * (size_t)p.ldx * 2 < (size_t)0x7fffffff &&
           (size_t)p.T_half * p.halves * p.U * ub < (size_t)0x7fffffff;
}

template <int WF, int ADT, bool GATED, bool IS_G1>
static int launch_prefill_t(hipStream_t st, const GemmParams& p, int max_tiles) {
    constexpr size_t lds = kPfLdsBytes;
    const int TPH = GATED ? 8 : 16;
    const int RG = ceil_div(p.T_half, TPH);
    dim3 grid(RG, max_tiles), block(512);
    GemmParams pp = p;
    if (p.xcd_map) {
        pp.xcd_map = RG;
        grid = dim3(8 * p.xcd_map * RG, 1);
    }
    auto kern = gemm_prefill_kernel<WF, ADT, GATED, IS_G1>;
    LKM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    LKM_LAUNCH_GEMM(kern, grid, block, lds, st, pp);
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}

template <typename WFC, typename ADTC>
static bool launch_prefill_if(hipStream_t st, const LaunchCfg& cfg, const GemmParams& p, bool gated, bool is_g1,
                              int max_tiles, int* rc, WFC, ADTC) {
    constexpr int ADT = ADTC::v;
    constexpr int WF = WFC::v;
    if (cfg.tiled != 256 || cfg.pf != 8) return false;
    if (!prefill_kernel_ok(p, p.x_rows, WF)) {     // (pick_cfg only plans the kernel for shapes that qualify)
        set_error("gemm_prefill: K = %d / %d units or the operand sizes do not fit the kernel", p.Kreal, p.U);
        *rc = LKM_E_INVALID;
        return true;
    }
    if (is_g1) *rc = gated ? launch_prefill_t<WF, ADT, true, true>(st, p, max_tiles) : launch_prefill_t<WF, ADT, false, true>(st, p, max_tiles);
    else *rc = launch_prefill_t<WF, ADT, false, false>(st, p, max_tiles);
    return true;
}

}  // namespace lkm
