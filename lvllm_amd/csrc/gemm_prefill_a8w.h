// gemm_prefill_a8w.h -- per-expert grouped GEMMs for the prefill regime, fp8 weights x fp8 activations (W8A8, the
// in-tree block-fp8 semantics: fused_moe.py:298-610, native_w8a8_block_matmul tests/kernels/quant_utils.py:91-154),
// round-3 kernel: 256 weight rows x up to 256 tokens per workgroup on v_mfma_scale_f32_16x16x128_f8f6f4.
//
// What round 2's kernel (gemm_prefill_a8.h) was bound by: ONE 66 KiB LDS-DMA burst in flight per CU (both operands
// through two LDS buffers), waves of a SIMD released in phase by a per-unit barrier, operand reads scheduled by the
// compiler with lgkmcnt(0) in front of every step (profiles/r02_glm_a8_prefill.md).  This one:
//   * WEIGHTS never touch LDS.  The pre-shuffled image (lkm_common.h) makes a wave-wide buffer_load_dwordx4 the A
//     operand itself; a wave owns two 16-row tiles for the whole K loop and streams them HBM/L2 -> VGPR through a
//     3-slot register ring (2 units = 8 KiB per wave in flight while a third is multiplied);
//   * TOKENS go through a 4-stage LDS ring (32 KiB per 128-k unit), filled by LDS-DMA two to three units ahead, read
//     as B operands by all eight waves (a wave multiplies its two tiles by EVERY 16-token block of the tile);
//   * token scales ride in 16-byte pieces (4 units per token) into a 2 x 4 KiB LDS ring; weight-block scales sit in
//     two registers per wave (lane l = unit l) and reach the multiplier by v_readlane;
//   * every load, DMA, wait and barrier of the K loop is issued from inline asm with hand-counted vmcnt (a load the
//     compiler can see next to an LDS-DMA turns each of its waits into vmcnt(0)); ~190 KiB in flight per CU;
//   * the K loop's operands live in a FIXED register map (v40..v255, below) that the compiler never allocates
//     (amdgpu_num_vgpr caps it at v0..v39), so values that arrive asynchronously are never copied early;
//   * ONE barrier per unit, placed after the first token block: a wave's MFMAs never wait for it (they need only
//     their own A slot and B operands prefetched before the barrier), it only gates the DMA issue;
//   * the accumulator update acc += (ws * xs) * partial of block b runs between the MFMAs of block b + 1, plain
//     v_fmac_f32 (packed fp32 next to MFMAs is slower on this chip);
//   * ragged experts: the sort cuts an expert's rows into EQUAL tiles in 32-row steps (dispatch.hip, tile_gran), the
//     kernel runs exactly the 32-row pairs that hold rows (GLM-4.5-Air: 512 +- 22 rows = 3 x 176 instead of
//     256 + 256 + a 64-row stub that still streams every weight byte).
//
// Fixed VGPR map (per lane; 2 waves per SIMD -> 256 registers):
//   v0..v39    compiler (addresses, scalars-in-flight, epilogue temporaries)
//   v40..v55   B[2]      token operand of the current / next block (8 each)
//   v56..v71   P[2][2]   MFMA results of block parity x tile (4 each)
//   v72..v73   x[2]      token scale of the current / next block
//   v74..v77   f[2][2]   ws * xs of block parity x tile
//   v78        E8M0 1.0 x 4 (scale operand of the MFMA), v79 spare
//   v80..v127  A ring    slot s: tile 0 = v[80+16s .. +7], tile 1 = v[88+16s .. +7]
//   v128..v255 acc       block b, tile t: v[128 + 8b + 4t .. +3]
//
// vmcnt ledger (per wave, ops in issue order).  Iteration u issues, after its barrier, tokens(u+3) [4 pieces, + 1
// scale piece on waves 0..3 when u % 4 == 1] and, after its last MFMA, A(u+3) [4 loads].  Top of iteration u needs
// A(u): later ops are tokens(u+1) A(u+1) tokens(u+2) A(u+2) = 16 -> vmcnt(16).  Before the barrier of iteration u the
// wave's pieces of tokens(u+1) must have landed (they are read after the barrier by every wave: end-of-unit prefetch
// of block 0 of unit u+1): later ops are A(u+1) tokens(u+2) A(u+2) = 12 -> vmcnt(12).  The optional scale piece only
// makes a wait stricter, never wrong (it is older than everything the wait leaves outstanding).
#pragma once
#include "gemm_tiled.h"

namespace lkm {

namespace a8w {
constexpr int kNumVgpr = 40;
constexpr int kB = 40, kP = 56, kX = 72, kF = 74, kOne = 78, kA = 80, kAcc = 128;
constexpr int kStage = 256 * 128, kStages = 4;
constexpr int kScBase = kStage * kStages, kScBuf = 256 * 16;
constexpr int kLdsBytes = kScBase + 2 * kScBuf;          // 139 264
}  // namespace a8w

typedef __attribute__((ext_vector_type(4))) int a8w_i32x4;

#if defined(__HIP_DEVICE_COMPILE__)
// clobber lists: tell the compiler (for the kernel descriptor's register count) which fixed registers the asm owns
#define A8W_CLOB_B "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55"
#define A8W_CLOB_TOP "v255"

#define A8W_MFMA(P, A, B) \
    "v_mfma_scale_f32_16x16x128_f8f6f4 v[" P ":" P "+3], v[" A ":" A "+7], v[" B ":" B "+7], 0, v[%c[one]], v[%c[one]] op_sel_hi:[0,0,0]\n\t"
#define A8W_FMAC4(ACC, F, P)                            \
    "v_fmac_f32 v[" ACC "+0], v[" F "], v[" P "+0]\n\t" \
    "v_fmac_f32 v[" ACC "+1], v[" F "], v[" P "+1]\n\t" \
    "v_fmac_f32 v[" ACC "+2], v[" F "], v[" P "+2]\n\t" \
    "v_fmac_f32 v[" ACC "+3], v[" F "], v[" P "+3]\n\t"

// One 16-token block of one K unit: prefetch the next block's B operand and token scale, form this block's two scale
// products, multiply both weight tiles, and -- between the MFMAs -- add the PREVIOUS block's partial sums into its
// accumulators.  SLOT: A ring slot of the unit.  HASPREV: block b-1 of the same unit exists.  The prefetch address is
// the caller's: (same stage, block b+1) or (next stage, block 0).
template <int SLOT, int B, bool HASPREV, int BOFF, int XOFF>
__device__ __forceinline__ void a8w_block_t(int vblo, int vbhi, int vxs, int ws0, int ws1) {
    constexpr int par = B & 1, npar = par ^ 1;
    constexpr int BC = a8w::kB + par * 8, BN = a8w::kB + npar * 8;
    constexpr int PC = a8w::kP + par * 8, PP = a8w::kP + npar * 8;
    constexpr int XC = a8w::kX + par, XN = a8w::kX + npar;
    constexpr int FC = a8w::kF + par * 2, FP = a8w::kF + npar * 2;
    constexpr int A = a8w::kA + SLOT * 16;
    constexpr int ACC = a8w::kAcc + (HASPREV ? B - 1 : 0) * 8;
    if constexpr (HASPREV) {
        asm volatile(
            "ds_read_b128 v[%c[bn]:%c[bn]+3], %[vblo] offset:%c[boff]\n\t"
            "ds_read_b128 v[%c[bn]+4:%c[bn]+7], %[vbhi] offset:%c[boff]\n\t"
            "ds_read_b32 v[%c[xn]], %[vxs] offset:%c[xoff]\n\t"
            "v_mul_f32 v[%c[fc]], %[ws0], v[%c[xc]]\n\t"
            "v_mul_f32 v[%c[fc]+1], %[ws1], v[%c[xc]]\n\t"
            A8W_MFMA("%c[pc]", "%c[a]", "%c[bc]")
            A8W_FMAC4("%c[acc]", "%c[fp]", "%c[pp]")
            A8W_MFMA("%c[pc]+4", "%c[a]+8", "%c[bc]")
            A8W_FMAC4("%c[acc]+4", "%c[fp]+1", "%c[pp]+4")
            "s_waitcnt lgkmcnt(0)\n\t"
            :
            : [bn] "i"(BN), [bc] "i"(BC), [pc] "i"(PC), [pp] "i"(PP), [xc] "i"(XC), [xn] "i"(XN), [fc] "i"(FC),
              [fp] "i"(FP), [a] "i"(A), [acc] "i"(ACC), [one] "i"(a8w::kOne), [boff] "i"(BOFF), [xoff] "i"(XOFF),
              [vblo] "v"(vblo), [vbhi] "v"(vbhi), [vxs] "v"(vxs), [ws0] "s"(ws0), [ws1] "s"(ws1)
            : "memory", A8W_CLOB_B, A8W_CLOB_TOP);
    } else {
        asm volatile(
            "ds_read_b128 v[%c[bn]:%c[bn]+3], %[vblo] offset:%c[boff]\n\t"
            "ds_read_b128 v[%c[bn]+4:%c[bn]+7], %[vbhi] offset:%c[boff]\n\t"
            "ds_read_b32 v[%c[xn]], %[vxs] offset:%c[xoff]\n\t"
            "v_mul_f32 v[%c[fc]], %[ws0], v[%c[xc]]\n\t"
            "v_mul_f32 v[%c[fc]+1], %[ws1], v[%c[xc]]\n\t"
            A8W_MFMA("%c[pc]", "%c[a]", "%c[bc]")
            A8W_MFMA("%c[pc]+4", "%c[a]+8", "%c[bc]")
            "s_waitcnt lgkmcnt(0)\n\t"
            :
            : [bn] "i"(BN), [bc] "i"(BC), [pc] "i"(PC), [xc] "i"(XC), [xn] "i"(XN), [fc] "i"(FC), [a] "i"(A),
              [one] "i"(a8w::kOne), [boff] "i"(BOFF), [xoff] "i"(XOFF), [vblo] "v"(vblo), [vbhi] "v"(vbhi),
              [vxs] "v"(vxs), [ws0] "s"(ws0), [ws1] "s"(ws1)
            : "memory", A8W_CLOB_B, A8W_CLOB_TOP);
    }
}

// the accumulator update of a unit's LAST block (its MFMAs were the last two instructions of the matrix pipe: 16 wait
// states cover the 11 an 8-pass result needs before a VALU may read it)
template <int B>
__device__ __forceinline__ void a8w_flush() {
    constexpr int par = B & 1;
    constexpr int PP = a8w::kP + par * 8, FP = a8w::kF + par * 2, ACC = a8w::kAcc + B * 8;
    asm volatile("s_nop 15\n\t"
                 A8W_FMAC4("%c[acc]", "%c[fp]", "%c[pp]")
                 A8W_FMAC4("%c[acc]+4", "%c[fp]+1", "%c[pp]+4")
                 :
                 : [acc] "i"(ACC), [fp] "i"(FP), [pp] "i"(PP)
                 : "memory", A8W_CLOB_TOP);
}

// A(u) -> ring slot: two tiles x two 1-KiB loads, straight from the pre-shuffled image
template <int SLOT>
__device__ __forceinline__ void a8w_load_a(int voff, a8w_i32x4 rs, int s0, int s1) {
    asm volatile("s_nop 4\n\t"
                 "buffer_load_dwordx4 v[%c[a]:%c[a]+3], %[voff], %[rs], %[s0] offen\n\t"
                 "buffer_load_dwordx4 v[%c[a]+4:%c[a]+7], %[voff], %[rs], %[s0] offen offset:1024\n\t"
                 "buffer_load_dwordx4 v[%c[a]+8:%c[a]+11], %[voff], %[rs], %[s1] offen\n\t"
                 "buffer_load_dwordx4 v[%c[a]+12:%c[a]+15], %[voff], %[rs], %[s1] offen offset:1024\n\t"
                 :
                 : [a] "i"(a8w::kA + SLOT * 16), [voff] "v"(voff), [rs] "s"(rs), [s0] "s"(s0), [s1] "s"(s1)
                 : "memory", A8W_CLOB_TOP);
}

// one LDS-DMA piece: 64 lanes x SZ bytes from per-lane global offsets to lds_addr + lane * SZ
__device__ __forceinline__ void a8w_dma16(int lds_addr, int voff, a8w_i32x4 rs, int soff) {
    int keep;
    asm volatile("s_nop 4\n\t"
                 "s_mov_b32 %[keep], m0\n\t"
                 "s_mov_b32 m0, %[la]\n\t"
                 "s_nop 0\n\t"
                 "buffer_load_dwordx4 %[voff], %[rs], %[soff] offen lds\n\t"
                 "s_mov_b32 m0, %[keep]\n\t"
                 : [keep] "=&s"(keep)
                 : [la] "s"(lds_addr), [voff] "v"(voff), [rs] "s"(rs), [soff] "s"(soff)
                 : "memory");
}
#endif   // __HIP_DEVICE_COMPILE__

// DBG (development ablations, results wrong by construction): 1 = no loads / DMA inside the K loop (compute skeleton),
// 2 = no MFMA blocks (data movement + barriers only).  Tuning key "dbg"; bit 4 (serialised units) is a run-time flag.
template <int ADT, bool GATED, bool IS_G1, int DBG = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(a8w::kNumVgpr))) void gemm_prefill_a8w_kernel(GemmParams p) {
    static_assert(!GATED || IS_G1, "only GEMM1 is gated");
#if defined(__HIP_DEVICE_COMPILE__)
    using namespace a8w;
    typedef __attribute__((address_space(3))) char* LdsPtr;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int ti = blockIdx.y, bx = blockIdx.x;
    if (p.xcd_map) {     // XCD-aware 1-D mapping: see gemm_tiled_kernel and dispatch.hip (xcd_cut)
        const int RG = p.xcd_map;
        const int L = blockIdx.x, c = L & 7, sidx = L >> 3;
        const int first = p.meta[8 + c], n_c = p.meta[9 + c] - first;
        if (sidx >= n_c * RG) return;
        ti = first + sidx / RG;
        bx = sidx % RG;
    }
    const int n_tiles = p.meta[3];
    if (ti >= n_tiles) return;
    const int e = p.tile_e[ti], r0 = p.tile_r0[ti];
    const int m_e = p.counts[e], off_e = p.offsets[e];
    // rows of this token tile: up to the next tile of the same expert (the sort cuts an expert into equal tiles)
    int rows = m_e - r0;
    if (ti + 1 < n_tiles && p.tile_e[ti + 1] == e) rows = p.tile_r0[ti + 1] - r0;
    if (rows > 256) rows = 256;
    const int NQ = __builtin_amdgcn_readfirstlane((rows + 31) >> 5);       // 32-row pairs of blocks that hold rows
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int T_all = p.T_half * p.halves;
    const int U = __builtin_amdgcn_readfirstlane(p.U);
    const bool dbg_serial = __builtin_amdgcn_readfirstlane(p.dbg & 4) != 0;
    constexpr int TPH = GATED ? 8 : 16;                       // tiles per half taken by one workgroup
    const int tbase = bx * TPH;
    // the wave's two weight tiles: gated = gate tile w and up tile w of the same rows; else two adjacent tiles
    auto clampt = [&](int t) { return t < p.T_half ? t : p.T_half - 1; };
    const int gt0 = GATED ? clampt(tbase + wave) : clampt(tbase + 2 * wave);
    const int gt1 = GATED ? p.T_half + clampt(tbase + wave) : clampt(tbase + 2 * wave + 1);

    // ---- buffer resources (wave-uniform by construction: kernel arguments and e)
    auto make_rs = [&](const void* base, unsigned bytes) {
        const unsigned long long a = (unsigned long long)base;
        a8w_i32x4 r;
        r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
        r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
        r.z = __builtin_amdgcn_readfirstlane((int)bytes);
        r.w = 0x00020000;
        return r;
    };
    const size_t wbytes = (size_t)T_all * U * 2048;
    const a8w_i32x4 rs_w = make_rs((const char*)p.w + (size_t)e * wbytes, (unsigned)wbytes);
    const a8w_i32x4 rs_x = make_rs(p.x, (unsigned)((size_t)p.x_rows * (size_t)p.ldx));
    const a8w_i32x4 rs_xs = make_rs(p.xscale, (unsigned)((size_t)p.x_rows * (size_t)p.ld_xscale * 4));
    const int wub = __builtin_amdgcn_readfirstlane((int)(p.w_ustride * 16));
    const int a_s0 = __builtin_amdgcn_readfirstlane((int)(gt0 * p.w_tstride * 16));
    const int a_s1 = __builtin_amdgcn_readfirstlane((int)(gt1 * p.w_tstride * 16));
    const int a_voff = lane * 16;

    // ---- token pieces: piece q of this wave = token rows 64 q + 8 wave .. + 7, eight 16-byte slots each; the slot
    // permutation of gemm_tiled.h (x_swizzle) is applied on the SOURCE side, the LDS image stays lane-linear
    int bvoff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int pc = q * 512 + tid;
        const int row = pc >> 3, pslot = pc & 7;
        const int lslot = pslot ^ x_swizzle<128>(row);
        const int rr = row < rows ? r0 + row : r0;
        const int src_row = IS_G1 ? p.sorted_slot[off_e + rr] / p.top_k : off_e + rr;
        bvoff[q] = src_row * p.ldx + lslot * 16;              // fp8: bytes == elements; < 2 GiB (launcher)
    }
    // token-scale pieces (waves 0..3): lane = token 64 wave + lane, 16 bytes = the scales of four K units
    int svoff = 0;
    {
        const int row = (wave & 3) * 64 + lane;
        const int rr = row < rows ? r0 + row : r0;
        const int src_row = IS_G1 ? p.sorted_slot[off_e + rr] / p.top_k : off_e + rr;
        svoff = src_row * p.ld_xscale * 4;
    }
    const int n_grp = (U + 3) >> 2;
    // everything above that came from memory is consumed HERE (the compiler's own loads must not be waited for
    // inside the hand-counted loop)
    asm volatile("" ::"v"(bvoff[0]), "v"(bvoff[1]), "v"(bvoff[2]), "v"(bvoff[3]), "v"(svoff), "s"(NQ), "s"(a_s0), "s"(a_s1));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    const int lds0 = (int)(unsigned)(uintptr_t)(LdsPtr)lds;
    // B operand addresses: lane (g, j) reads token row 16 b + j, 16-byte slots g and 4 + g (swizzled), + 2048 b
    const int sw = x_swizzle<128>(j);
    const int vb_lo = lds0 + j * 128 + ((g ^ sw) * 16);
    const int vb_hi = lds0 + j * 128 + (((4 + g) ^ sw) * 16);
    const int vx0 = lds0 + kScBase + j * 16;                  // + 256 b + 4096 (group & 1) + 4 (u & 3)

    auto issue_tokens = [&](int u, int q) __attribute__((always_inline)) {
        const int uu = u < U ? u : U - 1;
        a8w_dma16(lds0 + (uu & 3) * kStage + q * 8192 + wave * 1024, bvoff[q], rs_x, uu * 128);
    };
    auto issue_scales = [&](int grp) __attribute__((always_inline)) {    // waves 0..3 only (caller)
        const int gg = grp < n_grp ? grp : n_grp - 1;
        a8w_dma16(lds0 + kScBase + (gg & 1) * kScBuf + wave * 1024, svoff, rs_xs, gg * 16);
    };
    auto issue_a = [&](int u, auto SLOT) __attribute__((always_inline)) {
        const int uu = u < U ? u : U - 1;
        a8w_load_a<decltype(SLOT)::v>(a_voff, rs_w, a_s0 + uu * wub, a_s1 + uu * wub);
    };

    // ---- weight-block scales: lane l <- scale of unit (chunk * 64 + l) of each tile (row 0 of the tile: one block
    // scale per 16-row tile, prefill_a8w_ok)
    float wsv0 = 0.f, wsv1 = 0.f;
    auto load_ws = [&](int chunk) __attribute__((always_inline)) {
        const int ul = chunk * 64 + lane;
        const int uc = ul < U ? ul : U - 1;
        const float* s0 = (const float*)p.s + ((size_t)e * T_all + gt0) * U * 16 + (size_t)uc * 16;
        const float* s1 = (const float*)p.s + ((size_t)e * T_all + gt1) * U * 16 + (size_t)uc * 16;
        asm volatile("global_load_dword %0, %2, off\n\t"
                     "global_load_dword %1, %3, off\n\t"
                     : "=&v"(wsv0), "=&v"(wsv1)
                     : "v"(s0), "v"(s1)
                     : "memory");
    };

    // ---- prologue
    asm volatile("v_mov_b32 v[%c0], 0x7f7f7f7f" ::"i"(kOne) : "memory", A8W_CLOB_TOP);
#pragma unroll
    for (int i = 0; i < 128; i += 4)
        asm volatile("v_mov_b32 v[%c0+0], 0\n\tv_mov_b32 v[%c0+1], 0\n\tv_mov_b32 v[%c0+2], 0\n\tv_mov_b32 v[%c0+3], 0" ::"i"(kAcc + i) : "memory");
    load_ws(0);
    if (wave < 4) issue_scales(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_tokens(0, q);
    issue_a(0, IC<0>{});
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_tokens(1, q);
    issue_a(1, IC<1>{});
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_tokens(2, q);
    issue_a(2, IC<2>{});
    asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" : "+v"(wsv0), "+v"(wsv1)::"memory");
    // block 0 of unit 0: B operand and token scale
    asm volatile("ds_read_b128 v[%c[b]:%c[b]+3], %[lo]\n\t"
                 "ds_read_b128 v[%c[b]+4:%c[b]+7], %[hi]\n\t"
                 "ds_read_b32 v[%c[x]], %[xs]\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 :
                 : [b] "i"(kB), [x] "i"(kX), [lo] "v"(vb_lo), [hi] "v"(vb_hi), [xs] "v"(vx0)
                 : "memory", A8W_CLOB_B);

    // ---- one K unit; SLOT = u % 3 (compile time: the A ring is register-indexed)
    auto unit = [&](int u, auto SLOTC) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(SLOTC)::v;
        if (u > 0 && (u & 63) == 0) {        // K > 8192: next 64 weight-block scales (drains the pipeline once)
            load_ws(u >> 6);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(wsv0), "+v"(wsv1)::"memory");
        }
        const int ws0 = __builtin_amdgcn_readlane(__builtin_bit_cast(int, wsv0), u & 63);
        const int ws1 = __builtin_amdgcn_readlane(__builtin_bit_cast(int, wsv1), u & 63);
        const int st = (u & 3) * kStage, stn = ((u + 1) & 3) * kStage;
        const int lo_c = vb_lo + st, hi_c = vb_hi + st, lo_n = vb_lo + stn, hi_n = vb_hi + stn;
        const int xs_c = vx0 + ((u >> 2) & 1) * kScBuf + (u & 3) * 4;
        const int xs_n = vx0 + (((u + 1) >> 2) & 1) * kScBuf + ((u + 1) & 3) * 4;
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                 // A(u) is in its slot
        // (opaque per unit: a loop-invariant q < NQ would be hoisted into sixteen SGPR-pair booleans, which spill)
        int nq = NQ;
        asm volatile("" : "+s"(nq));
        static_for<8>([&](auto QC) __attribute__((always_inline)) {
            constexpr int q = decltype(QC)::v;
            if (q < nq) {
                if constexpr (!(DBG & 2)) a8w_block_t<SLOT, 2 * q, (q > 0), (2 * q + 1) * 2048, (2 * q + 1) * 256>(lo_c, hi_c, xs_c, ws0, ws1);
                if constexpr (q == 0) {
                    // tokens(u+1) of THIS wave have landed; after the barrier every wave's have, and nobody reads
                    // stage (u+3) % 4 = (u-1) % 4 any more
                    asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
                    if constexpr (!(DBG & 1))
                        if ((u & 3) == 1 && wave < 4) issue_scales((u + 3) >> 2);
                }
                if (q + 1 < nq) {
                    if constexpr (!(DBG & 2)) a8w_block_t<SLOT, 2 * q + 1, true, (2 * q + 2) * 2048, (2 * q + 2) * 256>(lo_c, hi_c, xs_c, ws0, ws1);
                    if constexpr (q < 4 && !(DBG & 1)) issue_tokens(u + 3, q);
                } else {
                    // last block of the unit: prefetch block 0 of unit u+1 (next stage), then the rest of the unit's
                    // loads, then the block's own accumulator update
                    if constexpr (!(DBG & 2)) a8w_block_t<SLOT, 2 * q + 1, true, 0, 0>(lo_n, hi_n, xs_n, ws0, ws1);
                    if constexpr (!(DBG & 1)) {
                        static_for<4>([&](auto KC) __attribute__((always_inline)) {
                            if constexpr (decltype(KC)::v >= q) issue_tokens(u + 3, decltype(KC)::v);
                        });
                        issue_a(u + 3, IC<SLOT>{});
                    }
                    if constexpr (!(DBG & 2)) a8w_flush<2 * q + 1>();
                }
            }
        });
        if (dbg_serial) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (debug: serialised units)
    };
    for (int u = 0; u < U; u += 3) {
        unit(u, IC<0>{});
        if (u + 1 < U) unit(u + 1, IC<1>{});
        if (u + 2 < U) unit(u + 2, IC<2>{});
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // trailing (clamped) loads must not outlive the wave

    // ---- epilogue (D layout lane (g, j): rows tile*16 + g*4 + r, token column j of block b)
    static_for<16>([&](auto BC) __attribute__((always_inline)) {
        constexpr int b = decltype(BC)::v;
        if (b < 2 * NQ) {      // (uniform)
            const int rt = b * 16 + j;
            f32x4 c0, c1;
            asm volatile("v_mov_b32 %0, v[%c8+0]\n\tv_mov_b32 %1, v[%c8+1]\n\tv_mov_b32 %2, v[%c8+2]\n\tv_mov_b32 %3, v[%c8+3]\n\t"
                         "v_mov_b32 %4, v[%c8+4]\n\tv_mov_b32 %5, v[%c8+5]\n\tv_mov_b32 %6, v[%c8+6]\n\tv_mov_b32 %7, v[%c8+7]"
                         : "=&v"(c0.x), "=&v"(c0.y), "=&v"(c0.z), "=&v"(c0.w), "=&v"(c1.x), "=&v"(c1.y), "=&v"(c1.z), "=&v"(c1.w)
                         : "i"(kAcc + b * 8));
            if (rt < rows) {
                const size_t orow = (size_t)(off_e + r0 + rt);
                if constexpr (GATED) {
                    const int n = (tbase + wave) * 16 + g * 4;
                    if (tbase + wave < p.T_half && n < p.n_real) store_gemm1_frag<ADT, true>(p, c0, c1, orow, n);
                } else {
                    static_for<2>([&](auto TC) __attribute__((always_inline)) {
                        constexpr int t = decltype(TC)::v;
                        const int tl = tbase + 2 * wave + t;
                        const int n = tl * 16 + g * 4;
                        const f32x4 v = t ? c1 : c0;
                        if (tl < p.T_half && n < p.n_real) {
                            if constexpr (IS_G1) store_gemm1_frag<ADT, false>(p, v, v, orow, n);
                            else if (p.y_dt == LKM_DT_F32) store_gemm2_frag(p, v, 0, orow, n);
                            else {      // the reference's block-fp8 GEMM rounds its output to the activation dtype
                                unsigned short* o = (unsigned short*)p.out + orow * p.ldo + n;
                                if (n + 4 <= p.n_real) {
                                    *(u32x2*)o = u32x2{ActT<ADT>::pack2(v.x, v.y), ActT<ADT>::pack2(v.z, v.w)};
                                } else {
                                    if (n + 0 < p.n_real) o[0] = ActT<ADT>::from_f32(v.x);
                                    if (n + 1 < p.n_real) o[1] = ActT<ADT>::from_f32(v.y);
                                    if (n + 2 < p.n_real) o[2] = ActT<ADT>::from_f32(v.z);
                                    if (n + 3 < p.n_real) o[3] = ActT<ADT>::from_f32(v.w);
                                }
                            }
                        }
                    });
                }
            }
        }
    });
#else
    (void)p;
#endif
}

// usable when K is a whole number of 128-byte units, every 16-row weight tile has ONE block scale per K unit (groupN a
// multiple of 16), the weight image is tile-major or unit-major with 32-bit offsets, and the operand matrices fit the
// 2 GiB buffer windows; otherwise the plan stays on gemm_tiled_kernel (pick_cfg asks prefill_a8w_shape_ok first)
inline bool prefill_a8w_ok(const GemmParams& p) {
    return p.Kreal % 128 == 0 && p.tile_uniform_scale && p.U >= 1 &&
           (size_t)p.x_rows * (size_t)p.ldx < (size_t)0x7fffffff &&
           (size_t)p.x_rows * (size_t)p.ld_xscale * 4 < (size_t)0x7fffffff &&
           (size_t)p.T_half * p.halves * p.U * 2048 < (size_t)0x7fffffff;
}

template <int ADT, bool GATED, bool IS_G1, int DBG = 0>
static int launch_prefill_a8w_t(hipStream_t st, const GemmParams& p, int max_tiles) {
    constexpr size_t lds = (size_t)a8w::kLdsBytes;
    const int TPH = GATED ? 8 : 16;
    const int RG = ceil_div(p.T_half, TPH);
    dim3 grid(RG, max_tiles), block(512);
    GemmParams pp = p;
    if (p.xcd_map) {
        pp.xcd_map = RG;
        grid = dim3(8 * p.xcd_map * RG, 1);      // p.xcd_map = upper bound of the tiles in one XCD's run (host)
    }
    auto kern = gemm_prefill_a8w_kernel<ADT, GATED, IS_G1, DBG>;
    LKM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, grid, block, lds, st, pp);
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}

template <typename ADTC>
static bool launch_prefill_a8w_if(hipStream_t st, const LaunchCfg& cfg, const GemmParams& p, bool gated, bool is_g1,
                                  int max_tiles, int* rc, ADTC) {
    constexpr int ADT = ADTC::v;
    if (cfg.tiled != 256 || cfg.pf != 9) return false;
    if (!prefill_a8w_ok(p)) {
        set_error("fp8 W8A8 prefill kernel: shape not eligible (K %% 128, scale granularity or 2 GiB windows)");
        *rc = LKM_E_INVALID;
        return true;
    }
    if constexpr (ADT == LKM_DT_BF16) {     // ablation builds: gated GEMM1, bf16 activations only
        if (is_g1 && gated && (p.dbg & 3)) {
            *rc = (p.dbg & 1) ? launch_prefill_a8w_t<ADT, true, true, 1>(st, p, max_tiles) : launch_prefill_a8w_t<ADT, true, true, 2>(st, p, max_tiles);
            return true;
        }
    }
    if (is_g1) *rc = gated ? launch_prefill_a8w_t<ADT, true, true>(st, p, max_tiles) : launch_prefill_a8w_t<ADT, false, true>(st, p, max_tiles);
    else *rc = launch_prefill_a8w_t<ADT, false, false>(st, p, max_tiles);
    return true;
}

}  // namespace lkm
