// gemm_prefill_a8w.h -- per-expert grouped GEMMs for the prefill regime, fp8 weights x fp8 activations (W8A8, the
// in-tree block-fp8 semantics: fused_moe.py:298-610, native_w8a8_block_matmul tests/kernels/quant_utils.py:91-154),
// round-3 kernel: PERSISTENT workgroups (one per CU) walking a device-built list of items = 256 weight rows x up to 256
// tokens x all of K, on v_mfma_f32_16x16x128_f8f6f4 (one instruction per 16-token block, 16-row tile and 128-k block).
//
// What round 2's kernel (gemm_prefill_a8.h) was bound by: ONE 66 KiB LDS-DMA burst in flight per CU (both operands
// through two LDS buffers), waves of a SIMD released in phase by a per-unit barrier, operand reads scheduled by the
// compiler with lgkmcnt(0) in front of every step (profiles/r02_glm_a8_prefill.md).  This one:
//   * WEIGHTS never touch LDS.  The pre-shuffled image (lkm_common.h) makes a wave-wide buffer_load_dwordx4 the A
//     operand itself; a wave owns two 16-row tiles for the whole K loop and streams them HBM/L2 -> VGPR through a
//     3-slot register ring (2 units = 8 KiB per wave in flight while a third is multiplied);
//   * TOKENS go through a 4-stage LDS ring (32 KiB per 128-k unit), filled by LDS-DMA two to three units ahead, read
//     as B operands by all eight waves (a wave multiplies its two tiles by EVERY 16-token block of the tile);
//   * token scales ride in 16-byte pieces (4 units per token) into a 3 x 4 KiB LDS ring; an item's weight-block scales
//     and its row table (source row offsets) are fetched by LDS-DMA while the previous item runs and read from LDS;
//   * every load, DMA, wait and barrier of the K loop is issued from inline asm with hand-counted vmcnt (a load the
//     compiler can see next to an LDS-DMA turns each of its waits into vmcnt(0)); ~190 KiB in flight per CU;
//   * the K loop's operands live in a FIXED register map (v40..v255, below) that the compiler never allocates: every asm
//     statement of the loop names the registers it owns as clobbers and tools/scan_a8w_codegen.py checks the generated
//     code (amdgpu_num_vgpr is ignored below ~57 registers), so values that arrive asynchronously are never copied early;
//   * the pipeline does not drain between items: the last three unit positions of an item fetch the first three units
//     of the next one, whose record / row table / scales arrived earlier the same way;
//   * ONE barrier per unit, placed after the first token block: a wave's MFMAs never wait for it (they need only
//     their own A slot and B operands prefetched before the barrier), it only gates the DMA issue;
//   * the accumulator update acc += (ws * xs) * partial of block b runs between the MFMAs of block b + 1, plain
//     v_fmac_f32 (packed fp32 next to MFMAs is slower on this chip);
//   * the epilogue stores 16 bytes per lane: two lanes 16 apart exchange their 8-byte pieces of two tiles (GEMM2) or two
//     token blocks (gated GEMM1) with v_permlane16_swap_b32 (a8w_pair16) -- half the store instructions, 64 contiguous
//     bytes per row -- and the gated GEMM1 quantises its own output where the host asks for it (GemmParams::out_q: an item
//     is 256 rows x exactly one 1 x 128 group of the W8A8 intermediate; row maxima through ds_max_u32, one barrier per
//     half item, the quantiser's own arithmetic: bit-identical to the separate pass, tests/test_gpu_moe.py);
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
//   v78..v79   spare (v78 held the E8M0 unit scale while the MFMA was the MX-scaled form: two instructions per MFMA)
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
constexpr int kTbl = kScBase + 3 * kScBuf;               // two row tables (current / next item) of 2 KiB
constexpr int kStA = kTbl + 2 * 2048;                    // per-wave landing zones of the next item's record (8 x 256 B)
constexpr int kStB = kStA + 2048;                        // (unused since the item records carry the expert's row offset)
constexpr int kRaw = kStB + 2048;                        // 256 gathered sorted_slot entries of the next item
constexpr int kWs = kRaw + 1024;                         // weight-block scales: [2 items][8 waves][2 tiles][64 units] fp32
constexpr int kRmax = kWs + 2 * 8 * 512;                 // row maxima of the fused output quantisation: [2 halves][256 rows] u32
constexpr int kLdsBytes = kRmax + 2048;                  // 162 816: four token stages, three token-scale groups, two row
                                                         // tables, the landing zones, the weight-block scales
}  // namespace a8w

typedef __attribute__((ext_vector_type(4))) int a8w_i32x4;

#ifndef LKM_A8W_WIDE_EPI
#define LKM_A8W_WIDE_EPI 2      // 16-byte epilogue stores: 1 = GEMM2, 2 = GEMM2 and the gated GEMM1 (0: the 8-byte stores of the first version)
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// clobber lists: tell the compiler (for the kernel descriptor's register count) which fixed registers the asm owns
// clobber lists: every fixed register is named in the asm statements of the K loop, so that the compiler neither counts
// them free (its SGPR-spill lanes went to v69 -- an MFMA result register -- when only the B registers were listed) nor
// undercounts the kernel descriptor's register file
#define A8W_CLOB_B "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55"
#define A8W_CLOB_TOP "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"

// (the plain 128-k MFMA, both operands e4m3: the MX-scaled form with both scale operands 1.0 computes the same bits but is
// two instructions -- v_mfma_ld_scale_b32 + the MFMA -- 16 bytes and one more issue slot per MFMA)
#define A8W_MFMA(P, A, B) \
    "v_mfma_f32_16x16x128_f8f6f4 v[" P ":" P "+3], v[" A ":" A "+7], v[" B ":" B "+7], 0\n\t"
#define A8W_FMAC4(ACC, F, P)                            \
    "v_fmac_f32 v[" ACC "+0], v[" F "], v[" P "+0]\n\t" \
    "v_fmac_f32 v[" ACC "+1], v[" F "], v[" P "+1]\n\t" \
    "v_fmac_f32 v[" ACC "+2], v[" F "], v[" P "+2]\n\t" \
    "v_fmac_f32 v[" ACC "+3], v[" F "], v[" P "+3]\n\t"

#define A8W_MUL4(ACC, F, P)                            \
    "v_mul_f32 v[" ACC "+0], v[" F "], v[" P "+0]\n\t" \
    "v_mul_f32 v[" ACC "+1], v[" F "], v[" P "+1]\n\t" \
    "v_mul_f32 v[" ACC "+2], v[" F "], v[" P "+2]\n\t" \
    "v_mul_f32 v[" ACC "+3], v[" F "], v[" P "+3]\n\t"
// accumulator update: += in general, = in an item's first K unit (OPT bit 0: the accumulators are never zeroed)
#define A8W_ACC4(ACC, F, P) ".if %c[first]\n\t" A8W_MUL4(ACC, F, P) ".else\n\t" A8W_FMAC4(ACC, F, P) ".endif\n\t"

// One 16-token block of one K unit: prefetch the next block's B operand and token scale, form this block's two scale
// products, multiply both weight tiles, and -- between the MFMAs -- add the PREVIOUS block's partial sums into its
// accumulators.  SLOT: A ring slot of the unit.  HASPREV: block b-1 of the same unit exists.  The prefetch address is
// the caller's: (same stage, block b+1) or (next stage, block 0).
// DBG (ablations, assembler conditionals): 8 = no LDS reads, 16 = no VALU (scale products, accumulator updates),
// 32 = no MFMA, 128 = the block does not wait for its LDS reads
#define A8W_IF(bit) ".if (%c[dbg] & " #bit ") == 0\n\t"
#define A8W_FI ".endif\n\t"
// OPT: bit 0 = the unit is the item's first (accumulators are assigned, not added to), bit 1 = both weight tiles share one
// block scale (GEMM2 and the plain GEMM1: two adjacent 16-row tiles of one 128-row scale block) -> one scale product
template <int SLOT, int B, bool HASPREV, int BOFF, int XOFF, int DBG, int OPT>
__device__ __forceinline__ void a8w_block_t(int vblo, int vbhi, int vxs, int ws0, int ws1) {
    constexpr int par = B & 1, npar = par ^ 1;
    constexpr int BC = a8w::kB + par * 8, BN = a8w::kB + npar * 8;
    constexpr int PC = a8w::kP + par * 8, PP = a8w::kP + npar * 8;
    constexpr int XC = a8w::kX + par, XN = a8w::kX + npar;
    constexpr int FC = a8w::kF + par * 2, FP = a8w::kF + npar * 2;
    constexpr int A = a8w::kA + SLOT * 16;
    constexpr int ACC = a8w::kAcc + (HASPREV ? B - 1 : 0) * 8;
    asm volatile(
        A8W_IF(8)
        "ds_read_b128 v[%c[bn]:%c[bn]+3], %[vblo] offset:%c[boff]\n\t"
        "ds_read_b128 v[%c[bn]+4:%c[bn]+7], %[vbhi] offset:%c[boff]\n\t"
        "ds_read_b32 v[%c[xn]], %[vxs] offset:%c[xoff]\n\t"
        A8W_FI
        A8W_IF(16)
        "v_mul_f32 v[%c[fc]], %[ws0], v[%c[xc]]\n\t"
        ".if %c[same] == 0\n\t"
        "v_mul_f32 v[%c[fc]+1], %[ws1], v[%c[xc]]\n\t"
        ".endif\n\t"
        A8W_FI
        A8W_IF(32)
        A8W_MFMA("%c[pc]", "%c[a]", "%c[bc]")
        A8W_FI
        ".if %c[hasprev] && ((%c[dbg] & 16) == 0)\n\t"
        A8W_ACC4("%c[acc]", "%c[fp]", "%c[pp]")
        A8W_FI
        A8W_IF(32)
        A8W_MFMA("%c[pc]+4", "%c[a]+8", "%c[bc]")
        A8W_FI

        ".if %c[hasprev] && ((%c[dbg] & 16) == 0)\n\t"
        A8W_ACC4("%c[acc]+4", "%c[fp1]", "%c[pp]+4")
        A8W_FI
        A8W_IF(128)
        "s_waitcnt lgkmcnt(0)\n\t"
        A8W_FI
        :
        : [bn] "i"(BN), [bc] "i"(BC), [pc] "i"(PC), [pp] "i"(PP), [xc] "i"(XC), [xn] "i"(XN), [fc] "i"(FC),
          [fp] "i"(FP), [fp1] "i"(FP + ((OPT & 2) ? 0 : 1)), [a] "i"(A), [acc] "i"(ACC), [boff] "i"(BOFF), [xoff] "i"(XOFF),
          [hasprev] "i"(HASPREV ? 1 : 0), [dbg] "i"(DBG), [first] "i"(OPT & 1), [same] "i"((OPT >> 1) & 1),
          [vblo] "v"(vblo), [vbhi] "v"(vbhi), [vxs] "v"(vxs), [ws0] "s"(ws0), [ws1] "s"(ws1)
        : "memory", A8W_CLOB_B, A8W_CLOB_TOP);
}

// Block 0 of a unit (no previous block): as above, plus what the unit's OTHER work needs, fetched under the block's MFMAs
// and complete at the asm's end (lgkmcnt(0)), all from LDS (a scalar load here would park every later lgkmcnt(0) behind
// the scalar cache): the weight-block scales of the NEXT unit and the source offsets of the unit's four token pieces
// and one scale piece (row table).
template <int SLOT, int DBG, int OPT>
__device__ __forceinline__ void a8w_block0(int vblo, int vbhi, int vxs, int ws0, int ws1, int wsaddr, int taddr,
                                           int& ws0n, int& ws1n, int (&t)[4]) {
    constexpr int A = a8w::kA + SLOT * 16;
    asm volatile(
        A8W_IF(8)
        "ds_read_b128 v[%c[bn]:%c[bn]+3], %[vblo] offset:2048\n\t"
        "ds_read_b128 v[%c[bn]+4:%c[bn]+7], %[vbhi] offset:2048\n\t"
        "ds_read_b32 v[%c[xn]], %[vxs] offset:256\n\t"
        A8W_FI
        "ds_read_b32 %[ws0n], %[wsaddr]\n\t"
        "ds_read_b32 %[ws1n], %[wsaddr] offset:256\n\t"
        "ds_read_b32 %[t0], %[taddr]\n\t"
        "ds_read_b32 %[t1], %[taddr] offset:256\n\t"
        "ds_read_b32 %[t2], %[taddr] offset:512\n\t"
        "ds_read_b32 %[t3], %[taddr] offset:768\n\t"
        A8W_IF(16)
        "v_mul_f32 v[%c[fc]], %[ws0], v[%c[xc]]\n\t"
        ".if %c[same] == 0\n\t"
        "v_mul_f32 v[%c[fc]+1], %[ws1], v[%c[xc]]\n\t"
        ".endif\n\t"
        A8W_FI
        A8W_IF(32)
        A8W_MFMA("%c[pc]", "%c[a]", "%c[bc]")
        A8W_MFMA("%c[pc]+4", "%c[a]+8", "%c[bc]")
        A8W_FI
        "s_waitcnt lgkmcnt(0)\n\t"
        : [ws0n] "=&v"(ws0n), [ws1n] "=&v"(ws1n), [t0] "=&v"(t[0]), [t1] "=&v"(t[1]), [t2] "=&v"(t[2]), [t3] "=&v"(t[3])
        : [bn] "i"(a8w::kB + 8), [bc] "i"(a8w::kB), [pc] "i"(a8w::kP), [xc] "i"(a8w::kX), [xn] "i"(a8w::kX + 1),
          [fc] "i"(a8w::kF), [a] "i"(A), [dbg] "i"(DBG), [same] "i"((OPT >> 1) & 1), [vblo] "v"(vblo), [vbhi] "v"(vbhi),
          [vxs] "v"(vxs), [ws0] "s"(ws0), [ws1] "s"(ws1), [wsaddr] "v"(wsaddr), [taddr] "v"(taddr)
        : "memory", A8W_CLOB_B, A8W_CLOB_TOP);
}

// the accumulator update of a unit's LAST block (its MFMAs were the last two instructions of the matrix pipe: 16 wait
// states cover the 11 an 8-pass result needs before a VALU may read it)
template <int B, int DBG, int OPT>
__device__ __forceinline__ void a8w_flush() {
    if constexpr (DBG & 16) return;
    constexpr int par = B & 1;
    constexpr int PP = a8w::kP + par * 8, FP = a8w::kF + par * 2, ACC = a8w::kAcc + B * 8;
    asm volatile("s_nop 15\n\t"
                 A8W_ACC4("%c[acc]", "%c[fp]", "%c[pp]")
                 A8W_ACC4("%c[acc]+4", "%c[fp1]", "%c[pp]+4")
                 :
                 : [acc] "i"(ACC), [fp] "i"(FP), [fp1] "i"(FP + ((OPT & 2) ? 0 : 1)), [pp] "i"(PP), [first] "i"(OPT & 1)
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
// 64 lanes x 4 bytes from per-lane 64-bit addresses to lds_addr + lane * 4 (item metadata: whatever is in flight between
// two points of the K loop lives in LDS, never in a register the compiler could copy before the data has landed)
__device__ __forceinline__ void a8w_dma4_flat(int lds_addr, const void* addr) {
    int keep;
    asm volatile("s_mov_b32 %[keep], m0\n\t"
                 "s_mov_b32 m0, %[la]\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dword %[ad], off\n\t"
                 "s_mov_b32 m0, %[keep]\n\t"
                 : [keep] "=&s"(keep)
                 : [la] "s"(lds_addr), [ad] "v"(addr)
                 : "memory");
}
#endif   // __HIP_DEVICE_COMPILE__

// silu(g) * u with the hardware exp / reciprocal (v_exp_f32, v_rcp_f32: ~1 ulp each) instead of the shared
// deterministic expf + IEEE division of store_gemm1_frag: the epilogue of a 256 x 256 tile is 32 fragments per lane, and
// at ~50 VALU per element it cost 8 us of every 60-us tile.  Same rounding points as the block-fp8 reference
// (activation_kernels.cu:57-75: gate and up rounded to the activation dtype by the GEMM, T(silu_f32(g)) * u); only
// the last bit of the fp32 sigmoid differs, far inside the operator's tolerance (tests/kernels/moe/test_block_fp8.py).
template <int ADT>
__device__ __forceinline__ u32x2 a8w_silu_mul(const GemmParams& p, const f32x4& gate, const f32x4& upv) {
    float v[4];
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
    return u32x2{ActT<ADT>::pack2(v[0], v[1]), ActT<ADT>::pack2(v[2], v[3])};
}
template <int ADT>
__device__ __forceinline__ void a8w_store_silu_mul(const GemmParams& p, const f32x4& gate, const f32x4& upv, size_t out_row, int n) {
    const u32x2 o2 = a8w_silu_mul<ADT>(p, gate, upv);
    unsigned short* o = (unsigned short*)p.out + out_row * p.ldo + n;
    if (n + 4 <= p.n_real) {
        *(u32x2*)o = o2;
    } else {
        const unsigned short h[4] = {(unsigned short)o2.x, (unsigned short)(o2.x >> 16), (unsigned short)o2.y, (unsigned short)(o2.y >> 16)};
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n + r < p.n_real) o[r] = h[r];
    }
}

// Two lanes 16 apart hold neighbouring 8-byte pieces of an output row: lane (g, j) columns 4 g .. 4 g + 3 of row j.
// v_permlane16_swap_b32 (odd 16-lane rows of the first operand <-> even rows of the second) turns two such pieces per
// lane -- `a` and `b`, of two different rows (GEMM1: two token blocks) or two different tiles (GEMM2) -- into ONE 16-byte
// piece per lane: even g keeps its `a` and receives g + 1's `a` (columns 4 g .. 4 g + 7 of a's row / tile), odd g
// receives g - 1's `b` and keeps its own (columns 4 (g - 1) .. 4 g + 3 of b's).  Half the store instructions, 64
// contiguous bytes per row and instruction instead of 32: the epilogue is store-issue bound.
__device__ __forceinline__ u32x4 a8w_pair16(const u32x2& a, const u32x2& b) {
    const u32x2 lo = __builtin_amdgcn_permlane16_swap(a.x, b.x, false, false);
    const u32x2 hi = __builtin_amdgcn_permlane16_swap(a.y, b.y, false, false);
    return u32x4{lo.x, hi.x, lo.y, hi.y};
}

// DBG (development ablations, results wrong by construction): 1 = no loads / DMA inside the K loop (compute skeleton),
// 2 = no token blocks (data movement + barriers only), 8 / 16 / 32 = blocks without LDS reads / VALU / MFMA, 64 = no
// per-unit barrier, 512 = the epilogue computes nothing and stores nothing.  Tuning key "dbg"; bit 4 (serialised units) is a run-time flag.  The ablation kernels live in
// their own translation unit (gemm_a8w_dbg.hip).
//
// PERSISTENT: the grid is one workgroup per CU; a workgroup walks its share of the (token tile, weight row group) items
// and the load pipeline runs THROUGH the item boundary (the last three units of an item fetch the first three of the
// next one), so an item pays its epilogue and nothing else -- no workgroup launch, no metadata round trips, no pipeline
// refill (measured on GLM-4.5-Air: 325 of 1030 us were prologue + epilogue + launch of the 3 590 one-item workgroups,
// profiles/r03_a8w_ablations_v1.log "dbg=3").  Per-item state is five scalars; what the lanes need of an item -- the
// source row of each of its 256 token rows, pre-multiplied by the two row strides -- sits in a 2-KiB LDS table per item
// (current / next), written once (256 gathered sorted_slot entries) and read back five words per lane and unit.
template <int ADT, bool GATED, bool IS_G1, int DBG = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(a8w::kNumVgpr))) void gemm_prefill_a8w_kernel(GemmParams p) {
    static_assert(!GATED || IS_G1, "only GEMM1 is gated");
#if defined(__HIP_DEVICE_COMPILE__)
    using namespace a8w;
    typedef __attribute__((address_space(3))) char* LdsPtr;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int T_half = __builtin_amdgcn_readfirstlane(p.T_half);
    const int T_all = T_half * p.halves;
    const int U = __builtin_amdgcn_readfirstlane(p.U);
    const int n_grp = (U + 3) >> 2;
    const bool dbg_serial = __builtin_amdgcn_readfirstlane(p.dbg & 4) != 0;
    constexpr int TPH = GATED ? 8 : 16;                       // tiles per half taken by one item
    const int RG = __builtin_amdgcn_readfirstlane((T_half + TPH - 1) / TPH);
    const int n_tiles = __builtin_amdgcn_readfirstlane(p.meta[3]);
    // ---- this workgroup's items: positions s = w, w + stride, ... of its part of the item list (p.items, built by
    // dispatch.hip build_items_kernel: {expert, first output row, rows, row group} per item, an expert's token tiles of one
    // row group adjacent).  With the XCD-aware runs (dispatch.hip xcd_cut) a part is one XCD's run of tiles x the row
    // groups, and its workgroups the ones the dispatcher places there (block b -> XCD b % 8: a speed assumption, never a
    // correctness one)
    int first = 0, n_c = n_tiles, w = blockIdx.x, stride = gridDim.x;
    if (p.xcd_map) {
        const int c = blockIdx.x & 7;
        first = p.meta[8 + c];
        n_c = p.meta[9 + c] - first;
        w = blockIdx.x >> 3;
        stride = gridDim.x >> 3;
    }
    first = __builtin_amdgcn_readfirstlane(first);
    const int n_items = __builtin_amdgcn_readfirstlane(n_c * RG);
    w = __builtin_amdgcn_readfirstlane(w);
    stride = __builtin_amdgcn_readfirstlane(stride);
    if (w >= n_items) return;

    const size_t wbytes = (size_t)T_all * U * 2048;
    const int wub = __builtin_amdgcn_readfirstlane((int)(p.w_ustride * 16));
    const int wtb = __builtin_amdgcn_readfirstlane((int)(p.w_tstride * 16));
    // ---- item descriptors: wave-uniform scalars only
    struct Meta {
        int valid, bx, e, orow0, rows;
    };
    const int lds0 = (int)(unsigned)(uintptr_t)(LdsPtr)lds;
    const int item0 = __builtin_amdgcn_readfirstlane(first * RG);
    // Metadata of the item two ahead: its 16-byte record by 4-byte LDS-DMA into this wave's landing zone at an item switch
    // (behind the asm blocks' "memory" clobbers and the epilogue's stores the compiler cannot keep such loads scalar, and a
    // load it issues itself is waited for with vmcnt(0): the whole pipeline; a register destination written by an asm
    // load is copied around by the register allocator before the data lands), read back with a plain LDS read at the
    // next switch, a whole item later.
    auto meta_a = [&](int s) __attribute__((always_inline)) {
        Meta m;
        m.valid = s < n_items;
        const int ss = m.valid ? s : 0;
        int ln = lane;
        asm volatile("" : "+v"(ln));            // (per-item address arithmetic must not be hoisted out of the item loop: registers)
        a8w_dma4_flat(lds0 + kStA + wave * 256, p.items + (size_t)(item0 + ss) * 4 + (ln & 3));
        m.bx = m.e = m.orow0 = m.rows = 0;
        return m;
    };
    auto meta_a_done = [&](Meta& m) __attribute__((always_inline)) {      // (landed: caller's ledger)
        a8w_i32x4 v;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(lds0 + kStA + wave * 256) : "memory");
        m.e = __builtin_amdgcn_readfirstlane(v.x);
        m.orow0 = __builtin_amdgcn_readfirstlane(v.y);
        m.rows = __builtin_amdgcn_readfirstlane(v.z);
        m.bx = __builtin_amdgcn_readfirstlane(v.w);
    };
    struct Item {
        int nq;        // 32-row pairs of token blocks that hold rows; 0 = no such item
        int rows, orow0, tbase, e;
        int wlo, whi;  // the expert's weight image (buffer resource words 0 / 1)
        int as0, as1;  // byte offsets of this wave's two tiles inside it
    };
    auto make_item = [&](const Meta& m) __attribute__((always_inline)) {
        Item it;
        const int rows = m.rows;
        it.rows = rows;
        it.nq = m.valid ? (rows + 31) >> 5 : 0;
        it.orow0 = m.orow0;
        it.tbase = m.bx * TPH;
        it.e = m.e;
        const unsigned long long wa = (unsigned long long)((const char*)p.w + (size_t)m.e * wbytes);
        it.wlo = __builtin_amdgcn_readfirstlane((int)(unsigned)wa);
        it.whi = __builtin_amdgcn_readfirstlane((int)((unsigned)(wa >> 32) & 0xffffu));
        const int t0 = GATED ? it.tbase + wave : it.tbase + 2 * wave;
        const int t1 = GATED ? it.tbase + wave : it.tbase + 2 * wave + 1;
        it.as0 = (t0 < T_half ? t0 : T_half - 1) * wtb;
        it.as1 = ((GATED ? T_half : 0) + (t1 < T_half ? t1 : T_half - 1)) * wtb;
        return it;
    };
    // the wave's two weight tiles of an item: gated = gate tile w and up tile w of the same rows; else two adjacent tiles
    auto tile0 = [&](const Item& it) __attribute__((always_inline)) {
        const int t = GATED ? it.tbase + wave : it.tbase + 2 * wave;
        return t < T_half ? t : T_half - 1;
    };
    auto tile1 = [&](const Item& it) __attribute__((always_inline)) {
        const int t = GATED ? it.tbase + wave : it.tbase + 2 * wave + 1;
        return (GATED ? T_half : 0) + (t < T_half ? t : T_half - 1);
    };
    auto make_rs = [&](const void* base, unsigned bytes) __attribute__((always_inline)) {
        const unsigned long long a = (unsigned long long)base;
        a8w_i32x4 r;
        r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
        r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
        r.z = __builtin_amdgcn_readfirstlane((int)bytes);
        r.w = 0x00020000;
        return r;
    };
    const a8w_i32x4 rs_x = make_rs(p.x, (unsigned)((size_t)p.x_rows * (size_t)p.ldx));
    const a8w_i32x4 rs_xs = make_rs(p.xscale, (unsigned)((size_t)p.x_rows * (size_t)p.ld_xscale * 4));
    const int ldx = p.ldx, ldxs4 = p.ld_xscale * 4, top_k = p.top_k;
    // fused 1 x 128 fp8 quantisation of the gated GEMM1's output (the W8A8 intermediate): p.out_q / p.out_qs instead of p.out
    const bool fq = GATED && __builtin_amdgcn_readfirstlane(p.out_q != nullptr ? 1 : 0) != 0;
    const float rcp_top_k = p.rcp_top_k;

    // ---- an item's row table (LDS, kTbl + 2048 * buffer): [256] source row * ldx, then [256] source row * ld_xscale * 4.
    // Waves 0..3 gather one sorted_slot entry per lane (GEMM1; GEMM2 rows are their own sources) into the landing zone ...
    auto gather_item = [&](const Item& it) __attribute__((always_inline)) {
        if constexpr (IS_G1) {
            if (wave < 4) {
                int ln = lane;
                asm volatile("" : "+v"(ln));
                const int r = wave * 64 + ln;
                a8w_dma4_flat(lds0 + kRaw + wave * 256, p.sorted_slot + it.orow0 + (r < it.rows ? r : 0));
            }
        }
    };
    // ... and, once the entries have landed (two barriers later), store the two pre-multiplied offsets of their rows
    auto store_table = [&](int buf, const Item& it) __attribute__((always_inline)) {
        if (wave < 4) {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int r = wave * 64 + ln;
            int src;
            if constexpr (IS_G1) {
                int raw;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(raw) : "v"(lds0 + kRaw + r * 4) : "memory");
                // slot -> token: raw / top_k through a SCALAR float reciprocal (the compiler's integer division keeps its
                // magic number in a vector register across the whole kernel, and the K loop has none to spare); exact
                // for slot numbers below 2^22 (prefill_a8w_ok): the quotient is off by at most one before the correction
                src = (int)((float)raw * rcp_top_k);
                const int rem = raw - src * top_k;
                src += (rem >= top_k ? 1 : 0) - (rem < 0 ? 1 : 0);
            } else {
                src = it.orow0 + (r < it.rows ? r : 0);
            }
            const int a = lds0 + kTbl + buf * 2048 + r * 4;
            asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:1024" ::"v"(a), "v"(src * ldx), "v"(src * ldxs4) : "memory");
        }
    };
    // Lane-derived LDS addresses (loop invariants; the compiler's registers stay below the fixed map in the K loop and
    // below v78 in the epilogue: tools/scan_a8w_codegen.py).
    //   B operand: lane (g, j) reads token row 16 b + j, 16-byte slots g and 4 + g (swizzled), + 2048 b
    //   token pieces: piece q of this wave = token rows 64 q + 8 wave .. + 7, eight 16-byte slots each; the slot permutation
    //     of gemm_tiled.h (x_swizzle, period 16 in the row: the same for every q) is applied on the SOURCE side, the LDS
    //     image stays lane-linear.  Scale pieces (waves 0..3): lane = token 64 wave + lane, 16 bytes = four K units
    struct LaneAddr {
        int vb_lo, vb_hi, vx0, plslot, tb_tok;
    };
    auto lane_addr = [&](int t) __attribute__((always_inline)) {
        const int ln = t & 63, gg = ln >> 4, jj = ln & 15;
        const int sw = x_swizzle<128>(jj);
        LaneAddr a;
        a.vb_lo = lds0 + jj * 128 + ((gg ^ sw) * 16);
        a.vb_hi = lds0 + jj * 128 + (((4 + gg) ^ sw) * 16);
        a.vx0 = lds0 + kScBase + jj * 16;                          // + 256 b + kScBuf * ring slot + 4 (u & 3)
        a.plslot = ((t & 7) ^ x_swizzle<128>(t >> 3)) * 16;
        a.tb_tok = lds0 + kTbl + (t >> 3) * 4;                     // + 256 q + 2048 * buffer
        return a;
    };
    // an item's weight-block scales (row 0 of each tile: one block scale per 16-row tile and K unit, prefill_a8w_ok;
    // lane l <- unit l) into this wave's landing zone of item buffer `buf`
    auto fetch_ws = [&](const Item& it, int buf) __attribute__((always_inline)) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int uc = ln < U ? ln : U - 1;
        const char* sb = (const char*)p.s + ((size_t)it.e * T_all * U + uc) * 64;
        a8w_dma4_flat(lds0 + kWs + buf * 4096 + wave * 512, sb + (size_t)tile0(it) * U * 64);
        a8w_dma4_flat(lds0 + kWs + buf * 4096 + wave * 512 + 256, sb + (size_t)tile1(it) * U * 64);
    };

    Item cur, nxt;
    Meta pend;
    {
        // (three items of metadata through the one landing zone, each waited for: the pipeline has not started)
        Meta m0 = meta_a(w);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        meta_a_done(m0);
        cur = make_item(m0);
        gather_item(cur);
        fetch_ws(cur, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        store_table(0, cur);
        Meta m1 = meta_a(w + stride);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        meta_a_done(m1);
        nxt = make_item(m1);
        gather_item(nxt);
        fetch_ws(nxt, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        store_table(1, nxt);
        pend = meta_a(w + 2 * stride);
        if constexpr (GATED) asm volatile("ds_write_b32 %0, %1" ::"v"(lds0 + kRmax + tid * 4), "v"(0) : "memory");      // (row maxima: 512 words)
        // everything above that came from memory is consumed HERE (the compiler's own loads must not be waited for
        // inside the hand-counted loop)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    int it_next = w + 3 * stride;          // the item after `pend`
    int tbuf = 0;                          // row table / weight scales of the current item (the next item's: tbuf ^ 1)
    int stoff = 0;                         // LDS stage of the current unit (byte offset; + kStage mod 4 stages per unit)
    int xsl = 0;                           // token-scale ring slot of the current unit's group of four
    int xsl_nxt0 = n_grp % 3;              // ... of the next item's first group
    // An item occupies Up = 3 * ceil(U / 3) unit POSITIONS: the A ring has three register-indexed slots, so the unit
    // code exists once per slot and an item must start at slot 0; the Up - U positions behind the last K unit are NULL
    // units (no multiplication; they keep the barrier cadence and issue their share of the next item's loads).
    const int Up = ((U + 2) / 3) * 3;
    const LaneAddr la = lane_addr(tid);
    auto next3 = [](int v) __attribute__((always_inline)) { return v == 2 ? 0 : v + 1; };

    // the scale piece's source offset (waves 0..3; read when a piece is due: once per four units)
    auto scale_src = [&](int tsel) __attribute__((always_inline)) {
        int ln = lane, ts;
        asm volatile("" : "+v"(ln));
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(ts) : "v"(lds0 + kTbl + 1024 + tsel + ((wave & 3) * 64 + ln) * 4) : "memory");
        return ts;
    };
    // loads of the unit three positions ahead
    auto issue_tokens = [&](int q, int toff, int dma_base, int soff) __attribute__((always_inline)) {
        a8w_dma16(dma_base + q * 8192, toff + la.plslot, rs_x, soff);
    };
    auto issue_a = [&](auto SLOT, const Item& it, int uu) __attribute__((always_inline)) {
        a8w_i32x4 rs_w;
        rs_w.x = it.wlo;
        rs_w.y = it.whi;
        rs_w.z = (int)(unsigned)wbytes;
        rs_w.w = 0x00020000;
        a8w_load_a<decltype(SLOT)::v>(lane * 16, rs_w, it.as0 + uu * wub, it.as1 + uu * wub);
    };

    // ---- prologue: the pipeline's first three units
    int ws0, ws1;          // the current unit's two weight-block scales (fp32 bits)
    int st_cls = 0;        // store instructions of the last epilogue: 2 = at least 16, 1 = at least 8 (the ledger's waits of the next three units)
    {
        int t[4], ts, w0, w1;
        asm volatile("ds_read_b32 %0, %7\n\tds_read_b32 %1, %7 offset:256\n\tds_read_b32 %2, %7 offset:512\n\t"
                     "ds_read_b32 %3, %7 offset:768\n\tds_read_b32 %4, %8\n\tds_read_b32 %5, %9\n\t"
                     "ds_read_b32 %6, %9 offset:256\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(ts), "=&v"(w0), "=&v"(w1)
                     : "v"(la.tb_tok), "v"(lds0 + kTbl + 1024 + ((wave & 3) * 64 + lane) * 4), "v"(lds0 + kWs + wave * 512)
                     : "memory");
        ws0 = __builtin_amdgcn_readfirstlane(w0);
        ws1 = __builtin_amdgcn_readfirstlane(w1);
        if (wave < 4) a8w_dma16(lds0 + kScBase + wave * 1024, ts, rs_xs, 0);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
            for (int q = 0; q < 4; ++q) issue_tokens(q, t[q], lds0 + k * kStage + wave * 1024, k * 128);
            if (k == 0) issue_a(IC<0>{}, cur, 0);
            if (k == 1) issue_a(IC<1>{}, cur, 1);
            if (k == 2) issue_a(IC<2>{}, cur, 2);
        }
        asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
        // block 0 of unit 0: B operand and token scale
        asm volatile("ds_read_b128 v[%c[b]:%c[b]+3], %[lo]\n\t"
                     "ds_read_b128 v[%c[b]+4:%c[b]+7], %[hi]\n\t"
                     "ds_read_b32 v[%c[x]], %[xs]\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     :
                     : [b] "i"(kB), [x] "i"(kX), [lo] "v"(la.vb_lo), [hi] "v"(la.vb_hi), [xs] "v"(la.vx0)
                     : "memory", A8W_CLOB_B);
    }

    // ---- one unit position.  SLOT = position % 3 (the A ring is register-indexed).  KIND: 0 = one of the item's first
    // three (waits that step over the previous epilogue's stores; the hook that finishes the next item's tables), 1 =
    // middle (everything it touches belongs to the current item: no selects), 2 = one of the last three (its loads are the
    // NEXT item's first units; the position may be a null unit).
    // DBG 256 (development): cycles this wave spends at the unit's three synchronisation points -- the wait for its weight
    // loads, the wait for its token pieces, the barrier -- summed over the kernel; one wave prints them at the end
    unsigned t_top = 0, t_mid = 0, t_bar = 0;
    auto now = [&]() __attribute__((always_inline)) {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        return (unsigned)t;
    };
    unsigned t_start = 0;
    if constexpr (DBG & 256) t_start = now();
    auto unit = [&](auto SLOTC, auto KINDC, int pp) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(SLOTC)::v, KIND = decltype(KINDC)::v;
        // (a tail position is a kernel-wide constant, Up - 3 + SLOT: opaque, or the compiler hoists the lane addresses that
        // depend on it out of the item loop into vector registers the K loop does not have)
        if constexpr (KIND == 2) asm volatile("" : "+s"(pp));
        // the item's first unit assigns the accumulators (they are never zeroed); two tiles of one scale block share the product
        constexpr int OPT = ((KIND == 0 && SLOT == 0) ? 1 : 0) | (GATED ? 0 : 2);
        const int stn = (stoff + kStage) & (kStages * kStage - 1);
        const int lo_c = la.vb_lo + stoff, hi_c = la.vb_hi + stoff, lo_n = la.vb_lo + stn, hi_n = la.vb_hi + stn;
        const int xs_c = la.vx0 + xsl * kScBuf + (pp & 3) * 4;
        const int un = pp + 1;
        const int xs_n = la.vx0 + ((un & 3) ? xsl : next3(xsl)) * kScBuf + (un & 3) * 4;
        const int nxt_ok = nxt.nq ? 1 : 0;
        const int dma_base = lds0 + ((stoff + 3 * kStage) & (kStages * kStage - 1)) + wave * 1024;
        // next real unit's weight-block scales; row table, source unit of the position three ahead
        int wsaddr, tsel, soff;
        if constexpr (KIND == 2) {
            const int far = un >= U ? nxt_ok : 0;
            wsaddr = lds0 + kWs + wave * 512 + (far ^ tbuf) * 4096 + (un < U ? un : (nxt_ok ? 0 : U - 1)) * 4;
            tsel = (nxt_ok ^ tbuf) * 2048;
            soff = (nxt_ok ? SLOT : U - 1) * 128;               // (the tail's k-th position fetches the next item's unit k)
        } else {
            wsaddr = lds0 + kWs + wave * 512 + tbuf * 4096 + un * 4;
            tsel = tbuf * 2048;
            const int tu = pp + 3;
            soff = (tu < U ? tu : U - 1) * 128;                 // (a null position's loads: a harmless repeat)
        }
        const bool real = KIND != 2 || pp < U;
        int ws0n = ws0, ws1n = ws1, t[4];
        // A(u) is in its slot.  (The three units behind an item switch: the epilogue's stores are in the ledger behind
        // the loads these waits are about -- at least 16 (8) of them when st_cls is 2 (1) -- and must not be waited for.
        // (Counting FEWER stores than were issued only makes a wait stricter; counting more would let it pass with a
        // load outstanding.)
        const bool late = KIND == 0 && st_cls == 2, late8 = KIND == 0 && st_cls == 1;
        unsigned tt0 = 0;
        if constexpr (DBG & 256) tt0 = now();
        if (late) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else if (late8) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        if constexpr (DBG & 256) t_top += now() - tt0;
        if (real && !(DBG & 2)) {
            a8w_block0<SLOT, DBG, OPT>(lo_c, hi_c, xs_c, ws0, ws1, wsaddr, la.tb_tok + tsel, ws0n, ws1n, t);
            ws0n = __builtin_amdgcn_readfirstlane(ws0n);       // (scalars from here on: two registers less across the unit)
            ws1n = __builtin_amdgcn_readfirstlane(ws1n);
        } else {
            asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:256\n\tds_read_b32 %2, %4 offset:512\n\t"
                         "ds_read_b32 %3, %4 offset:768\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])
                         : "v"(la.tb_tok + tsel)
                         : "memory");
        }
        // tokens(u+1) of THIS wave have landed; after the barrier every wave's have, and nobody reads the stage behind
        // the current one any more (the DMA's target)
        if constexpr (DBG & 256) {
            const unsigned a0 = now();
            if (late) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
            else if (late8) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            const unsigned a1 = now();
            asm volatile("s_barrier" ::: "memory");
            t_mid += a1 - a0;
            t_bar += now() - a1;
        } else if constexpr (DBG & 64) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");      // (ablation: no barrier)
        else if (late) asm volatile("s_waitcnt vmcnt(28)\n\ts_barrier" ::: "memory");
        else if (late8) asm volatile("s_waitcnt vmcnt(20)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
        if constexpr (!(DBG & 1)) {
            // token scales: a group of four units is fetched three positions before its first unit
            if constexpr (KIND == 2) {
                if (SLOT == 0 && nxt_ok && wave < 4) a8w_dma16(lds0 + kScBase + xsl_nxt0 * kScBuf + wave * 1024, scale_src(tsel), rs_xs, 0);
            } else {
                const int tu = pp + 3;
                if ((tu & 3) == 0 && tu < U && wave < 4)
                    a8w_dma16(lds0 + kScBase + next3(xsl) * kScBuf + wave * 1024, scale_src(tsel), rs_xs, (tu >> 2) * 16);
            }
        }
        if constexpr (GATED && KIND == 0 && SLOT == 0) {
            // one barrier after the item switch: every wave has read the previous item's row maxima (fused output
            // quantisation) -> clear them for this item's epilogue, a whole item away
            if (fq) asm volatile("ds_write_b32 %0, %1" ::"v"(lds0 + kRmax + tid * 4), "v"(0) : "memory");
        }
        if constexpr (KIND == 0 && SLOT == 2) {
            // two barriers after the item switch: the next item's gathered rows have landed (ledger: sixteen loads were
            // issued after them) -> its table
            if (nxt.nq) store_table(tbuf ^ 1, nxt);
        }
        const int nq = real ? cur.nq : 0;
        static_for<8>([&](auto QC) __attribute__((always_inline)) {
            constexpr int q = decltype(QC)::v;
            if (q < nq) {
                if constexpr (q > 0) {
                    if constexpr (!(DBG & 2)) a8w_block_t<SLOT, 2 * q, true, (2 * q + 1) * 2048, (2 * q + 1) * 256, DBG, OPT>(lo_c, hi_c, xs_c, ws0, ws1);
                }
                if (q + 1 < nq) {
                    if constexpr (!(DBG & 2)) a8w_block_t<SLOT, 2 * q + 1, true, (2 * q + 2) * 2048, (2 * q + 2) * 256, DBG, OPT>(lo_c, hi_c, xs_c, ws0, ws1);
                    if constexpr (q < 4 && !(DBG & 1)) issue_tokens(q, t[q], dma_base, soff);
                } else {
                    // last block of the unit: prefetch block 0 of the next unit (next stage), then the rest of the unit's
                    // token pieces (below), then the block's own accumulator update
                    if constexpr (!(DBG & 2)) a8w_block_t<SLOT, 2 * q + 1, true, 0, 0, DBG, OPT>(lo_n, hi_n, xs_n, ws0, ws1);
                    if constexpr (!(DBG & 1)) {
                        static_for<4>([&](auto KC) __attribute__((always_inline)) {
                            if constexpr (decltype(KC)::v >= q) issue_tokens(decltype(KC)::v, t[decltype(KC)::v], dma_base, soff);
                        });
                    }
                    if constexpr (!(DBG & 2)) a8w_flush<2 * q + 1, DBG, OPT>();
                }
            }
        });
        if constexpr (!(DBG & 1)) {
            if (nq == 0) {      // (a null unit: all four pieces)
#pragma unroll
                for (int q = 0; q < 4; ++q) issue_tokens(q, t[q], dma_base, soff);
            }
            // A of the position three ahead, into the slot this position has just finished with
            if constexpr (KIND == 2) {
                if (nxt_ok) issue_a(IC<SLOT>{}, nxt, SLOT);
                else issue_a(IC<SLOT>{}, cur, U - 1);
            } else {
                const int tu = pp + 3;
                issue_a(IC<SLOT>{}, cur, tu < U ? tu : U - 1);
            }
        }
        if (dbg_serial) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (debug: serialised units)
        ws0 = ws0n;
        ws1 = ws1n;
        stoff = stn;
        if ((pp & 3) == 3) xsl = next3(xsl);
    };

    // ---- the items: head triple, middle triples, tail triple, then the epilogue (D layout lane (g, j): rows tile*16 +
    // g*4 + r, token column j of block b) and the switch to the next item, whose first units are already in flight
    for (;;) {
        unit(IC<0>{}, IC<0>{}, 0);
        unit(IC<1>{}, IC<0>{}, 1);
        unit(IC<2>{}, IC<0>{}, 2);
        for (int p0 = 3; p0 + 3 < Up; p0 += 3) {
            unit(IC<0>{}, IC<1>{}, p0);
            unit(IC<1>{}, IC<1>{}, p0 + 1);
            unit(IC<2>{}, IC<1>{}, p0 + 2);
        }
        unit(IC<0>{}, IC<2>{}, Up - 3);
        unit(IC<1>{}, IC<2>{}, Up - 2);
        unit(IC<2>{}, IC<2>{}, Up - 1);
        {
            const Item done = cur;
            const bool more = nxt.nq != 0;
            // 16-byte stores (a8w_pair16) where this wave's tiles lie whole inside the matrix and the output is 16-bit
            bool wide = false;
            if constexpr (LKM_A8W_WIDE_EPI >= 2 && GATED) {
                wide = done.tbase + wave < T_half && (done.tbase + wave + 1) * 16 <= p.n_real && p.act_type != LKM_ACT_SWIGLUOAI &&
                       (p.ldo & 7) == 0;
            } else if constexpr (LKM_A8W_WIDE_EPI && !IS_G1) {
                wide = done.tbase + 2 * wave + 1 < T_half && (done.tbase + 2 * wave + 2) * 16 <= p.n_real && p.y_dt != LKM_DT_F32 &&
                       (p.ldo & 7) == 0;
            }
            {   // store INSTRUCTIONS this wave is about to issue (one with no active lane is branched over): 8-byte stores =
                // blocks that hold rows x its tiles inside the matrix; 16-byte stores = one per 32-row pair of blocks
                // (gated GEMM1: two blocks of one tile) or one per block (GEMM2: both tiles)
                const int nb = (done.rows + 15) >> 4;
                const int t0 = GATED ? done.tbase + wave : done.tbase + 2 * wave;
                const int nt = GATED ? (t0 < T_half && t0 * 16 < p.n_real ? 1 : 0)
                                     : (t0 < T_half && t0 * 16 < p.n_real ? 1 : 0) + (t0 + 1 < T_half && (t0 + 1) * 16 < p.n_real ? 1 : 0);
                // (fused quantisation: one 8-byte store per pair, plus one scale store on wave q of pair q)
                const int st_n = (DBG & 512) ? 0 : wide ? (GATED ? done.nq + ((fq && wave < done.nq) ? 1 : 0) : nb) : nb * nt;
                st_cls = st_n >= 16 ? 2 : st_n >= 8 ? 1 : 0;
            }
            // (opaque lane coordinates: the compiler must not hoist the epilogue's sixteen row offsets out of the K loop --
            // it has 39 registers, and a spilled value is a scratch load inside the hand-counted loop)
            // (from here to A8W_EPILOGUE_END the compiler may use v40..v77: nothing of the fixed map below the A ring is live
            // across an item switch -- block 0 of the next unit re-reads its B operand and token scale behind the epilogue)
            asm volatile("; A8W_EPILOGUE_BEGIN" ::: "memory");
            int jj = j, gg = g;
            asm volatile("" : "+v"(jj), "+v"(gg));
            if (more) {
                // the switch first (its gather goes out before the epilogue's stores), the epilogue after
                xsl = xsl_nxt0;
                xsl_nxt0 = (xsl + n_grp) % 3;
                tbuf ^= 1;
                cur = nxt;
                meta_a_done(pend);                     // (its record landed a whole item ago)
                nxt = make_item(pend);
                gather_item(nxt);                      // (no such item: repeats item 0's addresses, harmless)
                fetch_ws(nxt, tbuf ^ 1);
                pend = meta_a(it_next);
                it_next += stride;
            }
            auto read_acc = [&](auto BC, f32x4& c0, f32x4& c1) __attribute__((always_inline)) {
                asm volatile("v_mov_b32 %0, v[%c8+0]\n\tv_mov_b32 %1, v[%c8+1]\n\tv_mov_b32 %2, v[%c8+2]\n\tv_mov_b32 %3, v[%c8+3]\n\t"
                             "v_mov_b32 %4, v[%c8+4]\n\tv_mov_b32 %5, v[%c8+5]\n\tv_mov_b32 %6, v[%c8+6]\n\tv_mov_b32 %7, v[%c8+7]"
                             : "=&v"(c0.x), "=&v"(c0.y), "=&v"(c0.z), "=&v"(c0.w), "=&v"(c1.x), "=&v"(c1.y), "=&v"(c1.z), "=&v"(c1.w)
                             : "i"(kAcc + decltype(BC)::v * 8)
                             : "memory");
            };
            if (GATED && fq && !(DBG & 512)) {
                // ---- gated GEMM1 with the intermediate's 1 x 128 fp8 quantisation fused in (per_token_group_quant_fp8,
                // dispatch.hip quant_fp8_rows_kernel: same operations on the same bf16 / f16 values, bit-identical
                // bytes and scales).  An item is 256 rows x ONE 128-column group: wave w holds columns 16 w .. 16 w + 15.
                // Per half of the item (four 32-row pairs; the packed outputs of a half are 16 registers): row maxima
                // of the wave's 16 columns -> ds_max_u32 into the half's 256 words -> barrier -> every wave reads the
                // row's maximum over the 128 columns, quantises its eight values and stores eight bytes; wave q of pair q
                // also stores the scale.  The host enables this only where every wave's tile lies inside the matrix
                // (I % 128 == 0: all eight waves take this path and meet at the barriers).
                if constexpr (GATED) {
#pragma clang fp contract(off)
                    const int godd = gg & 1, gcol = (gg & ~1) * 4;
                    const int col = (done.tbase + wave) * 16 + gcol;
                    const int qgrp = done.tbase >> 3;                 // the 128-column group (TPH = 8 tiles)
                    int rbase = lds0 + kRmax;                         // (a plain local: the nested generic lambdas capture it)
                    static_for<2>([&](auto HC) __attribute__((always_inline)) {
                        constexpr int hf = decltype(HC)::v;
                        if (hf * 4 < done.nq) {      // (uniform over the workgroup)
                            u32x4 piece[4];
                            static_for<4>([&](auto QC) __attribute__((always_inline)) {
                                constexpr int qq = decltype(QC)::v, q = hf * 4 + qq;
                                if (q < done.nq) {
                                    f32x4 c0, c1;
                                    read_acc(IC<2 * q>{}, c0, c1);
                                    const u32x2 oa = a8w_silu_mul<ADT>(p, c0, c1);
                                    read_acc(IC<2 * q + 1>{}, c0, c1);
                                    const u32x2 ob = a8w_silu_mul<ADT>(p, c0, c1);
                                    piece[qq] = a8w_pair16(oa, ob);
                                    float am = 0.0f;
#pragma unroll
                                    for (int i = 0; i < 4; ++i) {
                                        am = fmaxf(am, fabsf(ActT<ADT>::to_f32((unsigned short)(piece[qq][i] & 0xffffu))));
                                        am = fmaxf(am, fabsf(ActT<ADT>::to_f32((unsigned short)(piece[qq][i] >> 16))));
                                    }
                                    // (lane ^ 32 = the other eight columns of the same row)
                                    // (the elements through scalars: __builtin_bit_cast of a vector ELEMENT reads element 0)
                                    const unsigned ab = __builtin_bit_cast(unsigned, am);
                                    const u32x2 sw = __builtin_amdgcn_permlane32_swap(ab, ab, false, false);
                                    const unsigned sw0 = sw.x, sw1 = sw.y;
                                    am = fmaxf(__builtin_bit_cast(float, sw0), __builtin_bit_cast(float, sw1));
                                    const int rt = (2 * q + godd) * 16 + jj;
                                    const int ra = rbase + hf * 1024 + rt * 4;
                                    const unsigned amb = __builtin_bit_cast(unsigned, am);
                                    if (gg < 2) asm volatile("ds_max_u32 %0, %1" ::"v"(ra), "v"(amb) : "memory");
                                }
                            });
                            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                            static_for<4>([&](auto QC) __attribute__((always_inline)) {
                                constexpr int qq = decltype(QC)::v, q = hf * 4 + qq;
                                if (q < done.nq) {
                                    const int rt = (2 * q + godd) * 16 + jj;
                                    const int ra = rbase + hf * 1024 + rt * 4;
                                    unsigned mb;
                                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(mb) : "v"(ra) : "memory");
                                    float amax = __builtin_bit_cast(float, mb);
                                    if (amax < 1e-10f) amax = 1e-10f;
                                    const float sc = amax / 448.0f;
                                    const DivBy d = make_div_by(sc);
                                    float qv[8];
#pragma unroll
                                    for (int i = 0; i < 4; ++i) {
                                        qv[2 * i] = fminf(fmaxf(div_by(ActT<ADT>::to_f32((unsigned short)(piece[qq][i] & 0xffffu)), d), -448.0f), 448.0f);
                                        qv[2 * i + 1] = fminf(fmaxf(div_by(ActT<ADT>::to_f32((unsigned short)(piece[qq][i] >> 16)), d), -448.0f), 448.0f);
                                    }
                                    u32x2 o;
                                    int pk = __builtin_amdgcn_cvt_pk_fp8_f32(qv[0], qv[1], 0, false);
                                    o.x = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(qv[2], qv[3], pk, true);
                                    pk = __builtin_amdgcn_cvt_pk_fp8_f32(qv[4], qv[5], 0, false);
                                    o.y = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(qv[6], qv[7], pk, true);
                                    if (rt < done.rows) {
                                        *(u32x2*)(p.out_q + (size_t)(done.orow0 + rt) * p.ldo + col) = o;
                                        if (gg < 2 && wave == (q & 7)) p.out_qs[(size_t)(done.orow0 + rt) * p.ld_qs + qgrp] = sc;
                                    }
                                }
                            });
                        }
                    });
                }
            } else if (wide && !(DBG & 512)) {
                const int godd = gg & 1, gcol = (gg & ~1) * 4;
                static_for<8>([&](auto QC) __attribute__((always_inline)) {
                    constexpr int q = decltype(QC)::v;
                    if (q < done.nq) {      // (uniform)
                        if constexpr (GATED) {
                            // blocks 2 q and 2 q + 1 of the wave's one output tile: even g stores the first's row, odd g the second's
                            f32x4 c0, c1;
                            read_acc(IC<2 * q>{}, c0, c1);
                            const u32x2 oa = a8w_silu_mul<ADT>(p, c0, c1);
                            read_acc(IC<2 * q + 1>{}, c0, c1);
                            const u32x2 ob = a8w_silu_mul<ADT>(p, c0, c1);
                            const u32x4 piece = a8w_pair16(oa, ob);
                            const int rt = (2 * q + godd) * 16 + jj;
                            if (rt < done.rows)
                                *(u32x4*)((unsigned short*)p.out + (size_t)(done.orow0 + rt) * p.ldo + (done.tbase + wave) * 16 + gcol) = piece;
                        } else {
                            // the wave's two adjacent tiles of one block: even g stores into the first, odd g into the second
                            static_for<2>([&](auto HC) __attribute__((always_inline)) {
                                constexpr int b = 2 * q + decltype(HC)::v;
                                f32x4 c0, c1;
                                read_acc(IC<b>{}, c0, c1);
                                const u32x4 piece = a8w_pair16(u32x2{ActT<ADT>::pack2(c0.x, c0.y), ActT<ADT>::pack2(c0.z, c0.w)},
                                                               u32x2{ActT<ADT>::pack2(c1.x, c1.y), ActT<ADT>::pack2(c1.z, c1.w)});
                                const int rt = b * 16 + jj;
                                if (rt < done.rows)
                                    *(u32x4*)((unsigned short*)p.out + (size_t)(done.orow0 + rt) * p.ldo + (done.tbase + 2 * wave + godd) * 16 + gcol) = piece;
                            });
                        }
                    }
                });
            } else
            static_for<16>([&](auto BC) __attribute__((always_inline)) {
                constexpr int b = decltype(BC)::v;
                if (b < 2 * done.nq) {      // (uniform)
                    const int rt = b * 16 + jj;
                    f32x4 c0, c1;
                    read_acc(BC, c0, c1);
                    if (rt < done.rows && !(DBG & 512)) {
                        const size_t orow = (size_t)(done.orow0 + rt);
                        if constexpr (GATED) {
                            const int n = (done.tbase + wave) * 16 + gg * 4;
                            if (done.tbase + wave < T_half && n < p.n_real) {
                                if (p.act_type == LKM_ACT_SWIGLUOAI) store_gemm1_frag<ADT, true, true>(p, c0, c1, orow, n);   // (LEGACY: gemm_skinny.h act_silu_poly)
                                else a8w_store_silu_mul<ADT>(p, c0, c1, orow, n);
                            }
                        } else {
                            static_for<2>([&](auto TC) __attribute__((always_inline)) {
                                constexpr int t = decltype(TC)::v;
                                const int tl = done.tbase + 2 * wave + t;
                                const int n = tl * 16 + gg * 4;
                                const f32x4 v = t ? c1 : c0;
                                if (tl < T_half && n < p.n_real) {
                                    if constexpr (IS_G1) store_gemm1_frag<ADT, false, true>(p, v, v, orow, n);
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
            asm volatile("; A8W_EPILOGUE_END" ::: "memory");
            if (!more) break;
            {
                // Block 0 of the new item's first unit: B operand and token scale, read HERE and not by the old item's last
                // block -- the compiler's epilogue code is free to use the B / P / x / f registers as temporaries (it
                // must stay below v78, under the A ring: tests/test_a8w_codegen.py), and nothing else of
                // the fixed map holds a value across the epilogue.  Visible since the barrier of the old item's last unit.
                asm volatile("ds_read_b128 v[%c[b]:%c[b]+3], %[lo]\n\t"
                             "ds_read_b128 v[%c[b]+4:%c[b]+7], %[hi]\n\t"
                             "ds_read_b32 v[%c[x]], %[xs]\n\t"
                             "s_waitcnt lgkmcnt(0)\n\t"
                             :
                             : [b] "i"(kB), [x] "i"(kX), [lo] "v"(la.vb_lo + stoff), [hi] "v"(la.vb_hi + stoff), [xs] "v"(la.vx0 + xsl * kScBuf)
                             : "memory", A8W_CLOB_B, A8W_CLOB_TOP);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // trailing (clamped) loads must not outlive the wave
    if constexpr (DBG & 256) {
        const unsigned tot = now() - t_start;
        if ((blockIdx.x == 8 || blockIdx.x == 77) && lane == 0 && (wave == 0 || wave == 5))
            printf("a8w wg %d wave %d: total %u ticks, wait weights %u, wait tokens %u, barrier %u\n", (int)blockIdx.x, wave, tot, t_top, t_mid, t_bar);
    }
#else
    (void)p;
#endif
}

// usable when K is a whole number of 128-byte units, every 16-row weight tile has ONE block scale per K unit (groupN a
// multiple of 16), the weight image is tile-major or unit-major with 32-bit offsets, and the operand matrices fit the
// 2 GiB buffer windows; otherwise the plan stays on gemm_tiled_kernel (pick_cfg asks prefill_a8w_shape_ok first)
inline bool prefill_a8w_ok(const GemmParams& p, bool is_g1 = true) {
    // (the slot -> token division through a float reciprocal is exact below 2^22 slots and only GEMM1 divides: its rows are
    // gathered through sorted_slot; GEMM2 reads the expert-sorted intermediate row by row -- ADVICE r3)
    const bool slots_ok = !is_g1 || (size_t)p.x_rows * (size_t)(p.top_k > 0 ? p.top_k : 1) < ((size_t)1 << 22);
    return p.Kreal % 128 == 0 && p.tile_uniform_scale && p.U >= 8 && p.U <= 64 && slots_ok &&      // (the item-boundary pipeline; 64 weight-block scales per tile in the landing zone)
           (size_t)p.x_rows * (size_t)p.ldx < (size_t)0x7fffffff &&
           (size_t)p.x_rows * (size_t)p.ld_xscale * 4 < (size_t)0x7fffffff &&
           (size_t)p.T_half * p.halves * p.U * 2048 < (size_t)0x7fffffff;
}

template <int ADT, bool GATED, bool IS_G1, int DBG = 0>
static int launch_prefill_a8w_t(hipStream_t st, const GemmParams& p, int max_tiles) {
    constexpr size_t lds = (size_t)a8w::kLdsBytes;
    // persistent: one workgroup per CU (143 KiB of LDS and 256 registers x 512 threads admit exactly one)
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0, v = 0;
        LKM_HIP_CHECK(hipGetDevice(&dev));
        LKM_HIP_CHECK(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
        n_cu = v > 0 ? (v / 8) * 8 : 256;
        if (n_cu < 8) n_cu = 8;
    }
    (void)max_tiles;
    dim3 grid(n_cu), block(512);
    auto kern = gemm_prefill_a8w_kernel<ADT, GATED, IS_G1, DBG>;
    LKM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    LKM_LAUNCH_GEMM(kern, grid, block, lds, st, p);
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}

// gemm_a8w_dbg.hip: the ablation instantiations (bf16: gated GEMM1, or GEMM2 when the tuning value carries bit 1024)
int launch_prefill_a8w_dbg(hipStream_t st, const GemmParams& p, int max_tiles, int dbg, bool gemm2);

template <typename ADTC>
static bool launch_prefill_a8w_if(hipStream_t st, const LaunchCfg& cfg, const GemmParams& p, bool gated, bool is_g1,
                                  int max_tiles, int* rc, ADTC) {
    constexpr int ADT = ADTC::v;
    if (cfg.tiled != 256 || cfg.pf != 9) return false;
    if (!prefill_a8w_ok(p, is_g1)) {
        set_error("fp8 W8A8 prefill kernel: shape not eligible (K %% 128, scale granularity, 2^22 slots or 2 GiB windows)");
        *rc = LKM_E_INVALID;
        return true;
    }
#ifdef LKM_ABLATIONS
    if constexpr (ADT == LKM_DT_BF16) {     // ablation builds: bf16 activations only; bit 1024 = of GEMM2 instead of the gated GEMM1
        if ((p.dbg & 0x3fb) && ((p.dbg & 0x400) ? !is_g1 : (is_g1 && gated))) {
            *rc = launch_prefill_a8w_dbg(st, p, max_tiles, p.dbg & 0x3fb, !is_g1);
            return true;
        }
    }
#endif
    if (is_g1) *rc = gated ? launch_prefill_a8w_t<ADT, true, true>(st, p, max_tiles) : launch_prefill_a8w_t<ADT, false, true>(st, p, max_tiles);
    else *rc = launch_prefill_a8w_t<ADT, false, false>(st, p, max_tiles);
    return true;
}

}  // namespace lkm
