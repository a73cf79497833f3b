// gemm_tiled.h -- per-expert grouped GEMMs for 32 < rows-per-expert (decode at large batch,
// prefill): same math as gemm_skinny.h, different data movement.
//
//   * a workgroup = WAVES wavefronts = one (expert, token tile of TM = 16*TBW rows) work item x one
//     group of WAVES*NT 16-row weight tiles.  The TOKEN operand (MFMA B) is staged through LDS once
//     per K unit and shared by all waves (XOR-swizzled rows -> conflict-free ds_read_b128), so its
//     L2 traffic is amortised over WAVES*NT*(1|2) weight tiles instead of being re-read per tile;
//   * the WEIGHT operand (MFMA A) still goes HBM/L2 -> VGPR directly in the pre-shuffled layout:
//     every wave owns different rows, so there is nothing to share and no LDS round trip;
//   * accumulators NTT x TBW x f32x4 per lane; one s_barrier per K unit, LDS double-buffered, the
//     next unit's token rows and weight fragments are in flight (registers) across the barrier.
// Work items come from the device-side list the sort kernel emits (tile_e / tile_r0, meta[3]).
#pragma once
#include "gemm_skinny.h"

namespace lkm {

template <int ROWB>
__device__ __forceinline__ int x_swizzle(int row) {
    // 16-byte slot permutation inside a token row so that 16 lanes reading the same logical slot of
    // 16 different rows hit 16 different 16-B bank groups of the 256-B LDS bank row.
    return ROWB == 128 ? ((row >> 1) & 7) : (row & 15);
}

template <int I>
struct IC {
    static constexpr int v = I;
};
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(IC<I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// v_mfma_f32_32x32x16 (gemm_w4x.h, gemm_w4e.h, gemm_prefill.h): A lane l = row l % 32, k = 8 (l / 32) .. + 7; B likewise per
// token; D register i = row 8 (i / 4) + 4 (l / 32) + i % 4, token l % 32
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int ADT>
struct Mfma32;
template <>
struct Mfma32<LKM_DT_BF16> {
    static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <>
struct Mfma32<LKM_DT_F16> {
    static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

// resident workgroups per CU the register allocation must allow (second __launch_bounds__ argument)
constexpr int tiled_min_blocks(int wf, int tbw, int waves, bool gated_g1, int pd) {
    // 4-bit decode tiles are latency/issue bound with two waves per SIMD, and the gated 64-row kernel
    // sits ~10 registers above the three-wave budget (168): forcing it there (60 B of scratch) measured
    // GEMM1 -8 % int4, -5 % NVFP4, -2 % MXFP4 at Mixtral M=128
    if ((wf == LKM_W_INT4_B8 || wf == LKM_W_MXFP4 || wf == LKM_W_NVFP4) && tbw == 4 && waves == 4 && gated_g1 && pd == 2) return 3;
    // the 32-row tile of the same formats is two registers above FOUR waves per SIMD (int4: 130)
    if ((wf == LKM_W_INT4_B8 || wf == LKM_W_MXFP4 || wf == LKM_W_NVFP4) && tbw == 2 && waves == 4 && gated_g1 && pd == 2) return 4;
    // (the fp8 kernels sit 34 registers above that budget: forced there they spill 156-236 B and LOSE,
    // DSv3 rank slice GEMM1 151 -> 178 us)
    return 1;
}

#ifndef LKM_ABL
#define LKM_ABL 0
#endif

template <int WF, int ADT, int NT, int TBW, int WAVES, bool GATED, bool IS_G1, int PD, bool NTL>
__global__ __launch_bounds__(WAVES * 64, tiled_min_blocks(WF, TBW, WAVES, GATED && IS_G1, PD)) void gemm_tiled_kernel(GemmParams p) {
    typedef Dec<WF, ADT> D;
    // PD = weight register stages (prefetch distance PD-1 units), XD = token-row register stages
    // (prefetch distance XD units).  The products of (resident waves) x (bytes in flight per wave)
    // must cover HBM latency x bandwidth (~16 MB chip-wide): a sub-16-bit unit is only 1 KiB per
    // tile, so those formats need the deeper ring.  (Measured for the 4-bit formats at Mixtral M=128:
    // eight stages are SLOWER than two -- GEMM1 int4 238 vs 141 us, MXFP4 119 vs 107 -- the registers
    // cost a resident wave, and resident waves hide more latency than a deeper ring: tiled_min_blocks.)
    static_assert(PD >= 2 && PD % 2 == 0, "PD even: the LDS buffer parity is the unroll index parity");
    constexpr int XD = PD > 2 ? 2 : 1;
    constexpr int NTT = (IS_G1 && GATED) ? 2 * NT : NT;
    constexpr int TM = TBW * 16;
    constexpr int THREADS = WAVES * 64;
    constexpr int XB = D::A8 ? 1 : 2;            // bytes per activation element (fp8 when W8A8)
    constexpr int ROWB = D::UNITK * XB;          // bytes of one token row per K unit
    constexpr int SLOTS = ROWB / 16;
    constexpr int BUFB = TM * ROWB + (D::XS ? TM * 4 : 0);   // + per-row scalars of the unit (W8A8 scales, INT4_PS sums)
    constexpr int PIECES = TM * SLOTS / THREADS;
    static_assert(PIECES >= 1 && TM * SLOTS % THREADS == 0, "staging split");
    extern __shared__ __attribute__((aligned(16))) char xlds[];   // [2]{[TM][ROWB], A8: float[TM]}

    // Measured and NOT used at 256-row tiles (GLM prefill, gemm1/gemm2 us): writing the next unit's token
    // rows to LDS at the top of the iteration + double-buffered token-fragment chunks interleaved with
    // the MFMAs by sched_group_barrier: 2123/1168 vs 1853/1048; 4 waves x 4 weight tiles x 256 tokens (one
    // wave per SIMD, 512 registers): 2452/1233.  What this kernel lacks for the MFMA roof is the
    // weight operand through LDS (glds) with a counted-vmcnt multi-phase schedule -- see DESIGN_history.md 6.
    // Work mapping: blockIdx.x = row group (fastest), blockIdx.y = (expert, token tile) item, default
    // round-robin XCD placement (all XCDs work on the same one or two items at a time).  Measured:
    // (1) the XCD-aware mapping below (`xcd` tuning knob, off by default) cuts GLM-prefill GEMM1 HBM
    // traffic 11.4 -> 7.6 GB per launch but the time only 5 % (GEMM2 +6 %), and loses 40 % on Mixtral
    // M=1024 where every expert has a single tile (profiles/r01_prefill_pmc.md); (2) skipping the
    // empty 16-token blocks of a partially filled tile with wave-uniform branches INSIDE the K loop
    // loses 10 % -- the block count is a template parameter of the loop instead (run<NB>).
    int ti = blockIdx.y + (p.tile_lo_meta ? p.meta[p.tile_lo_meta] : 0), bx = blockIdx.x;     // (mixed tile heights: my part of the list)
    if (p.xcd_map) {
        // 1-D grid; hardware places workgroup L on XCD L % 8.  XCD c takes the contiguous run of tiles the sort
        // kernel cut for it (dispatch.hip: meta[8 + c], balanced by routed rows; tiles of the same expert are
        // adjacent in the list) and, inside it, row group fastest: the workgroups that share a token tile or a
        // weight panel run on ONE L2 at the same time.
        const int RG = p.xcd_map;
        const int L = blockIdx.x, c = L & 7, sidx = L >> 3;
        const int first = p.meta[8 + c], n_c = p.meta[9 + c] - first;
        if (sidx >= n_c * RG) return;
        ti = first + sidx / RG;
        bx = sidx % RG;
    }
    if (ti >= p.meta[p.tile_hi_meta ? p.tile_hi_meta : 3]) return;
    const int e = p.tile_e[ti], r0 = p.tile_r0[ti];
    const int m_e = p.counts[e], off_e = p.offsets[e];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, j = lane & 15;
    const int tile0 = (bx * WAVES + wave) * NT;
    const bool wave_on = tile0 < p.T_half;            // tail group of a padded tile count
    const int T_all = p.T_half * p.halves;
    const int dparam = WF == LKM_W_NVFP4 ? __builtin_bit_cast(int, p.gs ? p.gs[e] : 1.0f) : p.spu;

    // split-K (GEMM2 only, blockIdx.z = slab): this workgroup multiplies units [u0, u0 + U_loc) and writes
    // its fp32 partial to slab `sk`; combine_kernel sums the slabs.  Used when an EP rank holds so few
    // experts that the (tile group x token tile) grid alone leaves most of the chip idle.
    const int sk = IS_G1 ? 0 : blockIdx.z;
    const int u0 = IS_G1 ? 0 : (int)((long long)sk * p.U / p.SK);
    const int U_loc = IS_G1 ? p.U : (int)((long long)(sk + 1) * p.U / p.SK) - u0;
    const int k_base = u0 * D::UNITK;                 // element offset of the slice
    const int Kreal_loc = p.Kreal - k_base;           // for the ragged-tail test of the last unit
    // a wave past the padded tile count (wave_on == false) streams tile 0 of the expert and drops the
    // result: every load of the K loop is unconditional (see the loop comment)
    const u32x4* wp[NTT];
    const char* auxp[NTT];
    const int aux_step = D::aux_step(p.spu), wstep = (int)p.w_ustride;
#pragma unroll
    for (int t = 0; t < NTT; ++t) {
        const int tile = (IS_G1 && GATED && t >= NT) ? p.T_half + tile0 + (t - NT) : tile0 + t;
        const size_t tl = (size_t)e * T_all + (wave_on ? tile : 0);
        wp[t] = (const u32x4*)p.w + (size_t)e * p.w_estride + (size_t)(wave_on ? tile : 0) * p.w_tstride + (size_t)u0 * p.w_ustride + lane;
        auxp[t] = D::aux_ptr(p.s, tl * p.U + u0, lane, p.spu);
    }

    // staging assignment of this thread: PIECES 16-byte pieces per unit
    const unsigned char* xrow[PIECES];
    int xsrc_off[PIECES];   // element offset of the piece inside the unit
    int xdst[PIECES];       // byte offset inside one LDS buffer
#pragma unroll
    for (int q = 0; q < PIECES; ++q) {
        const int pc = q * THREADS + tid;
        const int row = pc / SLOTS, pslot = pc % SLOTS;
        const int lslot = pslot ^ x_swizzle<ROWB>(row);
        const int r = r0 + row;
        const int rr = r < m_e ? r : r0;
        if (IS_G1) {
            const int slot = p.sorted_slot[off_e + rr];
            xrow[q] = (const unsigned char*)p.x + (size_t)(slot / p.top_k) * p.ldx * XB;
        } else {
            xrow[q] = (const unsigned char*)p.x + ((size_t)(off_e + rr) * p.ldx + k_base) * XB;
        }
        // 16-bit: slot ks*4+g holds k = ks*32 + g*8 .. +7;  fp8: slot i*4+g holds k = i*64 + g*16 .. +15
        xsrc_off[q] = lslot * (16 / XB);
        xdst[q] = row * ROWB + pslot * 16;
    }
    // W8A8: thread `tid` < TM also stages the activation scale of token row tid for the unit
    const float* xsrow = nullptr;
    if (D::XS) {   // threads >= TM fetch row TM-1's scale too (unconditional load) and drop it
        const int r = r0 + (tid < TM ? tid : TM - 1);
        const int rr = r < m_e ? r : r0;
        const size_t rowidx = IS_G1 ? (size_t)(p.sorted_slot[off_e + rr] / p.top_k) : (size_t)(off_e + rr);
        xsrow = p.xscale + rowidx * p.ld_xscale + u0;      // one activation scale per 128-k unit
    }
    float xsv[XD] = {};

    f32x4 acc[NTT][TBW];
#pragma unroll
    for (int t = 0; t < NTT; ++t)
#pragma unroll
        for (int b = 0; b < TBW; ++b) acc[t][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // The K loop is specialised on the number of 16-token blocks that actually hold rows (NB): a
    // partially filled tile (decode at large batch: ~32 rows in a 64-row tile) stages, reads and
    // multiplies only those blocks.  The choice is workgroup-uniform and made once, so every variant
    // keeps a branch-free steady loop (a runtime "skip empty blocks" test inside the loop measured
    // slower, see above).  Staging piece q of a thread covers rows [q*RPP, (q+1)*RPP).
    auto run_h = [&](auto NBC, auto HC) __attribute__((always_inline)) {
        constexpr int NB = decltype(NBC)::v;
        // int4 with one scale group per 128-k unit (g >= 128: the usual AWQ / GPTQ checkpoints): the two scale
        // multipliers of a weight row are formed once per unit instead of once per k-step (3 of 19 VALU per
        // fragment; the kernel is VALU-bound: 10 VALU per MFMA at 32 rows per expert).  Workgroup-uniform.
        constexpr bool HOIST = decltype(HC)::v != 0;
        constexpr int PCS = NB * 16 * SLOTS / THREADS;
        static_assert(PCS >= 1 && PCS <= PIECES && (NB * 16 * SLOTS) % THREADS == 0, "block granularity");
        struct WStage {
            u32x4 w[NTT][D::LOADS];
            typename D::Aux aux[NTT];
        };
        WStage ws[PD];
        u32x4 xs[XD][PIECES];

        // rows beyond the expert's count read a valid row (their D columns are never stored): the loads
        // are unconditional; only a ragged K tail (never for the model shapes) is zero-filled.
        // STEADY: the unit is not the last one, so it cannot be a ragged K tail -> no branch at all
        auto load_x = [&](u32x4 (&xs)[PIECES], float& xsv, int u, auto STEADY) __attribute__((always_inline)) {
            const bool tail = !decltype(STEADY)::value && (u + 1) * D::UNITK > Kreal_loc;   // workgroup-uniform
    #pragma unroll
            for (int q = 0; q < PCS; ++q) {
                const int k = u * D::UNITK + xsrc_off[q];          // relative to the slice (xrow is shifted)
                if (!tail) {
                    xs[q] = *(const u32x4*)(xrow[q] + (size_t)k * XB);
                } else {
                    u32x4 v = {0u, 0u, 0u, 0u};
                    if (k + 16 / XB <= Kreal_loc) v = *(const u32x4*)(xrow[q] + (size_t)k * XB);
                    xs[q] = v;
                }
            }
            if constexpr (D::XS) xsv = xsrow[u];
        };
        auto load_w = [&](WStage& s, int u) __attribute__((always_inline)) {
    #pragma unroll
            for (int t = 0; t < NTT; ++t) {
    #pragma unroll
                for (int l = 0; l < D::LOADS; ++l) {
                    const u32x4* a = wp[t] + (size_t)u * wstep + l * 64;
                    s.w[t][l] = NTL ? __builtin_nontemporal_load(a) : *a;
                }
                D::load_aux_at(s.aux[t], auxp[t] + (size_t)u * aux_step);
            }
        };
        auto store_x = [&](const u32x4 (&xs)[PIECES], float xsv, int buf) __attribute__((always_inline)) {
    #pragma unroll
            for (int q = 0; q < PCS; ++q) *(u32x4*)(xlds + buf * BUFB + xdst[q]) = xs[q];
            if (D::XS && tid < TM) *(float*)(xlds + buf * BUFB + TM * ROWB + tid * 4) = xsv;
        };
        auto compute = [&](const WStage& s, int buf) __attribute__((always_inline)) {
            if (!wave_on) return;
            const char* xb = xlds + buf * BUFB;
            if constexpr (D::A8) {
                // fp8 x fp8: one ds_read_b128 = the token operand of a PAIR of k-steps.  The unit's partial
                // sums (they take the block scales before they join the accumulator) are formed four token
                // blocks at a time: the weight fragment is a register select, so a second pass over the
                // k-steps costs nothing and the 128-row tile keeps `part` at NTT x 4 fragments.  (A 256-row
                // gated variant still spills 280 B and ran 2x slower: GLM-4.5-Air fp8 GEMM1 3456 vs 1728 us.)
                constexpr int BCH = NB > 4 ? 4 : NB;
                static_assert(NB % BCH == 0, "token blocks in chunks of four");
    #pragma unroll
                for (int b0 = 0; b0 < NB; b0 += BCH) {
                    f32x4 part[NTT][BCH];
    #pragma unroll
                    for (int t = 0; t < NTT; ++t)
    #pragma unroll
                        for (int b = 0; b < BCH; ++b) part[t][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    #pragma unroll
                    for (int i = 0; i < D::KSTEPS / 2; ++i) {
                        u32x4 bf[BCH];
    #pragma unroll
                        for (int b = 0; b < BCH; ++b) {
                            const int row = (b0 + b) * 16 + j;
                            bf[b] = *(const u32x4*)(xb + row * ROWB + (((i * 4 + g) ^ x_swizzle<ROWB>(row)) * 16));
                        }
    #pragma unroll
                        for (int q = 0; q < 2; ++q)
    #pragma unroll
                            for (int t = 0; t < NTT; ++t) {
                                const long a = D::frag8(s.w[t], 2 * i + q);
    #pragma unroll
                                for (int b = 0; b < BCH; ++b)
                                    part[t][b] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(
                                        a, __builtin_bit_cast(long, u32x2{bf[b][q * 2], bf[b][q * 2 + 1]}),
                                        part[t][b], 0, 0, 0);
                            }
                    }
    #pragma unroll
                    for (int b = 0; b < BCH; ++b) {
                        const f32x2 xsp = splat2_opaque(*(const float*)(xb + TM * ROWB + ((b0 + b) * 16 + j) * 4));
    #pragma unroll
                        for (int t = 0; t < NTT; ++t) acc[t][b0 + b] += scale4(s.aux[t].s, xsp) * part[t][b];
                    }
                }
            } else if constexpr (D::UNIT_SCALE) {
                f32x4 part[NTT][NB];
    #pragma unroll
                for (int t = 0; t < NTT; ++t)
    #pragma unroll
                    for (int b = 0; b < NB; ++b) part[t][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    #pragma unroll
                for (int ks = 0; ks < D::KSTEPS; ++ks) {
                    u32x4 bf[NB];
    #pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const int row = b * 16 + j;
                        bf[b] = *(const u32x4*)(xb + row * ROWB + (((ks * 4 + g) ^ x_swizzle<ROWB>(row)) * 16));
                    }
    #pragma unroll
                    for (int t = 0; t < NTT; ++t) {
                        const u32x4 a = D::frag(s.w[t], s.aux[t], ks, dparam);
    #pragma unroll
                        for (int b = 0; b < NB; ++b) part[t][b] = ActT<ADT>::mfma(a, bf[b], part[t][b]);
                    }
                }
    #pragma unroll
                for (int b = 0; b < NB; ++b) {
                    if constexpr (D::XS) {      // INT4_PS: s * (sum (BIAS + v) x - (BIAS + 8) sum x)
                        const f32x2 c = splat2_opaque(D::BIAS8 * *(const float*)(xb + TM * ROWB + (b * 16 + j) * 4));
    #pragma unroll
                        for (int t = 0; t < NTT; ++t) acc[t][b] += s.aux[t].s * sub4(part[t][b], c);
                    } else {
    #pragma unroll
                        for (int t = 0; t < NTT; ++t) acc[t][b] += s.aux[t].s * part[t][b];
                    }
                }
            } else if constexpr (WF == LKM_W_INT4_B8) {
                // in-register decode costs ~19 VALU per fragment against 4 x TBW/4 MFMAs: decode k-step
                // ks+1 while the MFMAs of k-step ks occupy the matrix pipe (one MFMA : DEC_PER VALU)
                static_assert(TBW <= 8, "int4 tiles: 64 or 128 rows");
                u32x4 a[2][NTT];
                typename D::Mult mu[NTT];
                auto dec = [&](int t, int ks) __attribute__((always_inline)) {
                    if constexpr ((LKM_ABL & 32) != 0) return s.w[t][0] + u32x4{(unsigned)ks, 0u, 0u, 0u};
                    else if constexpr (HOIST) return D::frag_m(s.w[t], ks, mu[t]);
                    else return D::frag(s.w[t], s.aux[t], ks, dparam);
                };
                if constexpr (HOIST) {
    #pragma unroll
                    for (int t = 0; t < NTT; ++t) mu[t] = D::mult(s.aux[t], 0, 0);
                }
    #pragma unroll
                for (int t = 0; t < NTT; ++t) a[0][t] = dec(t, 0);
    #pragma unroll
                for (int ks = 0; ks < D::KSTEPS; ++ks) {
                    u32x4 bf[NB];
    #pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const int row = b * 16 + j;
                        bf[b] = *(const u32x4*)(xb + row * ROWB + (((ks * 4 + g) ^ x_swizzle<ROWB>(row)) * 16));
                    }
                    if (ks + 1 < D::KSTEPS) {
    #pragma unroll
                        for (int t = 0; t < NTT; ++t) a[(ks + 1) & 1][t] = dec(t, ks + 1);
                    }
    #pragma unroll
                    for (int t = 0; t < NTT; ++t)
    #pragma unroll
                        for (int b = 0; b < NB; ++b) acc[t][b] = ActT<ADT>::mfma(a[ks & 1][t], bf[b], acc[t][b]);
                    if (ks + 1 < D::KSTEPS) {
                                                constexpr int DEC_PER = ((HOIST ? 16 : 19) * NTT + NTT * NB - 1) / (NTT * NB);
    #pragma unroll
                        for (int i = 0; i < NTT * NB; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         // one MFMA
                            __builtin_amdgcn_sched_group_barrier(0x002, DEC_PER, 0);   // then decode VALU
                        }
                    }
                }
            } else {
                // token fragments are read 8 blocks at a time so that the 256-row tile (TBW = 16) keeps
                // its 2 x 16 accumulators in registers
                constexpr int BCH = NB > 8 ? 8 : NB;
    #pragma unroll
                for (int ks = 0; ks < D::KSTEPS; ++ks) {
                    u32x4 a[NTT];
    #pragma unroll
                    for (int t = 0; t < NTT; ++t) {
                        if constexpr ((LKM_ABL & 32) != 0) a[t] = s.w[t][0] + u32x4{(unsigned)ks, 0u, 0u, 0u};
                        else a[t] = D::frag(s.w[t], s.aux[t], ks, dparam);
                    }
    #pragma unroll
                    for (int b0 = 0; b0 < NB; b0 += BCH) {
                        u32x4 bf[BCH];
    #pragma unroll
                        for (int b = 0; b < BCH; ++b) {
                            const int row = (b0 + b) * 16 + j;
                            bf[b] = *(const u32x4*)(xb + row * ROWB + (((ks * 4 + g) ^ x_swizzle<ROWB>(row)) * 16));
                        }
    #pragma unroll
                        for (int t = 0; t < NTT; ++t)
    #pragma unroll
                            for (int b = 0; b < BCH; ++b) acc[t][b0 + b] = ActT<ADT>::mfma(a[t], bf[b], acc[t][b0 + b]);
                    }
                }
            }
        };

        // K loop.  The hardware retires vector-memory loads in order and s_waitcnt counts them, so the
        // compiler can only leave the prefetched stages in flight if the number of loads issued after them
        // is the same on every path: one conditional load anywhere in the loop degrades every wait to
        // vmcnt(0) and the ring to no prefetch at all (measured: depth 2/4/8 identical until the steady
        // loop below was made branch-free).  Hence: steady loop = units whose look-ahead stays inside the
        // K range and away from the (possibly ragged) last unit, all loads unconditional; the last few
        // units run in the drain loop with the bounds checks.
        const int U = U_loc;
        typedef std::true_type Steady;
        typedef std::false_type Drain;
        load_x(xs[0], xsv[0], 0, Drain{});
    #pragma unroll
        for (int s = 0; s < PD - 1; ++s)
            if (s < U) load_w(ws[s], s);
        store_x(xs[0], xsv[0], 0);
        if (XD == 2 && 1 < U) load_x(xs[1], xsv[1], 1, Drain{});
        __syncthreads();
        constexpr int LOOK = PD - 1 > XD ? PD - 1 : XD;
        const int Um = U - 1 - LOOK > 0 ? (U - 1 - LOOK) / PD * PD : 0;
        int u = 0;
        for (; u < Um; u += PD) {
            static_for<PD>([&](auto H) __attribute__((always_inline)) {
                constexpr int h = decltype(H)::v;
                const int uu = u + h;
                // LKM_ABL: compile-time ablations of the steady loop for experiments (python -m lvllm_amd.build
                // --flag=-DLKM_ABL=n --only=gemm_tiled; results are wrong): 4 no barrier, 8 no token staging, 16 no
                // weight loads, 32 no weight decode (compute() below)
                if constexpr (!(LKM_ABL & 8)) load_x(xs[h % XD], xsv[h % XD], uu + XD, Steady{});   // token rows first: their wait
                if constexpr (!(LKM_ABL & 16)) load_w(ws[(h + PD - 1) % PD], uu + PD - 1);           // leaves the weights in flight
                __builtin_amdgcn_sched_barrier(0);                    // loads are issued before the MFMAs
                compute(ws[h], h & 1);
                if constexpr (!(LKM_ABL & 8)) store_x(xs[(h + 1) % XD], xsv[(h + 1) % XD], (h + 1) & 1);
                if constexpr (!(LKM_ABL & 4)) __syncthreads();
            });
        }
        for (; u < U; u += PD) {
            static_for<PD>([&](auto H) __attribute__((always_inline)) {
                constexpr int h = decltype(H)::v;
                const int uu = u + h;
                if (uu < U) {
                    if (uu + XD < U) load_x(xs[h % XD], xsv[h % XD], uu + XD, Drain{});
                    if (uu + PD - 1 < U) load_w(ws[(h + PD - 1) % PD], uu + PD - 1);
                    compute(ws[h], h & 1);
                    if (uu + 1 < U) store_x(xs[(h + 1) % XD], xsv[(h + 1) % XD], (h + 1) & 1);
                    __syncthreads();
                }
            });
        }

    };
    auto run = [&](auto NBC) __attribute__((always_inline)) {
        if constexpr (WF == LKM_W_INT4_B8) {
            if (p.spu <= 1) return run_h(NBC, IC<1>{});
        }
        run_h(NBC, IC<0>{});
    };
    {
        constexpr int GRAN = THREADS / (16 * SLOTS) > 1 ? THREADS / (16 * SLOTS) : 1;   // blocks per staging piece
        const int rows_here = m_e - r0 < TM ? m_e - r0 : TM;
        const int nb = (rows_here + 15) >> 4;
        if constexpr (TBW == 4 && GRAN == 1) {
            if (nb <= 1) run(IC<1>{});
            else if (nb == 2) run(IC<2>{});
            else if (nb == 3) run(IC<3>{});
            else run(IC<4>{});
        } else if constexpr (TBW == 4 && GRAN == 2) {
            if (nb <= 2) run(IC<2>{});
            else run(IC<4>{});
        } else if constexpr (TBW == 2 && GRAN == 1) {
            if (nb <= 1) run(IC<1>{});
            else run(IC<2>{});
        } else if constexpr (TBW >= 8 && (TBW / 4) % GRAN == 0) {
            // prefill tiles in quarters: the ragged last tile of an expert (GLM: 512 +- 22 rows over
            // 256-row tiles) costs what it holds, not a full tile
            constexpr int Q = TBW / 4;
            if (nb <= Q) run(IC<Q>{});
            else if (nb <= 2 * Q) run(IC<2 * Q>{});
            else if (nb <= 3 * Q) run(IC<3 * Q>{});
            else run(IC<TBW>{});
        } else {
            run(IC<TBW>{});
        }
    }

    // epilogue: D layout lane (g,j): rows tile*16 + g*4 + r, token column j of block b.  Blocks and
    // tiles are compile-time indices (static_for): a runtime index into acc[][] would move the whole
    // accumulator array to scratch memory.
    if (!wave_on) return;
    static_for<TBW>([&](auto BC) __attribute__((always_inline)) {
        constexpr int b = decltype(BC)::v;
        const int r_tok = r0 + b * 16 + j;
        if (r_tok < m_e) {
            static_for<NT>([&](auto TC) __attribute__((always_inline)) {
                constexpr int t = decltype(TC)::v;
                const int n = (tile0 + t) * 16 + g * 4;
                if (n < p.n_real) {
                    if constexpr (IS_G1) store_gemm1_frag<ADT, GATED>(p, acc[t][b], acc[NTT - NT + t][b], (size_t)(off_e + r_tok), n);
                    else store_gemm2_frag(p, acc[t][b], sk, (size_t)(off_e + r_tok), n);
                }
            });
        }
    });
}

template <int WF, int ADT, int NT, int TBW, int WAVES, bool GATED, bool IS_G1, int PD>
static int launch_tiled_t(hipStream_t st, const GemmParams& p, int max_tiles) {
    constexpr int ROWB = Dec<WF, ADT>::UNITK * (Dec<WF, ADT>::A8 ? 1 : 2);
    constexpr size_t lds = (size_t)2 * (TBW * 16 * ROWB + (Dec<WF, ADT>::XS ? TBW * 16 * 4 : 0));
    const int RG = ceil_div(p.T_half, WAVES * NT);
    dim3 grid(RG, max_tiles, IS_G1 ? 1 : p.SK), block(WAVES * 64);
    GemmParams pp = p;
    if (p.xcd_map) {     // 1-D launch, see the kernel's work mapping; p.xcd_map = longest run of one XCD
        pp.xcd_map = RG;
        grid = dim3(8 * p.xcd_map * RG, 1, IS_G1 ? 1 : p.SK);
    }
    auto kern = p.stream_nt ? gemm_tiled_kernel<WF, ADT, NT, TBW, WAVES, GATED, IS_G1, PD, true>
                            : gemm_tiled_kernel<WF, ADT, NT, TBW, WAVES, GATED, IS_G1, PD, false>;
    if (lds > 64 * 1024) {
        LKM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    LKM_LAUNCH_GEMM(kern, grid, block, lds, st, pp);
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}

// 16-bit and fp8 (W8A16) weights, 256-row tiles: the LDS-DMA prefill kernel (gemm_prefill.h) when the plan asks for it;
// returns false to fall through to gemm_tiled_kernel
template <typename WFC, typename ADTC>
static bool launch_prefill_if(hipStream_t st, const LaunchCfg& cfg, const GemmParams& p, bool gated, bool is_g1,
                              int max_tiles, int* rc, WFC, ADTC);
// fp8 x fp8, 256-row tiles: weights straight to registers, tokens through a 4-stage LDS ring (gemm_prefill_a8w.h)
template <typename ADTC>
static bool launch_prefill_a8w_if(hipStream_t st, const LaunchCfg& cfg, const GemmParams& p, bool gated, bool is_g1,
                                  int max_tiles, int* rc, ADTC);

// tiled variants built per format: (TM, WAVES, NT) = (64,4,1) (64,8,1) (128,8,1) (128,8,2 non-gated)
// and, for 16-bit weights only, (256,8,1): the prefill tile (weights re-read once per 256 tokens)
template <int WF>
struct W16Only {
    static constexpr bool value = WF == LKM_W_BF16 || WF == LKM_W_F16;
};
#define LKM_TILED_CASE_W16(TBW, WAVES, NT, G, IS1)                                              \
    if constexpr (W16Only<WF_>::value) {                                                        \
        if (cfg.tiled == TBW * 16 && cfg.waves == WAVES && cfg.nt == NT) {                      \
            if (cfg.pd == 4) return launch_tiled_t<WF_, ADT_, NT, TBW, WAVES, G, IS1, 4>(st, p, max_tiles); \
            return launch_tiled_t<WF_, ADT_, NT, TBW, WAVES, G, IS1, 2>(st, p, max_tiles);      \
        }                                                                                       \
    }
// formats gemm_prefill.h serves (16-bit activations): bf16 / fp16, fp8 W8A16, uint4b8 / MXFP4 / NVFP4
template <int WF>
struct PfFormat {
    static constexpr bool value = WF == LKM_W_BF16 || WF == LKM_W_F16 || WF == LKM_W_FP8_E4M3 || WF == LKM_W_INT4_B8 ||
                                  WF == LKM_W_INT4_ZP || WF == LKM_W_MXFP4 || WF == LKM_W_NVFP4;
};
template <int WF>
struct W4Only {
    static constexpr bool value = WF == LKM_W_INT4_B8 || WF == LKM_W_INT4_PS || WF == LKM_W_MXFP4 || WF == LKM_W_NVFP4;
};
#define LKM_TILED_CASE(TBW, WAVES, NT, G, IS1)                                             \
    if (cfg.tiled == TBW * 16 && cfg.waves == WAVES && cfg.nt == NT) {                                  \
        if (cfg.pd >= 4) return launch_tiled_t<WF_, ADT_, NT, TBW, WAVES, G, IS1, 4>(st, p, max_tiles); \
        return launch_tiled_t<WF_, ADT_, NT, TBW, WAVES, G, IS1, 2>(st, p, max_tiles);                 \
    }

#define LKM_DEFINE_TILED_LAUNCHERS(SUFFIX, WF, ADT)                                                   \
    int launch_gemm1_tiled_##SUFFIX(hipStream_t st, const LaunchCfg& cfg, const GemmParams& p,        \
                                    bool gated, int max_tiles) {                                      \
        constexpr int WF_ = WF, ADT_ = ADT;                                                           \
        if constexpr (PfFormat<WF_>::value) {                                                         \
            int rc = LKM_OK;                                                                          \
            if (launch_prefill_if(st, cfg, p, gated, true, max_tiles, &rc, IC<WF_>{}, IC<ADT_>{})) return rc; \
        }                                                                                             \
        if constexpr (WF_ == LKM_W_FP8_A8) {                                                          \
            int rc = LKM_OK;                                                                          \
            if (launch_prefill_a8w_if(st, cfg, p, gated, true, max_tiles, &rc, IC<ADT_>{})) return rc; \
        }                                                                                             \
        if (gated) {                                                                                  \
            LKM_TILED_CASE(2, 4, 1, true, true)                                                       \
            LKM_TILED_CASE(4, 4, 1, true, true)                                                       \
            LKM_TILED_CASE(4, 8, 1, true, true)                                                       \
            LKM_TILED_CASE(8, 8, 1, true, true)                                                       \
            LKM_TILED_CASE_W16(8, 4, 1, true, true)                                                   \
            LKM_TILED_CASE_W16(16, 8, 1, true, true)                                                  \
        } else {                                                                                      \
            LKM_TILED_CASE(2, 4, 1, false, true)                                                      \
            LKM_TILED_CASE(4, 4, 1, false, true)                                                      \
            LKM_TILED_CASE(4, 8, 1, false, true)                                                      \
            LKM_TILED_CASE(8, 8, 1, false, true)                                                      \
            LKM_TILED_CASE(8, 8, 2, false, true)                                                      \
            LKM_TILED_CASE_W16(8, 4, 1, false, true)                                                  \
            LKM_TILED_CASE_W16(16, 8, 1, false, true)                                                 \
        }                                                                                             \
        set_error("gemm1 tiled: variant tm=%d waves=%d nt=%d gated=%d not built", cfg.tiled,          \
                  cfg.waves, cfg.nt, (int)gated);                                                     \
        return LKM_E_INVALID;                                                                         \
    }                                                                                                 \
    int launch_gemm2_tiled_##SUFFIX(hipStream_t st, const LaunchCfg& cfg, const GemmParams& p,        \
                                    int max_tiles) {                                                  \
        constexpr int WF_ = WF, ADT_ = ADT;                                                           \
        if constexpr (PfFormat<WF_>::value) {                                                         \
            int rc = LKM_OK;                                                                          \
            if (launch_prefill_if(st, cfg, p, false, false, max_tiles, &rc, IC<WF_>{}, IC<ADT_>{})) return rc; \
        }                                                                                             \
        if constexpr (WF_ == LKM_W_FP8_A8) {                                                          \
            int rc = LKM_OK;                                                                          \
            if (launch_prefill_a8w_if(st, cfg, p, false, false, max_tiles, &rc, IC<ADT_>{})) return rc;\
        }                                                                                             \
        LKM_TILED_CASE(2, 4, 1, false, false)                                                         \
        LKM_TILED_CASE(4, 4, 1, false, false)                                                         \
        LKM_TILED_CASE(4, 8, 1, false, false)                                                         \
        LKM_TILED_CASE(4, 4, 2, false, false)                                                         \
        LKM_TILED_CASE(8, 8, 1, false, false)                                                         \
        LKM_TILED_CASE(8, 8, 2, false, false)                                                         \
        LKM_TILED_CASE_W16(8, 4, 1, false, false)                                                     \
        LKM_TILED_CASE_W16(16, 8, 1, false, false)                                                    \
        LKM_TILED_CASE_W16(16, 8, 2, false, false)                                                    \
        set_error("gemm2 tiled: variant tm=%d waves=%d nt=%d not built", cfg.tiled, cfg.waves,        \
                  cfg.nt);                                                                            \
        return LKM_E_INVALID;                                                                         \
    }

}  // namespace lkm
