// gemm_w4dma.h -- per-expert grouped GEMMs for the 4-bit weight formats at decode batch sizes (tens of rows per
// expert): the math and the work decomposition of gemm_tiled.h (4 waves x NT weight tiles x one 32/64-token tile,
// tokens shared through LDS), with a different DATA PATH.
//
// Why: a 128-k unit of a 4-bit weight tile is only 1 KiB.  gemm_tiled_kernel keeps one unit per tile in flight in
// VGPRs (2 KiB per wave at two tiles); at the 8-12 resident waves its registers allow that is ~24 KiB of weight
// bytes in flight per CU -- against ~1.9 us of loaded HBM latency, 3.3 TB/s chip-wide, which is what it measured
// (41 % of the HBM roof at Mixtral int4 M=128, issue slots to spare).  Deeper register rings cost resident waves.
// Here every operand reaches LDS by LDS-DMA (buffer_load ... lds): weights, their scales, the token rows, the
// per-token scalars of the int4 fast mode.  A DMA in flight costs no register, so the ring is DEPTH units deep
// (DEPTH - 1 units = 3-4x the bytes in flight per CU) and the kernel still needs only ONE barrier per unit:
//     wait vmcnt((DEPTH-2) x loads per unit)   -- my share of unit u has landed, younger units stay in flight
//     barrier                                 -- everybody's share has; everybody is done reading unit u-1
//     DMA of unit u+DEPTH-1 into the stage of unit u-1
//     fragments of unit u: ds_read -> in-register decode -> MFMA (as gemm_tiled.h)
// No ordinary vector-memory load exists in the loop (one would turn every counted wait into vmcnt(0)).
#pragma once
#include "gemm_tiled.h"

namespace lkm {

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int WF, int ADT, int NT, int TBW, bool GATED, bool IS_G1, int DEPTH>
__global__ __launch_bounds__(256) void gemm_w4dma_kernel(GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef Dec<WF, ADT> D;
    typedef __attribute__((address_space(3))) void* LdsPtr;
    static_assert(D::LOADS == 1 && D::UNITK == 128 && !D::A8, "4-bit formats: one KiB per (tile, unit), 16-bit activations");
    constexpr int WAVES = 4, THREADS = 256;
    constexpr int NTT = (IS_G1 && GATED) ? 2 * NT : NT;
    constexpr int TM = TBW * 16, ROWB = 256, SLOTS = 16;
    constexpr int XBYTES = TM * ROWB, XSBYTES = D::XS ? TM * 4 : 0;
    constexpr int WOFF = XBYTES + XSBYTES, WBYTES = WAVES * NTT * 1024;
    constexpr int AOFF = WOFF + WBYTES, ASLOT = 128, ABYTES = WAVES * NTT * ASLOT;
    constexpr int STAGE = AOFF + ABYTES;
    extern __shared__ __attribute__((aligned(16))) char lds[];

    int ti = blockIdx.y, bx = blockIdx.x;
    if (ti >= p.meta[3]) return;
    const int e = p.tile_e[ti], r0 = p.tile_r0[ti];
    const int m_e = p.counts[e], off_e = p.offsets[e];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int tile0 = (bx * WAVES + wave) * NT;
    const bool wave_on = tile0 < p.T_half;            // tail group of a padded tile count: streams tile 0, stores nothing
    const int T_all = p.T_half * p.halves;
    const int dparam = WF == LKM_W_NVFP4 ? __builtin_bit_cast(int, p.gs ? p.gs[e] : 1.0f) : p.spu;
    const int sk = IS_G1 ? 0 : blockIdx.z;
    const int u0 = IS_G1 ? 0 : (int)((long long)sk * p.U / p.SK);
    const int U = IS_G1 ? p.U : (int)((long long)(sk + 1) * p.U / p.SK) - u0;
    const int k_base = u0 * 128;

    // ---- DMA descriptors
    const int auxB = D::aux_step(p.spu);                                  // scale bytes per (tile, unit)
    const int aux_lane = (int)(D::aux_ptr((const void*)0, 0, lane, p.spu) - (const char*)0);   // this lane's offset in them
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.w + (size_t)e * T_all * p.U * 1024), 0, (int)((size_t)T_all * p.U * 1024), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.s + (size_t)e * T_all * p.U * auxB), 0, (int)((size_t)T_all * p.U * auxB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_xs = __builtin_amdgcn_make_buffer_rsrc((void*)p.xscale, 0, 0x7fffffff, 0x00020000);
    int woff[NTT], aoff[NTT];
#pragma unroll
    for (int t = 0; t < NTT; ++t) {
        const int tile = (IS_G1 && GATED && t >= NT) ? p.T_half + tile0 + (t - NT) : tile0 + t;
        const int tl = wave_on ? tile : 0;
        woff[t] = __builtin_amdgcn_readfirstlane((int)((tl * p.w_tstride + u0 * p.w_ustride) * 16));
        aoff[t] = __builtin_amdgcn_readfirstlane((tl * p.U + u0) * auxB);
    }
    constexpr int PIECES = TM * SLOTS / THREADS;
    int xv[PIECES];
#pragma unroll
    for (int q = 0; q < PIECES; ++q) {
        const int pc = q * THREADS + tid;
        const int row = pc / SLOTS, pslot = pc % SLOTS;
        const int lslot = pslot ^ x_swizzle<ROWB>(row);
        const int r = r0 + row;
        const int rr = r < m_e ? r : r0;
        const int src_row = IS_G1 ? p.sorted_slot[off_e + rr] / p.top_k : off_e + rr;
        xv[q] = (src_row * p.ldx + k_base) * 2 + lslot * 16;
    }
    int xsv = 0;
    if (D::XS) {
        const int row = wave * (TM / 4) + (lane < TM / 4 ? lane : 0);
        const int r = r0 + row;
        const int rr = r < m_e ? r : r0;
        const int src_row = IS_G1 ? p.sorted_slot[off_e + rr] / p.top_k : off_e + rr;
        xsv = (src_row * p.ld_xscale + u0) * 4;
    }

    f32x4 acc[NTT][TBW];
#pragma unroll
    for (int t = 0; t < NTT; ++t)
#pragma unroll
        for (int b = 0; b < TBW; ++b) acc[t][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto run_h = [&](auto NBC, auto HC) __attribute__((always_inline)) {
        constexpr int NB = decltype(NBC)::v;
        constexpr bool HOIST = decltype(HC)::v != 0;
        constexpr int PCS = NB * 16 * SLOTS / THREADS;                    // token pieces this thread moves per unit
        static_assert(PCS >= 1 && (NB * 16 * SLOTS) % THREADS == 0, "block granularity");
        constexpr int IPU = PCS + 2 * NTT + (D::XS ? 1 : 0);              // LDS-DMA instructions per wave and unit
        static_assert((DEPTH - 2) * IPU < 64, "vmcnt range");

        auto dma = [&](int u) __attribute__((always_inline)) {
            char* base = lds + (u % DEPTH) * STAGE;
#pragma unroll
            for (int q = 0; q < PCS; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (LdsPtr)(base + (q * THREADS + wave * 64) * 16), 16, xv[q],
                                                         u * ROWB, 0, 0);
            if constexpr (D::XS) {
                if (lane < TM / 4)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_xs, (LdsPtr)(base + XBYTES + wave * (TM / 4) * 4), 4, xsv,
                                                             u * 4, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < NTT; ++t)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (LdsPtr)(base + WOFF + (wave * NTT + t) * 1024), 16,
                                                         lane * 16, woff[t] + u * (int)(p.w_ustride * 16), 0, 2);
#pragma unroll
            for (int t = 0; t < NTT; ++t)
                if (lane * 4 < auxB)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (LdsPtr)(base + AOFF + (wave * NTT + t) * ASLOT), 4,
                                                             lane * 4, aoff[t] + u * auxB, 0, 0);
        };

        struct WStage {
            u32x4 w[NTT][1];
            typename D::Aux aux[NTT];
        };
        auto compute = [&](int u) __attribute__((always_inline)) {
            const char* xb = lds + (u % DEPTH) * STAGE;
            WStage s;
#pragma unroll
            for (int t = 0; t < NTT; ++t) {
                s.w[t][0] = *(const u32x4*)(xb + WOFF + (wave * NTT + t) * 1024 + lane * 16);
                D::load_aux_at(s.aux[t], xb + AOFF + (wave * NTT + t) * ASLOT + aux_lane);
            }
            if (!wave_on) return;
            if constexpr (D::UNIT_SCALE) {           // int4 fast mode: scale and bias correction on the unit's partial sums
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
                    const f32x2 c = splat2_opaque(D::BIAS8 * *(const float*)(xb + XBYTES + (b * 16 + j) * 4));
#pragma unroll
                    for (int t = 0; t < NTT; ++t) acc[t][b] += s.aux[t].s * sub4(part[t][b], c);
                }
            } else if constexpr (WF == LKM_W_INT4_B8) {
                // bit-exact decode (~16-19 VALU per fragment): decode k-step ks+1 under the MFMAs of k-step ks
                u32x4 a[2][NTT];
                typename D::Mult mu[NTT];
                auto dec = [&](int t, int ks) __attribute__((always_inline)) {
                    if constexpr (HOIST) return D::frag_m(s.w[t], ks, mu[t]);
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
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, DEC_PER, 0);
                        }
                    }
                }
            } else {                                 // MXFP4 / NVFP4: the scaled conversions are the decode
#pragma unroll
                for (int ks = 0; ks < D::KSTEPS; ++ks) {
                    u32x4 a[NTT], bf[NB];
#pragma unroll
                    for (int t = 0; t < NTT; ++t) a[t] = D::frag(s.w[t], s.aux[t], ks, dparam);
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const int row = b * 16 + j;
                        bf[b] = *(const u32x4*)(xb + row * ROWB + (((ks * 4 + g) ^ x_swizzle<ROWB>(row)) * 16));
                    }
#pragma unroll
                    for (int t = 0; t < NTT; ++t)
#pragma unroll
                        for (int b = 0; b < NB; ++b) acc[t][b] = ActT<ADT>::mfma(a[t], bf[b], acc[t][b]);
                }
            }
        };

#pragma unroll
        for (int s = 0; s < DEPTH - 1; ++s)
            if (s < U) dma(s);
        for (int u = 0; u < U; ++u) {
            const int younger = U - 1 - u;                // units issued after unit u that may stay in flight
            if (younger >= DEPTH - 2) wait_vmcnt<(DEPTH - 2) * IPU>();
            else if (DEPTH > 3 && younger == 1) wait_vmcnt<IPU>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (u + DEPTH - 1 < U) dma(u + DEPTH - 1);
            compute(u);
        }
    };
    auto run = [&](auto NBC) __attribute__((always_inline)) {
        if constexpr (WF == LKM_W_INT4_B8) {
            if (p.spu <= 1) return run_h(NBC, IC<1>{});
        }
        run_h(NBC, IC<0>{});
    };
    {
        const int rows_here = m_e - r0 < TM ? m_e - r0 : TM;
        const int nb = (rows_here + 15) >> 4;
        if constexpr (TBW == 4) {
            if (nb <= 1) run(IC<1>{});
            else if (nb == 2) run(IC<2>{});
            else if (nb == 3) run(IC<3>{});
            else run(IC<4>{});
        } else {
            static_assert(TBW == 2, "32- or 64-row tiles");
            if (nb <= 1) run(IC<1>{});
            else run(IC<2>{});
        }
    }
    wait_vmcnt<0>();

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
#else
    (void)p;
#endif
}

// usable when K is a whole number of 128-k units (a ragged tail would need the zero fill of the register path) and
// the operand matrices fit 2 GiB buffer windows
inline bool w4dma_ok(const GemmParams& p) {
    return p.Kreal % 128 == 0 && (size_t)p.x_rows * (size_t)p.ldx * 2 < (size_t)0x7fffffff &&
           (size_t)p.T_half * p.halves * p.U * 1024 < (size_t)0x7fffffff;
}

template <int WF, int ADT, int NT, int TBW, bool GATED, bool IS_G1, int DEPTH>
static int launch_w4dma_t(hipStream_t st, const GemmParams& p, int max_tiles) {
    typedef Dec<WF, ADT> D;
    constexpr int NTT = (IS_G1 && GATED) ? 2 * NT : NT;
    constexpr size_t stage = (size_t)TBW * 16 * 256 + (D::XS ? TBW * 16 * 4 : 0) + 4 * NTT * 1024 + 4 * NTT * 128;
    constexpr size_t lds = stage * DEPTH;
    dim3 grid(ceil_div(p.T_half, 4 * NT), max_tiles, IS_G1 ? 1 : p.SK), block(256);
    auto kern = gemm_w4dma_kernel<WF, ADT, NT, TBW, GATED, IS_G1, DEPTH>;
    if (lds > 64 * 1024) LKM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, grid, block, lds, st, p);
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}

// (tile rows, nt) variants: GEMM1 gated one gate + one up tile per wave, everything else two tiles per wave
template <int WF, int ADT>
static bool launch_w4dma_if(hipStream_t st, const LaunchCfg& cfg, const GemmParams& p, bool gated, bool is_g1,
                            int max_tiles, int* rc) {
    if (cfg.pf != 4 || cfg.waves != 4 || (cfg.tiled != 32 && cfg.tiled != 64) || !w4dma_ok(p)) return false;
    // ring depth by LDS budget: two workgroups per CU (2 x 80 KiB); every variant below has two tiles per wave
    const int ntt = (is_g1 && gated) ? 2 * cfg.nt : cfg.nt;
    const size_t stage = (size_t)cfg.tiled * 256 + (Dec<WF, ADT>::XS ? cfg.tiled * 4 : 0) + 4 * ntt * (1024 + 128);
    const int depth = cfg.pd == 3 ? 3 : (4 * stage <= 80 * 1024 ? 4 : 3);
#define LKM_W4(NT_, TBW_, G_, IS1_)                                                                    \
    *rc = depth == 4 ? launch_w4dma_t<WF, ADT, NT_, TBW_, G_, IS1_, 4>(st, p, max_tiles)               \
                     : launch_w4dma_t<WF, ADT, NT_, TBW_, G_, IS1_, 3>(st, p, max_tiles);              \
    return true;
    if (is_g1 && gated && cfg.nt == 1) {
        if (cfg.tiled == 32) { LKM_W4(1, 2, true, true) }
        LKM_W4(1, 4, true, true)
    }
    if (is_g1 && gated && cfg.nt == 2) {       // two gate + two up tiles per wave: half the token-fragment LDS reads per weight byte
        if (cfg.tiled == 32) { LKM_W4(2, 2, true, true) }
        LKM_W4(2, 4, true, true)
    }
    if (is_g1 && !gated && cfg.nt == 2) {
        if (cfg.tiled == 32) { LKM_W4(2, 2, false, true) }
        LKM_W4(2, 4, false, true)
    }
    if (is_g1 && !gated && cfg.nt == 1) {
        if (cfg.tiled == 32) { LKM_W4(1, 2, false, true) }
        LKM_W4(1, 4, false, true)
    }
    if (!is_g1 && cfg.nt == 2) {
        if (cfg.tiled == 32) { LKM_W4(2, 2, false, false) }
        LKM_W4(2, 4, false, false)
    }
    if (!is_g1 && cfg.nt == 1) {
        if (cfg.tiled == 32) { LKM_W4(1, 2, false, false) }
        LKM_W4(1, 4, false, false)
    }
#undef LKM_W4
    return false;
}

}  // namespace lkm
