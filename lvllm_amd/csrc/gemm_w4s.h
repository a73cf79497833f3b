// gemm_w4s.h -- the 4-bit decode kernel with TWO independent memory queues per workgroup (round 6): tuning key "pf" = 7.
//
// What rounds 4-5 measured on gemm_w4x.h / gemm_w4e.h (Mixtral int4 M=128 GEMM1, 138-150 us = 0.42-0.44 of the HBM roof) reads,
// in hindsight, as ONE cause: a wave's vector-memory operations retire IN ORDER (vmcnt), so
//   * gemm_w4x.h's waves, which also stage the token rows (L2 hits) through registers, wait for a token load issued AFTER
//     a weight load and thereby for that weight load: whatever the depth of the weight ring, every HBM load has to be back
//     within ~2 K units (the PD = 4 variant was "no faster");
//   * gemm_w4e.h's loader wave carries weights AND tokens in one queue of at most 63 operations = 2 K units ahead.
// Either way a CU keeps 24-32 KiB of weight bytes in flight against the ~48 KiB the chip needs per CU at ~2 us of loaded
// latency, and the SIMD's issue port -- the real bound of the exact decode, ~690 cycles per wave and K unit -- idles half
// of the time.  Here the two streams have their own queues:
//   * wave 0 (loader) moves ONLY token rows, by LDS-DMA, into an S-slot ring of K units (counted vmcnt, one s_barrier per
//     unit: "slot u has landed and everybody is done with slot u-1" -- gemm_w4e.h's protocol);
//   * the NC consumer waves stream ONLY their own weights + scales HBM -> VGPR (what gemm_skinny.h's streamer does at
//     0.84 of the roof) through an R-deep register ring, R-1 units = (R-1) x 2 KiB per wave ahead, the compiler's counted
//     vmcnt on a queue that holds nothing else.  Loads past the end of the K loop go through a NULL buffer descriptor
//     (num_records = 0: they return zeros and move nothing), so every load of the loop is unconditional and the waits stay
//     counted (gemm_tiled.h: one conditional load degrades every wait to vmcnt(0)).
// Math, decoders, weight image, MFMA form (32x32x16, a lane decodes one weight row), epilogue: gemm_w4x.h, bit for bit.
#pragma once
#include <cstdio>
#include <cstdlib>
#include "gemm_w4e.h"

// LKM_W4S_ABL: compile-time timing ablations of this kernel (development libraries only: python -m lvllm_amd.build
// --flag=-DLKM_W4S_ABL=n --only=gemm_w4x_int4; results are wrong): 1 no token DMA, 2 no weight / scale loads, 4 no decode,
// 8 no MFMA, 16 no barriers, 32 no token-fragment reads from the LDS, 64 no epilogue
#ifndef LKM_W4S_ABL
#define LKM_W4S_ABL 0
#endif

namespace lkm {

constexpr int w4s_lcm(int a, int b) {
    int x = a;
    while (x % b) x += a;
    return x;
}

template <int WF, int ADT, int CB, int NC, bool GATED, bool IS_G1, int R, int S>
__global__ __launch_bounds__((NC + 1) * 64) void gemm_w4s_kernel(GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef Dec<WF, ADT> D;
    typedef __attribute__((address_space(3))) void* LdsPtr;
    static_assert(D::LOADS == 1 && D::UNITK == 128 && !D::A8 && !D::XS && !D::UNIT_SCALE, "4-bit formats decoded per row");
    static_assert(S >= 3 && R >= 2, "token ring: the slot being read, the slot in flight, the slot being refilled");
    constexpr int TM = 32 * CB, ROWB = 256, STAGE = TM * ROWB;
    constexpr int UNR = w4s_lcm(R, S);                           // unit u: weight stage u % R, token slot u % S -- both compile-time
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int ti = blockIdx.y, bx = blockIdx.x;
    if (ti >= p.meta[3]) return;
    const int e = p.tile_e[ti], r0 = p.tile_r0[ti];
    const int m_e = p.counts[e], off_e = p.offsets[e];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0 = token loader, 1..NC = consumers
    const bool pairs = IS_G1 && GATED;                          // gate tile + its up tile; else two consecutive tiles
    const int T_all = p.T_half * p.halves;
    const int sk = IS_G1 ? 0 : blockIdx.z;
    const int u0 = IS_G1 ? 0 : (int)((long long)sk * p.U / p.SK);
    const int U = IS_G1 ? p.U : (int)((long long)(sk + 1) * p.U / p.SK) - u0;
    const int k_base = u0 * 128;
    const int rows_here = m_e - r0 < TM ? m_e - r0 : TM;
    const bool two_blocks = CB == 2 && rows_here > 32;           // 32-token column blocks that hold rows (workgroup-uniform)

    if (wave == 0) {
        // ================================================================== token loader (gemm_w4e.h's, tokens only)
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((const char*)p.x + (size_t)k_base * 2), 0, (int)0xffffffffu, 0x00020000);
        // instruction d moves rows d*4 .. d*4+3 (lane L: row d*4 + L/16, physical slot L%16 = logical ^ (row%16))
        constexpr int XI = TM / 4;
        int xv[XI];
#pragma unroll
        for (int d = 0; d < XI; ++d) {
            const int row = d * 4 + (lane >> 4), pslot = lane & 15;
            const int lslot = pslot ^ (row & 15);
            const int r = r0 + row;
            const int rr = r < m_e ? r : r0;
            const int src = IS_G1 ? p.sorted_slot[off_e + rr] / p.top_k : off_e + rr;
            xv[d] = (int)((unsigned)src * (unsigned)p.ldx * 2u + (unsigned)lslot * 16u);
        }
        auto run_loader = [&](auto CBC) __attribute__((always_inline)) {
            constexpr int IPU = decltype(CBC)::v * 8;            // DMA instructions per slot: the counted waits rely on it
            static_assert((S - 1) * IPU < 64, "vmcnt range");
            auto dma = [&](int u) __attribute__((always_inline)) {
                char* base = lds + (u % S) * STAGE;
#pragma unroll
                for (int d = 0; d < IPU; ++d)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (LdsPtr)(base + d * 1024), 16, xv[d], u * ROWB, 0, 0);
            };
            if constexpr ((LKM_W4S_ABL & 1) != 0) {
                for (int u = 0; u < U; ++u)
                    if constexpr (!(LKM_W4S_ABL & 16)) __builtin_amdgcn_s_barrier();
                return;
            }
#pragma unroll
            for (int s = 0; s < S - 1; ++s)
                if (s < U) dma(s);
            for (int u = 0; u < U; ++u) {
                const int younger = U - 1 - u;                    // slots issued after slot u (at most S - 2 of them)
                bool waited = false;
                static_for<S - 1>([&](auto YC) __attribute__((always_inline)) {
                    constexpr int y = S - 2 - decltype(YC)::v;    // S-2, S-3, .., 0, (-1 unused)
                    if constexpr (y >= 0) {
                        if (!waited && younger >= y) {
                            waited = true;
                            w4e_wait_vmcnt<y * IPU>();
                        }
                    }
                });
                if constexpr (!(LKM_W4S_ABL & 16)) __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (u + S - 1 < U) dma(u + S - 1);
            }
            w4e_wait_vmcnt<0>();
        };
        if constexpr (CB == 2) {
            if (two_blocks) run_loader(IC<2>{});
            else run_loader(IC<1>{});
        } else {
            run_loader(IC<1>{});
        }
        return;
    }

    // ====================================================================== consumers
    const int cw = wave - 1;                                     // consumer index = 32-row group inside the workgroup
    const int j = lane & 31, h = lane >> 5, i16 = lane & 15, sel = (lane >> 4) & 1;
    const int grp = bx * NC + cw;
    const int tile_lo = pairs ? grp : 2 * grp;
    const bool wave_on = tile_lo < p.T_half;                     // tail group of a padded tile count: streams tile 0, stores nothing
    const int my_tile = wave_on ? (pairs ? (sel ? p.T_half + grp : grp) : 2 * grp + sel) : 0;
    const int dparam = WF == LKM_W_NVFP4 ? __builtin_bit_cast(int, p.gs ? p.gs[e] : 1.0f) : p.spu;
    // weights: this expert's image behind a buffer descriptor (uniform base in SGPRs, 32-bit lane offsets, the unit's
    // byte offset as the scalar offset); the NULL descriptor serves the ring's loads past the last unit
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.w + (size_t)e * p.w_estride * 16), 0, (int)0xffffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_null = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0, 0x00020000);
    const int wstep = (int)(p.w_ustride * 16);
    int woff[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
        woff[q] = (int)((my_tile * p.w_tstride + (long long)u0 * p.w_ustride + (2 * h + q) * 16 + i16) * 16);
    const char* sbase = D::aux_ptr(p.s, (size_t)e * T_all * p.U, 0, p.spu);
    const unsigned aoff = (unsigned)(size_t)D::aux_ptr((const void*)0, (size_t)my_tile * p.U + u0, lane, p.spu);
    const unsigned astep = (unsigned)D::aux_step(p.spu);
    // (scales of this expert behind a descriptor as well: uint4b8 with one group per unit needs ONE 16-bit scale per lane
    //  and unit -- a naturally aligned buffer_load_ushort at (lane offset) + (scalar unit offset), no 64-bit address math)
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)sbase, 0, (int)0xffffffffu, 0x00020000);
    int baddr[4][2];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int q = 0; q < 2; ++q) baddr[s][q] = j * ROWB + (((s * 4 + 2 * h + q) ^ (j & 15)) * 16);

    f32x16 acc[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;

    auto run_c = [&](auto CBC, auto HC) __attribute__((always_inline)) {
        constexpr int CBR = decltype(CBC)::v;
        constexpr bool HOIST = decltype(HC)::v != 0;            // int4, one scale group per unit: multipliers once per unit
        struct WStage {
            u32x4 w[2][1];
            typename D::Aux aux;
            unsigned short a16;      // HOIST: the unit's one scale as loaded (widened where it is used, not where it lands)
        };
        WStage ws[R];
        // unconditional: a unit past the end reads through the NULL descriptor (zeros, no traffic) and re-reads the last
        // unit's scales (a few bytes from the L2)
        auto load_w = [&](WStage& s, int u) __attribute__((always_inline)) {
            if constexpr ((LKM_W4S_ABL & 2) != 0) {
                s.w[0][0] = s.w[1][0] = u32x4{0x12345678u + (unsigned)u, 0x9abcdef0u, 0x0fedcba9u, 0x87654321u};
                s.a16 = 0x3c00;
                asm volatile("" : "+v"(s.w[0][0]), "+v"(s.w[1][0]));
                return;
            }
            const bool in = u < U;
            const __amdgpu_buffer_rsrc_t rs = in ? rs_w : rs_null;
            const int uc = in ? u : U - 1;
#pragma unroll
            for (int q = 0; q < 2; ++q)
                s.w[q][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, woff[q], u * wstep, 2));
            if constexpr (HOIST && WF == LKM_W_INT4_B8)
                s.a16 = __builtin_amdgcn_raw_buffer_load_b16(rs_a, (int)aoff, uc * (int)astep, 0);
            else
                D::load_aux_at(s.aux, sbase + (size_t)uc * astep + aoff);
        };
        auto unit = [&](const WStage& s, const char* sb) __attribute__((always_inline)) {
            typename W4Int4<WF, ADT>::M mu;
            if constexpr (HOIST && WF == LKM_W_INT4_B8) {
                typename D::Aux ax;
                ax.raw = u32x2{(unsigned)s.a16, 0u};
                mu = W4Int4<WF, ADT>::mult(ax);
            } else if constexpr (HOIST) {
                mu = W4Int4<WF, ADT>::mult(s.aux);
            }
            auto dec = [&](int s_, int q_) __attribute__((always_inline)) {
                if constexpr ((LKM_W4S_ABL & 4) != 0) return s.w[q_][0] + u32x4{(unsigned)s_, 0u, 0u, 0u};
                else if constexpr (HOIST) return W4Int4<WF, ADT>::template frag<0>(s.w[q_], s_, mu);
                else return D::frag(s.w[q_], s.aux, s_, dparam);
            };
            u32x4 bf[2][2][CBR];                                 // [parity of s][q][column block]
            auto ldb = [&](int s_) __attribute__((always_inline)) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int c = 0; c < CBR; ++c) {
                        if constexpr ((LKM_W4S_ABL & 32) != 0) bf[s_ & 1][q][c] = u32x4{(unsigned)baddr[s_][q], 1u, 2u, (unsigned)c};
                        else bf[s_ & 1][q][c] = *(const u32x4*)(sb + c * 32 * ROWB + baddr[s_][q]);
                    }
            };
            ldb(0);
            u32x4 a = dec(0, 0);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int s_ = t >> 1, q_ = t & 1;
                if (q_ == 0 && s_ + 1 < 4) ldb(s_ + 1);
                u32x4 an = a;
                if (t + 1 < 8) an = dec((t + 1) >> 1, (t + 1) & 1);
#pragma unroll
                for (int c = 0; c < CBR; ++c) {
                    if constexpr ((LKM_W4S_ABL & 8) != 0) acc[c][0] += __builtin_bit_cast(float, a.x ^ bf[s_ & 1][q_][c].x);
                    else acc[c] = Mfma32<ADT>::run(a, bf[s_ & 1][q_][c], acc[c]);
                }
                a = an;
            }
        };
        // prologue in the steady state's issue order (weights, weights, scales per unit): the waits of the first loop
        // iteration are derived from it
#pragma unroll
        for (int s = 0; s < R - 1; ++s) {
            load_w(ws[s], s);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int ub = 0; ub < U; ub += UNR) {
            static_for<UNR>([&](auto SC) __attribute__((always_inline)) {
                constexpr int st = decltype(SC)::v;
                const int u = ub + st;
                load_w(ws[(st + R - 1) % R], u + R - 1);
                if (u < U) {                                     // (wave-uniform; every wave of the workgroup counts the same barriers)
                    if constexpr (!(LKM_W4S_ABL & 16)) __builtin_amdgcn_s_barrier();
                    if (wave_on) unit(ws[st % R], lds + (st % S) * STAGE);
                }
            });
        }
    };
    auto run = [&](auto CBC) __attribute__((always_inline)) {
        if constexpr (WF == LKM_W_INT4_B8) {
            if (p.spu <= 1) return run_c(CBC, IC<1>{});
        }
        run_c(CBC, IC<0>{});
    };
    if constexpr (CB == 2) {
        if (two_blocks) run(IC<2>{});
        else run(IC<1>{});
    } else {
        run(IC<1>{});
    }

    // epilogue: as gemm_w4x_kernel (D layout of the 32x32 MFMA)
    if (!wave_on) return;
    if constexpr ((LKM_W4S_ABL & 64) != 0) {
#pragma unroll
        for (int c = 0; c < CB; ++c) asm volatile("" ::"v"(acc[c]));
        return;
    }
    static_for<CB>([&](auto CC) __attribute__((always_inline)) {
        constexpr int c = decltype(CC)::v;
        const int r_tok = r0 + c * 32 + j;
        if (r_tok < m_e) {
            static_for<2>([&](auto RC) __attribute__((always_inline)) {
                constexpr int rr = decltype(RC)::v;
                const f32x4 lo = {acc[c][rr * 4 + 0], acc[c][rr * 4 + 1], acc[c][rr * 4 + 2], acc[c][rr * 4 + 3]};
                const f32x4 hi = {acc[c][8 + rr * 4 + 0], acc[c][8 + rr * 4 + 1], acc[c][8 + rr * 4 + 2], acc[c][8 + rr * 4 + 3]};
                const int nsub = rr * 8 + h * 4;
                if constexpr (IS_G1 && GATED) {
                    const int n = grp * 16 + nsub;
                    if (n < p.n_real) store_gemm1_frag<ADT, true>(p, lo, hi, (size_t)(off_e + r_tok), n);
                } else if constexpr (IS_G1) {
                    const int n0 = (2 * grp) * 16 + nsub, n1 = n0 + 16;
                    if (n0 < p.n_real) store_gemm1_frag<ADT, false>(p, lo, lo, (size_t)(off_e + r_tok), n0);
                    if (n1 < p.n_real) store_gemm1_frag<ADT, false>(p, hi, hi, (size_t)(off_e + r_tok), n1);
                } else {
                    const int n0 = (2 * grp) * 16 + nsub, n1 = n0 + 16;
                    if (n0 < p.n_real) store_gemm2_frag(p, lo, sk, (size_t)(off_e + r_tok), n0);
                    if (n1 < p.n_real) store_gemm2_frag(p, hi, sk, (size_t)(off_e + r_tok), n1);
                }
            });
        }
    });
#else
    (void)p;
#endif
}

template <int WF, int ADT, int CB, int NC, bool GATED, bool IS_G1, int R, int S>
static int launch_w4s_t(hipStream_t st, const GemmParams& p, int max_tiles) {
    constexpr size_t lds = (size_t)S * CB * 32 * 256;
    const int groups = (IS_G1 && GATED) ? p.T_half : p.T_half / 2;
    dim3 grid(ceil_div(groups, NC), max_tiles, IS_G1 ? 1 : p.SK), block((NC + 1) * 64);
    auto kern = gemm_w4s_kernel<WF, ADT, CB, NC, GATED, IS_G1, R, S>;
    if (lds > 64 * 1024) LKM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    static const bool dbg_occ = getenv("LKM_DEBUG_OCC") != nullptr;      // (development: what the occupancy API says about this variant)
    if (dbg_occ) {
        int nb = -1;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, (NC + 1) * 64, lds);
        fprintf(stderr, "[w4s] CB=%d NC=%d R=%d S=%d lds=%zu: occupancy API -> %d workgroups per CU (%s)\n", CB, NC, R, S, lds, nb, hipGetErrorString(e));
    }
    LKM_LAUNCH_GEMM(kern, grid, block, lds, st, p);
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}

// cfg.pf == 7 selects the kernel; cfg.tiled 32 / 64 -> one / two token column blocks; cfg.pd = depth R of the weight
// register ring (6 default, 4), the token ring has 3 slots when 3 divides R, else 4; cfg.waves = consumer waves per
// workgroup (4 default, 7 at 64-row tiles)
template <int WF, int ADT>
static bool launch_w4s_if(hipStream_t st, const LaunchCfg& cfg, const GemmParams& p, bool gated, bool is_g1, int max_tiles, int* rc) {
    if (cfg.pf != 7 || (cfg.tiled != 32 && cfg.tiled != 64) || !w4x_ok(p)) return false;
    const int cb = cfg.tiled / 32;
    const int r = cfg.pd == 4 ? 4 : 6;
    const int nc = cfg.waves == 7 ? 7 : 4;
#define LKM_W4S_1(CB_, NC_, G_, IS1_, R_, S_)                                                       \
    if (cb == CB_ && nc == NC_ && r == R_) {                                                         \
        *rc = launch_w4s_t<WF, ADT, CB_, NC_, G_, IS1_, R_, S_>(st, p, max_tiles);                   \
        return true;                                                                                 \
    }
    // built: ring depths 6 / 4, four or seven consumers (3 / 8 deep and eight consumers measured and dropped:
    // profiles/r06_int4_w4s_ab.log, r06_w4s_nc7.log)
#define LKM_W4S_R(CB_, NC_, G_, IS1_) LKM_W4S_1(CB_, NC_, G_, IS1_, 6, 3) LKM_W4S_1(CB_, NC_, G_, IS1_, 4, 4)
#define LKM_W4S_ALL(G_, IS1_) LKM_W4S_R(1, 4, G_, IS1_) LKM_W4S_R(2, 4, G_, IS1_) LKM_W4S_R(2, 7, G_, IS1_)
    if (is_g1 && gated) { LKM_W4S_ALL(true, true) }
    else if (is_g1) { LKM_W4S_ALL(false, true) }
    else { LKM_W4S_ALL(false, false) }
#undef LKM_W4S_ALL
#undef LKM_W4S_R
#undef LKM_W4S_1
    return false;
}

}  // namespace lkm
