// gemm_w4x.h -- per-expert grouped GEMMs for the 4-bit weight formats at decode batch sizes (tens of rows per expert),
// round 4: the 32 x 32 x 16 MFMA form of gemm_tiled.h's 4-bit path.
//
// Why ("4-bit decode"): the 4-bit tile kernels are bound by the SIMD's ONE vector issue port -- every
// weight byte costs ~19 decode instructions per 8 weights (exact uint4b8) and every v_mfma_f32_16x16x32 eight port
// cycles.  This kernel spends fewer port cycles per weight byte:
//   * one v_mfma_f32_32x32x16 per (32 weight rows, 32 tokens, 16 k) instead of four 16x16x32 ones per (2 x 16 rows,
//     2 x 16 tokens, 32 k): half the MFMA issue slots, and each of them leaves six free VALU slots under its 32 matrix
//     cycles instead of two;
//   * a lane decodes ONE weight row (row = lane mod 32; lanes 0-15 the gate tile, 16-31 the up tile of a gated GEMM1,
//     else two consecutive tiles), so the group scale and its two multipliers are formed once per lane and K unit, not
//     once per tile, and one scale load serves both tiles;
//   * weights, scales and token rows are addressed as (uniform 64-bit base in SGPRs) + (32-bit lane offset): the K loop
//     advances scalars, not per-lane 64-bit pointers.
// The weight image is the one repack.hip already writes ([tile][unit][lane = g*16 + i][16 B], dword s = k-step s):
// lane (row i of tile t, half h = lane / 32) loads the two 16-byte pieces of old lanes (2h, i) and (2h+1, i); piece q,
// dword s holds k = s*32 + (2h+q)*8 .. +7 -- an A operand of the 16-k MFMA whose B operand is the 16 bytes at byte
// s*64 + h*32 + q*16 of the token row.  Any A/B-consistent k assignment is valid (the MFMA sums over k).
// Math and rounding points: gemm_skinny.h (fused_moe.py:64-295: b = T((nib-8)*scale), fp32 accumulate); the decoders
// are gemm_skinny.h's Dec<>, bit for bit.
#pragma once
#include "gemm_tiled.h"

namespace lkm {

// uint4b8, exact decode without packed-fp32 instructions (DECV = 1): a v_pk_fma_f32 does not issue beside an MFMA at all
// and costs 2-3 slots after one (tools/probe_mfma_valu.py), a v_fma_f32 hides under it.  Same bits as Dec<>::frag_m.
template <int ADT>
__device__ __forceinline__ u32x4 int4_frag_plain(unsigned w, float s512, float m8) {
    const unsigned lo = w & 0x0f0f0f0fu, hi = (w >> 4) & 0x0f0f0f0fu;
    const f32x2 e01 = __builtin_amdgcn_cvt_pk_f32_fp8(lo, false), e23 = __builtin_amdgcn_cvt_pk_f32_fp8(lo, true);
    const f32x2 o01 = __builtin_amdgcn_cvt_pk_f32_fp8(hi, false), o23 = __builtin_amdgcn_cvt_pk_f32_fp8(hi, true);
    float r[8];
    const float v[8] = {e01.x, o01.x, e01.y, o01.y, e23.x, o23.x, e23.y, o23.y};
#pragma unroll
    for (int i = 0; i < 8; ++i) asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r[i]) : "v"(v[i]), "v"(s512), "v"(m8));   // (asm: the SLP vectoriser would re-pack them)
    u32x4 o;
    o.x = ActT<ADT>::pack2(r[0], r[1]);
    o.y = ActT<ADT>::pack2(r[2], r[3]);
    o.z = ActT<ADT>::pack2(r[4], r[5]);
    o.w = ActT<ADT>::pack2(r[6], r[7]);
    return o;
}

// What only uint4b8 has -- the multipliers of a (row, scale group) formed once per K unit when the group covers the unit
// (HOIST), the packed-fp32-free decoder, aligned LDS reads of its scales -- behind one interface that exists for every
// 4-bit format, so that the kernels' `if constexpr` branches type-check whatever WF is.
template <int WF, int ADT>
struct W4Int4 {
    struct M {};
    typedef typename Dec<WF, ADT>::Aux Aux;
    static __device__ __forceinline__ M mult(const Aux&) { return M{}; }
    template <int DECV>
    static __device__ __forceinline__ u32x4 frag(const u32x4 (&w)[1], int, const M&) { return w[0]; }
    static __device__ __forceinline__ void load_aux_lds(Aux& a, const char* p, int, bool) { Dec<WF, ADT>::load_aux_at(a, p); }
};
template <int ADT>
struct W4Int4<LKM_W_INT4_B8, ADT> {
    typedef Dec<LKM_W_INT4_B8, ADT> D;
    typedef typename D::Mult M;
    typedef typename D::Aux Aux;
    static __device__ __forceinline__ M mult(const Aux& a) { return D::mult(a, 0, 0); }
    template <int DECV>
    static __device__ __forceinline__ u32x4 frag(const u32x4 (&w)[1], int s, const M& m) {
        if constexpr (DECV == 1) return int4_frag_plain<ADT>(w[0][s], m.s512.x, m.m8.x);
        else return D::frag_m(w, s, m);
    }
    // aligned reads of what Dec<>::load_aux_at fetches with one unaligned 8-byte load (1 / 2 / 4 scales per unit)
    static __device__ __forceinline__ void load_aux_lds(Aux& a, const char* p, int spu, bool hoist) {
        if (hoist) a.raw = u32x2{(unsigned)*(const unsigned short*)p, 0u};
        else if (spu == 2) a.raw = u32x2{*(const unsigned*)p, 0u};
        else a.raw = *(const u32x2*)p;
    }
};

// uint4 with zero points (LKM_W_INT4_ZP): the same, the multipliers from a (scale, zero point) pair
template <int ADT>
struct W4Int4<LKM_W_INT4_ZP, ADT> {
    typedef Dec<LKM_W_INT4_ZP, ADT> D;
    typedef typename D::Mult M;
    typedef typename D::Aux Aux;
    static __device__ __forceinline__ M mult(const Aux& a) { return D::mult(a, 0, 0); }
    template <int DECV>
    static __device__ __forceinline__ u32x4 frag(const u32x4 (&w)[1], int s, const M& m) {
        return Dec<LKM_W_INT4_B8, ADT>::frag_m(w, s, m);
    }
    // aligned reads of the 1 / 2 / 4 pairs of a unit (4 bytes each)
    static __device__ __forceinline__ void load_aux_lds(Aux& a, const char* p, int spu, bool hoist) {
        if (hoist) a.raw = u32x4{*(const unsigned*)p, 0u, 0u, 0u};
        else if (spu == 2) a.raw = u32x4{((const unsigned*)p)[0], ((const unsigned*)p)[1], 0u, 0u};
        else a.raw = *(const u32x4*)p;
    }
};

// resident workgroups per CU the register allocation must allow
constexpr int w4x_min_blocks(int cb, int waves) { return waves == 8 ? (cb == 1 ? 2 : 1) : (cb == 1 ? 4 : 3); }

// ABL: development ablations of the steady loop (tuning key "dbg" >> 4, built with -DLKM_W4X_ABLS for the bf16 int4 gated
// GEMM1 at 64-row tiles only; results are wrong): 1 no barrier, 2 no token staging, 4 no weight loads, 8 no weight decode,
// 16 no MFMA
template <int WF, int ADT, int CB, int WAVES, bool GATED, bool IS_G1, int PD, int DECV, int ABL = 0>
__global__ __launch_bounds__(WAVES * 64, w4x_min_blocks(CB, WAVES)) void gemm_w4x_kernel(GemmParams p) {
    typedef Dec<WF, ADT> D;
    static_assert(D::LOADS == 1 && D::UNITK == 128 && !D::A8 && !D::XS && !D::UNIT_SCALE, "4-bit formats decoded per row");
    static_assert(PD >= 2 && PD % 2 == 0, "PD even: the LDS buffer parity is the unroll index parity");
    constexpr int TM = 32 * CB, THREADS = WAVES * 64, ROWB = 256;
    constexpr int STAGEB = TM * ROWB;
    constexpr int PIECES = TM * 16 / THREADS;
    static_assert(PIECES >= 1 && (TM * 16) % THREADS == 0, "staging split");
    extern __shared__ __attribute__((aligned(16))) char xlds[];   // [2][TM][256]

    const int ti = blockIdx.y, bx = blockIdx.x;
    if (ti >= p.meta[3]) return;
    const int e = p.tile_e[ti], r0 = p.tile_r0[ti];
    const int m_e = p.counts[e], off_e = p.offsets[e];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5, i16 = lane & 15, sel = (lane >> 4) & 1;
    const int grp = bx * WAVES + wave;                       // this wave's 32-row group
    const bool pairs = IS_G1 && GATED;                       // gate tile + its up tile; else two consecutive tiles
    const int tile_lo = pairs ? grp : 2 * grp;
    const bool wave_on = tile_lo < p.T_half;                 // tail group of a padded tile count: streams tile 0, stores nothing
    const int T_all = p.T_half * p.halves;
    const int my_tile = wave_on ? (pairs ? (sel ? p.T_half + grp : grp) : 2 * grp + sel) : 0;
    const int dparam = WF == LKM_W_NVFP4 ? __builtin_bit_cast(int, p.gs ? p.gs[e] : 1.0f) : p.spu;

    const int sk = IS_G1 ? 0 : blockIdx.z;
    const int u0 = IS_G1 ? 0 : (int)((long long)sk * p.U / p.SK);
    const int U = IS_G1 ? p.U : (int)((long long)(sk + 1) * p.U / p.SK) - u0;
    const int k_base = u0 * 128;

    // ---- operand addressing: uniform base + 32-bit lane offset
    const char* wbase = (const char*)p.w + (size_t)e * p.w_estride * 16;
    const unsigned wstep = (unsigned)(p.w_ustride * 16);
    unsigned woff[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
        woff[q] = (unsigned)((my_tile * p.w_tstride + (long long)u0 * p.w_ustride + (2 * h + q) * 16 + i16) * 16);
    const char* sbase = D::aux_ptr(p.s, (size_t)e * T_all * p.U, 0, p.spu);
    const unsigned aoff = (unsigned)(size_t)D::aux_ptr((const void*)0, (size_t)my_tile * p.U + u0, lane, p.spu);
    const unsigned astep = (unsigned)D::aux_step(p.spu);

    // token staging: piece pc = q*THREADS + tid is 16 bytes of row pc/16 at LDS byte pc*16; it holds the row's logical
    // 16-byte slot (pc%16) ^ (row%16), so that the 16 lanes of a ds_read_b128 group (16 different rows, one logical slot)
    // hit 16 different bank groups
    const char* xbase = (const char*)p.x + (size_t)k_base * 2;
    unsigned xoff[PIECES];
#pragma unroll
    for (int q = 0; q < PIECES; ++q) {
        const int pc = q * THREADS + tid;
        const int row = pc >> 4, pslot = pc & 15;
        const int lslot = pslot ^ (row & 15);
        const int r = r0 + row;
        const int rr = r < m_e ? r : r0;
        const int src = IS_G1 ? p.sorted_slot[off_e + rr] / p.top_k : off_e + rr;
        xoff[q] = (unsigned)src * (unsigned)p.ldx * 2u + (unsigned)lslot * 16u;
    }
    // token fragment of lane (token j, half h), k-step (s, q): logical slot s*4 + 2h + q of row cb*32 + j
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

    auto run_h = [&](auto CBC, auto HC) __attribute__((always_inline)) {
        constexpr int CBR = decltype(CBC)::v;                 // 32-token column blocks that hold rows
        constexpr bool HOIST = decltype(HC)::v != 0;          // int4, one scale group per unit: multipliers once per unit
        constexpr int PCS = CBR * 32 * 16 / THREADS;
        static_assert(PCS >= 1 && (CBR * 32 * 16) % THREADS == 0, "block granularity");
        struct WStage {
            u32x4 w[2][1];
            typename D::Aux aux;
        };
        WStage ws[PD];
        u32x4 xs[PIECES];

        auto load_x = [&](int u) __attribute__((always_inline)) {
            const char* xb = xbase + (size_t)u * ROWB;
#pragma unroll
            for (int q = 0; q < PCS; ++q) xs[q] = *(const u32x4*)(xb + xoff[q]);
        };
        auto load_w = [&](WStage& s, int u) __attribute__((always_inline)) {
            const char* wb = wbase + (size_t)u * wstep;
#pragma unroll
            for (int q = 0; q < 2; ++q) s.w[q][0] = __builtin_nontemporal_load((const u32x4*)(wb + woff[q]));
            D::load_aux_at(s.aux, sbase + (size_t)u * astep + aoff);
        };
        auto store_x = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < PCS; ++q) *(u32x4*)(xlds + buf * STAGEB + (q * THREADS + tid) * 16) = xs[q];
        };
        auto compute = [&](const WStage& s, int buf) __attribute__((always_inline)) {
            if (!wave_on) return;
            const char* xb = xlds + buf * STAGEB;
            typename W4Int4<WF, ADT>::M mu;
            if constexpr (HOIST) mu = W4Int4<WF, ADT>::mult(s.aux);
            auto dec = [&](int s_, int q_) __attribute__((always_inline)) {
                if constexpr ((ABL & 8) != 0) return s.w[q_][0] + u32x4{(unsigned)s_, 0u, 0u, 0u};
                else if constexpr (HOIST) return W4Int4<WF, ADT>::template frag<DECV>(s.w[q_], s_, mu);
                else return D::frag(s.w[q_], s.aux, s_, dparam);
            };
            u32x4 bf[2][2][CBR];                              // [parity of s][q][column block]
            auto ldb = [&](int s_) __attribute__((always_inline)) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int c = 0; c < CBR; ++c) bf[s_ & 1][q][c] = *(const u32x4*)(xb + c * 32 * ROWB + baddr[s_][q]);
            };
            ldb(0);
            u32x4 a = dec(0, 0);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int s_ = t >> 1, q_ = t & 1;
                if (q_ == 0 && s_ + 1 < 4) ldb(s_ + 1);
                u32x4 an = a;
                if (t + 1 < 8) an = dec((t + 1) >> 1, (t + 1) & 1);
                if constexpr (!(ABL & 16)) {
#pragma unroll
                    for (int c = 0; c < CBR; ++c) acc[c] = Mfma32<ADT>::run(a, bf[s_ & 1][q_][c], acc[c]);
                } else {
#pragma unroll
                    for (int c = 0; c < CBR; ++c) acc[c][0] += __builtin_bit_cast(float, a.x ^ bf[s_ & 1][q_][c].x);
                }
                a = an;
            }
        };

        typedef std::true_type Steady;
        load_x(0);
#pragma unroll
        for (int s = 0; s < PD - 1; ++s)
            if (s < U) load_w(ws[s], s);
        store_x(0);
        if (1 < U) load_x(1);
        __syncthreads();
        // steady loop: every load unconditional (see gemm_tiled.h on vmcnt); the last PD units run in the drain loop
        const int Um = U - PD > 0 ? (U - PD) / PD * PD : 0;
        int u = 0;
        for (; u < Um; u += PD) {
            static_for<PD>([&](auto H) __attribute__((always_inline)) {
                constexpr int hh = decltype(H)::v;
                const int uu = u + hh;
                if constexpr (!(ABL & 4)) load_w(ws[(hh + PD - 1) % PD], uu + PD - 1);
                __builtin_amdgcn_sched_barrier(0);
                compute(ws[hh], hh & 1);
                if constexpr (!(ABL & 2)) {
                    store_x((hh + 1) & 1);
                    load_x(uu + 2);
                }
                if constexpr (!(ABL & 1)) __syncthreads();
            });
        }
        for (; u < U; u += PD) {
            static_for<PD>([&](auto H) __attribute__((always_inline)) {
                constexpr int hh = decltype(H)::v;
                const int uu = u + hh;
                if (uu < U) {
                    if (uu + PD - 1 < U) load_w(ws[(hh + PD - 1) % PD], uu + PD - 1);
                    compute(ws[hh], hh & 1);
                    if (uu + 1 < U) store_x((hh + 1) & 1);
                    if (uu + 2 < U) load_x(uu + 2);
                    __syncthreads();
                }
            });
        }
    };
    auto run = [&](auto CBC) __attribute__((always_inline)) {
        if constexpr (WF == LKM_W_INT4_B8 || WF == LKM_W_INT4_ZP) {
            if (p.spu <= 1) return run_h(CBC, IC<1>{});
        }
        run_h(CBC, IC<0>{});
    };
    {
        const int rows_here = m_e - r0 < TM ? m_e - r0 : TM;
        if constexpr (CB == 2) {
            if (rows_here <= 32) run(IC<1>{});
            else run(IC<2>{});
        } else {
            run(IC<1>{});
        }
    }

    // epilogue.  D layout of the 32x32 MFMA (tools/probe_mfma_layout.hip): register i of lane l = weight row
    // 8*(i/4) + 4*(l/32) + i%4, token column l%32: registers 0-3 / 4-7 are two runs of four consecutive rows of the low
    // tile, 8-11 / 12-15 the same rows of the high tile -- the gate and the up value of one output feature meet in one lane.
    if (!wave_on) return;
    static_for<CB>([&](auto CC) __attribute__((always_inline)) {
        constexpr int c = decltype(CC)::v;
        const int r_tok = r0 + c * 32 + j;
        if (r_tok < m_e) {
            static_for<2>([&](auto RC) __attribute__((always_inline)) {
                constexpr int rr = decltype(RC)::v;        // run of four rows: 0 -> rows 4h.., 1 -> rows 8 + 4h..
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
}

// usable when K is a whole number of 128-k units, tiles come in pairs, and the operands fit 32-bit byte offsets
inline bool w4x_ok(const GemmParams& p) {
    return p.Kreal % 128 == 0 && p.T_half % 2 == 0 && (size_t)p.x_rows * (size_t)p.ldx * 2 < ((size_t)1 << 32) &&
           (size_t)p.T_half * p.halves * p.U * 1024 < ((size_t)1 << 32) && !p.xcd_map;
}

template <int WF, int ADT, int CB, int WAVES, bool GATED, bool IS_G1, int PD, int DECV>
static int launch_w4x_t(hipStream_t st, const GemmParams& p, int max_tiles) {
    constexpr size_t lds = (size_t)2 * CB * 32 * 256;
    const int groups = (IS_G1 && GATED) ? p.T_half : p.T_half / 2;
    dim3 grid(ceil_div(groups, WAVES), max_tiles, IS_G1 ? 1 : p.SK), block(WAVES * 64);
    LKM_LAUNCH_GEMM((gemm_w4x_kernel<WF, ADT, CB, WAVES, GATED, IS_G1, PD, DECV>), grid, block, lds, st, p);
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}

// cfg.pf == 5 selects the kernel; cfg.tiled 32 / 64 -> one / two token column blocks; cfg.waves 4 / 8; cfg.pd 2 / 4;
// p.dbg & 1 (int4): the packed-fp32-free decoder
template <int WF, int ADT>
static bool launch_w4x_if(hipStream_t st, const LaunchCfg& cfg, const GemmParams& p, bool gated, bool is_g1, int max_tiles, int* rc) {
    if (cfg.pf != 5 || (cfg.tiled != 32 && cfg.tiled != 64) || !w4x_ok(p)) return false;
    const int waves = 4, pd = 2, cb = cfg.tiled / 32;
    const int decv = (WF == LKM_W_INT4_B8 && (p.dbg & 1)) ? 1 : 0;
#ifdef LKM_W4X_ABLS
    if constexpr (WF == LKM_W_INT4_B8 && ADT == LKM_DT_BF16) {
        const int abl = p.dbg >> 4;
        if (abl && is_g1 && gated && cb == 2 && pd == 2) {
            const int groups = p.T_half;
            dim3 grid(ceil_div(groups, waves), max_tiles, 1), block(waves * 64);
#define LKM_W4X_A(A_)                                                                                                   \
    if (abl == A_) {                                                                                                    \
        if (waves == 4) LKM_LAUNCH_GEMM((gemm_w4x_kernel<WF, ADT, 2, 4, true, true, 2, 1, A_>), grid, block, 2 * 64 * 256, st, p); \
        else LKM_LAUNCH_GEMM((gemm_w4x_kernel<WF, ADT, 2, 8, true, true, 2, 1, A_>), grid, block, 2 * 64 * 256, st, p);            \
        *rc = LKM_OK;                                                                                                   \
        return true;                                                                                                    \
    }
            LKM_W4X_A(1) LKM_W4X_A(2) LKM_W4X_A(4) LKM_W4X_A(8) LKM_W4X_A(16) LKM_W4X_A(24) LKM_W4X_A(6) LKM_W4X_A(30) LKM_W4X_A(22) LKM_W4X_A(14)
#undef LKM_W4X_A
        }
    }
#endif
#define LKM_W4X_1(CB_, W_, G_, IS1_, PD_, DV_)                                                      \
    if (cb == CB_ && waves == W_ && pd == PD_ && decv == DV_) {                                      \
        *rc = launch_w4x_t<WF, ADT, CB_, W_, G_, IS1_, PD_, DV_>(st, p, max_tiles);                  \
        return true;                                                                                 \
    }
#define LKM_W4X_DV(CB_, W_, G_, IS1_, PD_)                                                          \
    LKM_W4X_1(CB_, W_, G_, IS1_, PD_, 0)                                                             \
    if constexpr (WF == LKM_W_INT4_B8) { LKM_W4X_1(CB_, W_, G_, IS1_, PD_, 1) }
#define LKM_W4X_ALL(G_, IS1_) LKM_W4X_DV(1, 4, G_, IS1_, 2) LKM_W4X_DV(2, 4, G_, IS1_, 2)   /* (8 waves / ring depth 4 measured and dropped: no faster) */
    if (is_g1 && gated) { LKM_W4X_ALL(true, true) }
    else if (is_g1) { LKM_W4X_ALL(false, true) }
    else { LKM_W4X_ALL(false, false) }
#undef LKM_W4X_ALL
#undef LKM_W4X_DV
#undef LKM_W4X_1
    return false;
}


}  // namespace lkm
