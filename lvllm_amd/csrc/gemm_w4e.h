// gemm_w4e.h -- gemm_w4x.h's math (32x32x16 MFMA, a lane decodes one weight row of a 4-bit format) with a LOADER wave:
// tuning key "pf" = 6.
//
// Measured on gemm_w4x_kernel (profiles/r04_w4x_ablations.log; Mixtral int4 M=128 GEMM1, 140-156 us by box): the data
// movement alone -- weights + scales + token staging + fragment reads, no decode, no MFMA -- takes 111 us (one K unit of
// prefetch x 12 waves = 24 KiB in flight per CU against ~1.4 us of loaded latency), the arithmetic alone 96 us, and the two
// overlap by a quarter.  A wave that also computes has no registers for a deeper ring.  Here wave 0 of a workgroup ONLY moves
// data: weights, scales and token rows of K unit u+S-1 go to an S-deep LDS ring by LDS-DMA (no register per byte in flight;
// 1-2 slots = 24-48 KiB in flight per workgroup, two workgroups per CU), and the NC consumer waves never touch global
// memory inside the K loop: they read their two 1-KiB weight pieces, their scale and the token fragments from the slot,
// decode and multiply.  One s_barrier per unit means "slot u has landed and everybody is done with slot u-1":
//   loader, unit u:   s_waitcnt vmcnt(slots that may stay in flight x IPU) -> barrier -> DMA of unit u+S-1 (stage of u-1)
//   consumer, unit u: barrier -> ds_read (weights, scale, token fragments) -> decode -> MFMA
// LDS slot: [TM token rows x 256 B; the 16-byte slots XOR-swizzled on the SOURCE side, the LDS image is lane-linear]
//           [NC x (lo tile 1 KiB, hi tile 1 KiB), lane-linear as in HBM][NC x 2 x aux bytes of the (tile, unit), dense].
// The loader's waits are counted (vector memory retires in order): every slot is exactly IPU DMA instructions, so IPU is a
// template parameter of the K loop and no other vector-memory instruction exists in the loader's loop.
#pragma once
#include <cstdio>
#include <cstdlib>
#include "gemm_w4x.h"

#ifndef LKM_W4E_VAR
#define LKM_W4E_VAR 0      // development variants (python -m lvllm_amd.build --flag=-DLKM_W4E_VAR=n --only=gemm_w4x_int4): see the #if sites
#endif

// LKM_W4E_ABL: compile-time timing ablations (development libraries only: python -m lvllm_amd.build --flag=-DLKM_W4E_ABL=n
// --only=gemm_w4x_int4 --out=...; results are wrong): 1 no token DMA, 2 no weight / scale DMA, 4 no decode, 8 no MFMA,
// 16 no token fragment reads from the LDS, 32 no barrier in the K loop, 64 no epilogue, 128 no weight / scale reads from the LDS
#ifndef LKM_W4E_ABL
#define LKM_W4E_ABL 0
#endif

namespace lkm {

template <int N>
__device__ __forceinline__ void w4e_wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// G (round 6): row-group SETS per workgroup.  A workgroup of G > 1 walks G consecutive sets of NC row groups of ITS token tile
// as ONE stream of G x U units: the loader's ring never drains between sets (the first units of set g + 1 are in flight while the
// consumers finish set g and run its epilogue), the tile's metadata and token-row offsets are fetched once, and Mixtral's GEMM1
// is 512 workgroups -- every one resident at once.  Needs U % S == 0 and groups % (NC x G) == 0 (launch_w4e_if checks both).
// SX (round 6): depth of the TOKEN ring when it differs from the weight ring's S.  A slot is [token rows | weights | scales] and
// the token rows (16 KiB at 64-row tiles) are more than half of it, although they come from the L2 and need no more than one
// step of lead; with one ring depth the WEIGHTS -- the HBM stream -- get one step of lead too (S = 2: two workgroups per CU)
// or the workgroup is alone on its CU (S = 3: 97 KiB).  S = 3 with SX = 2 keeps two steps of weight lead AND two workgroups
// per CU (2 x 79.25 KiB): at step u the loader issues the token rows of unit u + 1, THEN the weights of unit u + 2, and waits
// with exactly the latter's instructions outstanding (vector memory retires in order).
// PW (round 6): the consumers fetch the weights and scales of unit u + 1 into REGISTERS while they compute unit u.  Measured on the
// seven-consumer kernel (profiles/r06_w4e_nc7_ablations*.log, GEMM1 131 us): decode + MFMA on registers alone 67 us, with the
// barriers 76, with the three weight / scale LDS reads of a unit 94, with the sixteen token-fragment reads 105 -- the weight reads
// sit at the head of a unit's dependency chain (barrier -> read -> wait -> decode), the token reads are issued a k-step ahead.
// With PW the loader runs the weight stream one unit further ahead than the token stream (both rings keep their depth: a weight
// slot is free again once its unit is in the consumers' registers, i.e. at the barrier that opens the unit's own step), everything
// it has issued has landed at every barrier, and one extra barrier before the loop hands over unit 0's weights.
// R (round 6): row groups per CONSUMER, walked concurrently: a consumer decodes the fragments of R groups against ONE set of token
// fragments (a workgroup covers NC x R groups: consumer c owns groups c, c + NC, ...).  The ablations say what shares the CU with
// the arithmetic is the LDS path -- the token fragments (16 KiB per wave and unit at 64-row tiles, read for every 2 KiB of
// weights) and the token rows' DMA behind them; R = 2 halves both per weight byte and the number of workgroups, for 2 x the
// accumulators (one workgroup per CU).
template <int WF, int ADT, int CB, int NC, bool GATED, bool IS_G1, int S, int DECV, int G = 1, int SX = S, int PW = 0, int R = 1>
__global__ __launch_bounds__((NC + 1) * 64) void gemm_w4e_kernel(GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef Dec<WF, ADT> D;
    typedef __attribute__((address_space(3))) void* LdsPtr;
    static_assert(D::LOADS == 1 && D::UNITK == 128 && !D::A8 && !D::XS && !D::UNIT_SCALE, "4-bit formats decoded per row");
    static_assert(S >= 2, "ring: the slot being read and the slot in flight (S = 2: a slot is refilled right after its barrier; experiment)");
    constexpr int TM = 32 * CB, ROWB = 256;
    constexpr int NCW = NC * R;                                  // row groups of a workgroup (per set)
    constexpr int XBYTES = TM * ROWB, WBYTES = NCW * 2048;
    constexpr int AUXMAX = 128;                                  // scale bytes per (tile, unit): int4 32*spu, MXFP4 64, NVFP4 128
    constexpr int WSTAGE = WBYTES + NCW * 2 * AUXMAX;           // a weight-ring slot: [NCW x 2 KiB of weights][their scales]
    constexpr int WRING = SX * XBYTES;                          // LDS: [SX token slots][S weight slots]
    static_assert(SX == S || (SX == 2 && S == 3 && G == 1), "split rings: two token slots, three weight slots, one set");
    static_assert(!PW || (SX == S && G == 1), "weight prefetch: one ring depth, one set");
    static_assert(R == 1 || (R == 2 && G == 1 && !PW), "two row groups per consumer: one set, no register prefetch");
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int ti = blockIdx.y, bx = blockIdx.x;
    if (ti >= p.meta[3]) return;
    const int e = p.tile_e[ti], r0 = p.tile_r0[ti];
    const int m_e = p.counts[e], off_e = p.offsets[e];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0 = loader, 1..NC = consumers
    const bool pairs = IS_G1 && GATED;                          // gate tile + its up tile; else two consecutive tiles
    const int T_all = p.T_half * p.halves;
    const int sk = IS_G1 ? 0 : blockIdx.z;
    const int u0 = IS_G1 ? 0 : (int)((long long)sk * p.U / p.SK);
    const int U = IS_G1 ? p.U : (int)((long long)(sk + 1) * p.U / p.SK) - u0;
    const int k_base = u0 * 128;
    const int auxB = D::aux_step(p.spu), adw = auxB / 4;         // scale bytes / dwords per (tile, unit)
    const int rows_here = m_e - r0 < TM ? m_e - r0 : TM;
    const bool two_blocks = CB == 2 && rows_here > 32;           // 32-token column blocks that hold rows (workgroup-uniform)

    if (wave == 0) {
        // ================================================================== loader
#if LKM_W4E_VAR & 2
        __builtin_amdgcn_s_setprio(3);      // (round-6 experiment: the loader never waits for an issue slot behind the consumers)
#endif
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((const char*)p.w + (size_t)e * p.w_estride * 16), 0, (int)0xffffffffu, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
            (void*)D::aux_ptr(p.s, (size_t)e * T_all * p.U, 0, p.spu), 0, (int)0xffffffffu, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((const char*)p.x + (size_t)k_base * 2), 0, (int)0xffffffffu, 0x00020000);
        auto tile_of = [&](int c, int t2) __attribute__((always_inline)) {
            const int grp = bx * G * NCW + c;                     // (set 0; set g is NCW groups further: wset / aset below)
            const bool on = (pairs ? grp : 2 * grp) < p.T_half;   // a consumer past the padded tile count streams tile 0
            return on ? (pairs ? (t2 ? p.T_half + grp : grp) : 2 * grp + t2) : 0;
        };
        // weights: (consumer c, tile t2) -> one lane-linear 1-KiB DMA
        int woff[NCW][2];
#pragma unroll
        for (int c = 0; c < NCW; ++c)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
                woff[c][t2] = __builtin_amdgcn_readfirstlane((int)((tile_of(c, t2) * p.w_tstride + (long long)u0 * p.w_ustride) * 16));
        const int wstep = (int)(p.w_ustride * 16);
        const int wset = (int)((long long)NCW * (pairs ? 1 : 2) * p.w_tstride * 16);      // byte offset of the next set's tiles
        const int aset = NCW * (pairs ? 1 : 2) * p.U * auxB;
        // scales: dword dw = (c*2 + t2) * adw + w of the slot's dense aux area; instruction k moves dwords k*64 .. k*64+63,
        // the lanes past the last dword of the last instruction are masked off
        constexpr int AIMAX = (NCW * 2 * AUXMAX / 4 + 63) / 64;
        const int n_adw = NCW * 2 * adw;
        int av[AIMAX];
#pragma unroll
        for (int k = 0; k < AIMAX; ++k) {
            const int dw = (k * 64 + lane) % n_adw;
            const int ct = dw / adw, w = dw % adw;
            av[k] = (int)(((long long)tile_of(ct >> 1, ct & 1) * p.U + u0) * auxB) + w * 4;
        }
        // tokens: instruction d moves rows d*4 .. d*4+3 (lane L: row d*4 + L/16, physical slot L%16 = logical ^ (row%16))
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
        auto run_loader = [&](auto CBC, auto AIC) __attribute__((always_inline)) {
            constexpr int CBR = decltype(CBC)::v, AIR = decltype(AIC)::v;
            constexpr int XIR = CBR * 8;
            constexpr int IPU = XIR + 2 * NCW + AIR;             // DMA instructions per slot: the counted waits rely on it
            static_assert((S - 2) * IPU < 64, "vmcnt range");      // (launch_w4e_if refuses the variants that would not fit)
            auto dma = [&](int ug, bool do_x, bool do_w) __attribute__((always_inline)) {
                char* xb = lds + (ug % SX) * XBYTES;
                char* wb = lds + WRING + (ug % S) * WSTAGE;
                int set = 0, u = ug;
                if constexpr (G > 1) {
                    set = ug / U;
                    u = ug - set * U;
                }
                const int wso = set * wset, aso = set * aset;
                // (round-6 experiment, dropped: the HBM stream first and the L2-resident token rows behind it -- no difference,
                //  profiles/r06_w4e_variants.log; the split rings below NEED the token rows first)
                if (do_x && !(LKM_W4E_ABL & 1)) {
#pragma unroll
                    for (int d = 0; d < XIR; ++d)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (LdsPtr)(xb + d * 1024), 16, xv[d], u * ROWB, 0, 0);
                }
                if (do_w && !(LKM_W4E_ABL & 2)) {
#pragma unroll
                    for (int c = 0; c < NCW; ++c)
#pragma unroll
                        for (int t2 = 0; t2 < 2; ++t2)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (LdsPtr)(wb + (c * 2 + t2) * 1024), 16, lane * 16,
                                                                     woff[c][t2] + wso + u * wstep, 0, 2);
#pragma unroll
                    for (int k = 0; k < AIR; ++k)
                        if (k * 64 + lane < n_adw)      // (partial exec on the last instruction; every instruction has >= 1 lane)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (LdsPtr)(wb + WBYTES + k * 256), 4, av[k], aso + u * auxB, 0, 0);
                }
            };
            const int UT = G * U;                                 // units of the whole stream
            if constexpr (PW != 0) {
                // issue order W(0) | X(0) W(1) | step u: X(u+1) W(u+2); nothing stays in flight across a barrier
                dma(0, false, true);
                w4e_wait_vmcnt<0>();
                if constexpr (!(LKM_W4E_ABL & 32)) __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                dma(0, true, false);
                if (1 < UT) dma(1, false, true);
                for (int u = 0; u < UT; ++u) {
                    w4e_wait_vmcnt<0>();
                    if constexpr (!(LKM_W4E_ABL & 32)) __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    if (u + 1 < UT) dma(u + 1, true, false);
                    if (u + 2 < UT) dma(u + 2, false, true);
                }
            } else if constexpr (SX == S) {
#pragma unroll
                for (int s = 0; s < S - 1; ++s)
                    if (s < UT) dma(s, true, true);
                for (int u = 0; u < UT; ++u) {
                    const int younger = UT - 1 - u;               // slots issued after slot u
                    if (younger >= S - 2) w4e_wait_vmcnt<(S - 2) * IPU>();
                    else if (S > 3 && younger == 1) w4e_wait_vmcnt<IPU>();
                    else w4e_wait_vmcnt<0>();
                    if constexpr (!(LKM_W4E_ABL & 32)) __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    if (u + S - 1 < UT) dma(u + S - 1, true, true);
                }
            } else {
                // split rings: issue order X(0) W(0) W(1) | step u: X(u+1) W(u+2).  Before barrier u everything up to X(u) has
                // to be there (W(u) was issued a step earlier): only W(u+1), issued right behind X(u), may stay in flight
                constexpr int IPW = 2 * NCW + AIR;
                dma(0, true, true);
                if (1 < UT) dma(1, false, true);
                for (int u = 0; u < UT; ++u) {
                    if (u + 1 < UT) w4e_wait_vmcnt<IPW>();
                    else w4e_wait_vmcnt<0>();
                    if constexpr (!(LKM_W4E_ABL & 32)) __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    if (u + 1 < UT) dma(u + 1, true, false);
                    if (u + 2 < UT) dma(u + 2, false, true);
                }
            }
            w4e_wait_vmcnt<0>();
        };
        static_assert(S <= 4, "the tail waits above cover S = 3, 4");
        const int air = (n_adw + 63) / 64;          // instructions that carry at least one dword (the waits count exactly these)
        auto run_l = [&](auto CBC) __attribute__((always_inline)) {
            bool done = false;
            static_for<AIMAX>([&](auto KC) __attribute__((always_inline)) {
                constexpr int k = decltype(KC)::v + 1;
                // (scale dwords per (tile, unit) are 8, 16 or 32 for every format: only those instruction counts exist)
                constexpr bool possible = k == (NCW * 16 + 63) / 64 || k == (NCW * 32 + 63) / 64 || k == (NCW * 64 + 63) / 64;
                if constexpr (possible && (S - 2) * (CB * 8 + 2 * NCW + k) < 64) {
                    if (!done && air == k) {
                        done = true;
                        run_loader(CBC, IC<k>{});
                    }
                }
            });
        };
        if constexpr (CB == 2) {
            if (two_blocks) run_l(IC<2>{});
            else run_l(IC<1>{});
        } else {
            run_l(IC<1>{});
        }
        return;
    }

    // ====================================================================== consumers
    const int cw = wave - 1;                                     // consumer index = 32-row group inside the workgroup
    const int j = lane & 31, h = lane >> 5, i16 = lane & 15, sel = (lane >> 4) & 1;
    int grp = bx * G * NCW + cw;                                 // (set 0; the set loop below advances it by NC; group r of R: + r * NC)
    bool wave_on = (pairs ? grp : 2 * grp) < p.T_half;         // (R = 2: the second group may be off on its own -- it streams tile 0 and stores nothing)
    const int dparam = WF == LKM_W_NVFP4 ? __builtin_bit_cast(int, p.gs ? p.gs[e] : 1.0f) : p.spu;
    // weight pieces of lane (tile sel, row i16, half h): old lanes (2h + q, i16) of the tile's 1-KiB block
    const int wlds = (cw * 2 + sel) * 1024 + ((2 * h) * 16 + i16) * 16;             // (relative to the weight-ring slot)
    // this lane's scales inside the (tile, unit) block: what Dec<>::aux_ptr adds for lane and spu
    const int alds = WBYTES + (cw * 2 + sel) * auxB + (int)(size_t)D::aux_ptr((const void*)0, 0, lane, p.spu);
    int baddr[4][2];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int q = 0; q < 2; ++q) baddr[s][q] = j * ROWB + (((s * 4 + 2 * h + q) ^ (j & 15)) * 16);

    f32x16 acc[R][CB];

    auto run_c = [&](auto CBC, auto HC) __attribute__((always_inline)) {
        constexpr int CBR = decltype(CBC)::v;
        constexpr bool HOIST = decltype(HC)::v != 0;            // int4, one scale group per unit: multipliers once per unit
        // (round 5: the unit loop is unrolled by the ring depth, so a unit's stage is a compile-time constant and the LDS
        //  reads of the body take it as an immediate offset: 167 -> 159 vector instructions per unit in the general loop
        //  (tools/isa_loop_stats.py), GEMM1 at Mixtral int4 M=128 146 -> 138 us, profiles/r05_int4_unrolled_ab.log)
        struct WU {                                              // a unit's weights and scales in registers
            u32x4 w[R][2][1];
            typename D::Aux aux[R];
        };
        auto fetch = [&](WU& f, const char* sb) __attribute__((always_inline)) {                // from a weight-ring slot
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if constexpr ((LKM_W4E_ABL & 128) != 0) {
                    f.w[r][0][0] = u32x4{(unsigned)(size_t)sb, 0x57575757u, 0x12345678u, (unsigned)(lane + r)};
                    f.w[r][1][0] = u32x4{0x9abcdef0u, (unsigned)(size_t)sb, 0x75757575u, (unsigned)(lane + r)};
                    __builtin_memcpy(&f.aux[r], &f.w[r][0][0], sizeof(f.aux[r]) < 16 ? sizeof(f.aux[r]) : 16);
                } else {
                    f.w[r][0][0] = *(const u32x4*)(sb + wlds + r * NC * 2048);
                    f.w[r][1][0] = *(const u32x4*)(sb + wlds + r * NC * 2048 + 256);
                    W4Int4<WF, ADT>::load_aux_lds(f.aux[r], sb + alds + r * NC * 2 * auxB, p.spu, HOIST);
                }
            }
        };
        auto unit = [&](const char* xb, const WU& f) __attribute__((always_inline)) {           // token slot, the unit's weights
            typename W4Int4<WF, ADT>::M mu[R];
            if constexpr (HOIST) {
#pragma unroll
                for (int r = 0; r < R; ++r) mu[r] = W4Int4<WF, ADT>::mult(f.aux[r]);
            }
            auto dec = [&](int r_, int s_, int q_) __attribute__((always_inline)) {
                if constexpr ((LKM_W4E_ABL & 4) != 0) return f.w[r_][q_][0] + u32x4{(unsigned)s_, 0u, 0u, 0u};
                else if constexpr (HOIST) return W4Int4<WF, ADT>::template frag<DECV>(f.w[r_][q_], s_, mu[r_]);
                else return D::frag(f.w[r_][q_], f.aux[r_], s_, dparam);
            };
            // (measured, profiles/r04_w4e_schedule_ab.log: issuing all 16 fragment reads of the unit up front and fencing
            // every k-step so that the MFMAs stay one decode apart -- 146 registers -- runs GEMM1 166 us against 144 for
            // the compiler's own just-in-time reads below)
            u32x4 bf[2][2][CBR];                                 // [parity of s][q][column block]
            auto ldb = [&](int s_) __attribute__((always_inline)) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int c = 0; c < CBR; ++c) {
                        if constexpr ((LKM_W4E_ABL & 16) != 0) bf[s_ & 1][q][c] = u32x4{(unsigned)baddr[s_][q], 1u, 2u, (unsigned)c};
                        else bf[s_ & 1][q][c] = *(const u32x4*)(xb + c * 32 * ROWB + baddr[s_][q]);
                    }
            };
            ldb(0);
            u32x4 a[R];
#pragma unroll
            for (int r = 0; r < R; ++r) a[r] = dec(r, 0, 0);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int s_ = t >> 1, q_ = t & 1;
                if (q_ == 0 && s_ + 1 < 4) ldb(s_ + 1);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    u32x4 an = a[r];
                    if (t + 1 < 8) an = dec(r, (t + 1) >> 1, (t + 1) & 1);
#pragma unroll
                    for (int c = 0; c < CBR; ++c) {
                        if constexpr ((LKM_W4E_ABL & 8) != 0) acc[r][c][0] += __builtin_bit_cast(float, a[r].x ^ a[r].y ^ a[r].z ^ a[r].w ^ bf[s_ & 1][q_][c].x);
                        else acc[r][c] = Mfma32<ADT>::run(a[r], bf[s_ & 1][q_][c], acc[r][c]);
                    }
                    a[r] = an;
                }
            }
        };
        constexpr int UNR = SX == S ? S : S * SX;                // (both slots of a unit are compile-time constants)
        if constexpr (PW != 0) {
            WU cur, nxt;
            if constexpr (!(LKM_W4E_ABL & 32)) __builtin_amdgcn_s_barrier();          // unit 0's weights have landed
            if (wave_on) fetch(nxt, lds + WRING);
            for (int u0 = 0; u0 < U; u0 += UNR) {
                static_for<UNR>([&](auto SC) __attribute__((always_inline)) {
                    constexpr int st = decltype(SC)::v;
                    if (u0 + st < U) {
                        if constexpr (!(LKM_W4E_ABL & 32)) __builtin_amdgcn_s_barrier();
                        if (wave_on) {
                            cur = nxt;
                            if (u0 + st + 1 < U) fetch(nxt, lds + WRING + ((st + 1) % S) * WSTAGE);
                            unit(lds + (st % SX) * XBYTES, cur);
                        }
                    }
                });
            }
        } else
        for (int u0 = 0; u0 < U; u0 += UNR) {
            static_for<UNR>([&](auto SC) __attribute__((always_inline)) {
                constexpr int st = decltype(SC)::v;
                if (u0 + st < U) {                               // (wave-uniform; every wave of the workgroup counts the same barriers)
                    if constexpr (!(LKM_W4E_ABL & 32)) __builtin_amdgcn_s_barrier();
                    if (wave_on) {
                        WU f;
                        fetch(f, lds + WRING + (st % S) * WSTAGE);
                        unit(lds + (st % SX) * XBYTES, f);
                    }
                }
            });
        }
    };
    auto run = [&](auto CBC) __attribute__((always_inline)) {
        if constexpr (WF == LKM_W_INT4_B8 || WF == LKM_W_INT4_ZP) {
            if (p.spu <= 1) return run_c(CBC, IC<1>{});
        }
        run_c(CBC, IC<0>{});
    };
    for (int set = 0; set < G; ++set) {
    if (set) {
        grp += NC;
        wave_on = (pairs ? grp : 2 * grp) < p.T_half;
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < CB; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][c][i] = 0.f;
    if constexpr (CB == 2) {
        if (two_blocks) run(IC<2>{});
        else run(IC<1>{});
    } else {
        run(IC<1>{});
    }

    if constexpr ((LKM_W4E_ABL & 64) != 0) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int c = 0; c < CB; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) t += acc[r][c][i];
        if (t == 1.2345f) ((float*)p.out)[0] = t;
        continue;
    }
    // epilogue: as gemm_w4x_kernel (D layout of the 32x32 MFMA)
    static_for<R>([&](auto GC) __attribute__((always_inline)) {
    constexpr int gr = decltype(GC)::v;
    const int grp_s = grp;                                      // (shadowed below: this group's index)
    const int grp = grp_s + gr * NC;
    if (wave_on && (pairs ? grp : 2 * grp) < p.T_half)
    static_for<CB>([&](auto CC) __attribute__((always_inline)) {
        constexpr int c = decltype(CC)::v;
        const int r_tok = r0 + c * 32 + j;
        if (r_tok < m_e) {
            static_for<2>([&](auto RC) __attribute__((always_inline)) {
                constexpr int rr = decltype(RC)::v;
                const f32x4 lo = {acc[gr][c][rr * 4 + 0], acc[gr][c][rr * 4 + 1], acc[gr][c][rr * 4 + 2], acc[gr][c][rr * 4 + 3]};
                const f32x4 hi = {acc[gr][c][8 + rr * 4 + 0], acc[gr][c][8 + rr * 4 + 1], acc[gr][c][8 + rr * 4 + 2], acc[gr][c][8 + rr * 4 + 3]};
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
    });   // group of R
    }   // set
#else
    (void)p;
#endif
}

template <int WF, int ADT, int CB, int NC, bool GATED, bool IS_G1, int S, int DECV, int G = 1, int SX = S, int PW = 0, int R = 1>
static int launch_w4e_t(hipStream_t st, const GemmParams& p, int max_tiles) {
    constexpr size_t lds = (size_t)SX * (CB * 32 * 256) + (size_t)S * (NC * R * 2048 + NC * R * 2 * 128);
    const int groups = (IS_G1 && GATED) ? p.T_half : p.T_half / 2;
    dim3 grid(ceil_div(groups, NC * R * G), max_tiles, IS_G1 ? 1 : p.SK), block((NC + 1) * 64);
    auto kern = gemm_w4e_kernel<WF, ADT, CB, NC, GATED, IS_G1, S, DECV, G, SX, PW, R>;
    if (lds > 64 * 1024) LKM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    static const bool dbg_occ = getenv("LKM_DEBUG_OCC") != nullptr;      // (development: what the occupancy API says about this variant)
    if (dbg_occ) {
        int nb = -1;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, (NC + 1) * 64, lds);
        fprintf(stderr, "[w4e] CB=%d NC=%d R=%d S=%d SX=%d PW=%d lds=%zu: occupancy API -> %d workgroups per CU (%s)\n", CB, NC, R, S, SX, PW, lds, nb, hipGetErrorString(e));
    }
    LKM_LAUNCH_GEMM(kern, grid, block, lds, st, p);
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}

// cfg.pf == 6 selects the kernel; cfg.tiled 32 / 64 -> one / two token column blocks; cfg.pd = ring depth S (3 / 4);
// p.dbg & 1 (int4): the packed-fp32-free decoder
template <int WF, int ADT>
static bool launch_w4e_if(hipStream_t st, const LaunchCfg& cfg, const GemmParams& p, bool gated, bool is_g1, int max_tiles, int* rc) {
    if (cfg.pf != 6 || (cfg.tiled != 32 && cfg.tiled != 64) || !w4x_ok(p)) return false;
    // built: four consumer waves, ring depth 3 (7 / 8 / 14 consumers and depth 4 measured and dropped:
    // profiles/r04_w4e_schedule_ab.log, r04_w4x_batch_sweep.log)
    const bool split = cfg.pd == 32 && is_g1 && cfg.waves == 7 && cfg.tiled == 64 && !(p.dbg & 1) && cfg.kw != 2;                              // "pd" = 32: three weight slots, two token slots (gemm_w4e_kernel: SX)
    const int cb = cfg.tiled / 32, s = (cfg.pd == 2 || (cfg.pd == 34 && is_g1)) ? 2 : 3, nc_ask = cfg.waves == 7 ? 7 : 4;
    // round 6 (profiles/r06_w4e_nc7.log, r06_w4e_nc7_b.log): "pd" = 2 -> a two-slot ring (a slot is refilled right behind its
    // barrier; 51 KiB per four-consumer workgroup instead of 77); "waves" = 7 -> SEVEN consumers per workgroup on the two-slot
    // ring (65 KiB, two workgroups = 14 consumers per CU): Mixtral's 896 row groups per expert are 128 workgroups of 7, eight
    // experts = 1024 workgroups = exactly two rounds of the chip's 512 slots, where 224 workgroups of 4 leave the fourth round
    // half empty -- GEMM1 132.5 -> 124.3 us uniform, 145.4 -> 139 Zipf (same box, captured step, alternating).  Built for the
    // default decoder at 64-row tiles; eight and fourteen consumers, and seven on the three-slot ring, measured and dropped.
    // Later in round 6 (all at Mixtral int4 M = 128, GEMM1, same box, alternating):
    //   * where a workgroup's waves land (tools/probe_wave_place.hip): cyclically 0 -> 2 -> 1 -> 3 from a start that differs
    //     between the co-resident workgroups, so 2 x (loader + 7) puts [3 4 3 4] consumers on the SIMDs and 2 x (loader + 8) a
    //     balanced [4 4 4 4] -- which ran 150 us against 135 (profiles/r06_w4e_nc8.log: nine waves need a 96-register cap, 896
    //     workgroups are 1.75 rounds); dropped.
    //   * "pd" = 32: THREE weight slots beside TWO token slots (SX), i.e. two steps of lead for the HBM stream with two
    //     workgroups per CU still resident (79.25 KiB each) -- 129.9 us against 129.5 (profiles/r06_w4e_split_rings.log): the
    //     weights' lead is not what the step waits for.  Kept as an opt-in variant (tests/test_gpu_w4x.py).
    //   * "pd" = 34: the consumers fetch unit u + 1's weights into registers during unit u (PW) -- the ablations priced the three
    //     weight / scale reads at the head of a unit at 18 us (profiles/r06_w4e_nc7_ablations_b.log) -- 145 us against 139
    //     (profiles/r06_w4e_weight_prefetch.log): with four waves per SIMD that latency was already covered by the other waves,
    //     and the copies and the all-landed barrier cost more.  Opt-in variant.
    const int decv = (WF == LKM_W_INT4_B8 && (p.dbg & 1)) ? 1 : 0;
    const int nc = (nc_ask == 7 && cb == 2 && (s == 2 || split) && decv == 0) ? 7 : 4;
    // "kw" = 2 (round 6): TWO sets of seven row groups per workgroup, one uninterrupted stream of 2 U units (gemm_w4e_kernel: G)
    const int groups_all = (is_g1 && gated) ? p.T_half : p.T_half / 2;
    const int sets = (cfg.kw == 2 && nc == 7 && is_g1 && p.U % 2 == 0 && groups_all % 14 == 0) ? 2 : 1;
    // the loader's counted waits need (S - 2) x (DMA instructions per slot) < 64
    const int aux_b = WF == LKM_W_INT4_B8 ? 32 * p.spu : (WF == LKM_W_INT4_ZP ? 64 * p.spu : (WF == LKM_W_MXFP4 ? 64 : 128));      // = Dec<>::aux_step (device side)
    if (aux_b > 128) return false;                      // (zero points at 32-k groups: 256 bytes of pairs per (tile, unit) -- the tile kernel's)
    const int air = (nc * 2 * aux_b / 4 + 63) / 64;
    if ((s - 2) * (cb * 8 + 2 * nc + air) >= 64) return false;
#define LKM_W4E_1(CB_, NC_, G_, IS1_, S_, DV_)                                                      \
    if (cb == CB_ && nc == NC_ && s == S_ && decv == DV_) {                                          \
        *rc = launch_w4e_t<WF, ADT, CB_, NC_, G_, IS1_, S_, DV_>(st, p, max_tiles);                  \
        return true;                                                                                 \
    }
#define LKM_W4E_DV(CB_, G_, IS1_, S_)                                                               \
    LKM_W4E_1(CB_, 4, G_, IS1_, S_, 0)                                                               \
    if constexpr (WF == LKM_W_INT4_B8) { LKM_W4E_1(CB_, 4, G_, IS1_, S_, 1) }
#define LKM_W4E_ALL(G_, IS1_) LKM_W4E_DV(1, G_, IS1_, 3) LKM_W4E_DV(2, G_, IS1_, 3) LKM_W4E_DV(2, G_, IS1_, 2) LKM_W4E_1(2, 7, G_, IS1_, 2, 0)
    if constexpr (WF == LKM_W_INT4_B8) {               // (the round-6 experiments are built for uint4b8 only)
    if (cfg.pd == 34 && is_g1 && nc == 7 && sets == 1) {       // "pd" = 34 (experiment): the two-slot ring with the consumers' weight prefetch
        if (gated) *rc = launch_w4e_t<WF, ADT, 2, 7, true, true, 2, 0, 1, 2, 1>(st, p, max_tiles);
        else *rc = launch_w4e_t<WF, ADT, 2, 7, false, true, 2, 0, 1, 2, 1>(st, p, max_tiles);
        return true;
    }
    if ((cfg.pd == 36 || cfg.pd == 37) && is_g1 && nc_ask == 7 && cb == 2 && decv == 0 && cfg.kw != 2) {
        // "pd" = 36 / 37 (experiment): TWO row groups per consumer on a two- / three-slot ring (gemm_w4e_kernel: R)
        if (cfg.pd == 36) {
            if (gated) *rc = launch_w4e_t<WF, ADT, 2, 7, true, true, 2, 0, 1, 2, 0, 2>(st, p, max_tiles);
            else *rc = launch_w4e_t<WF, ADT, 2, 7, false, true, 2, 0, 1, 2, 0, 2>(st, p, max_tiles);
        } else {
            if (gated) *rc = launch_w4e_t<WF, ADT, 2, 7, true, true, 3, 0, 1, 3, 0, 2>(st, p, max_tiles);
            else *rc = launch_w4e_t<WF, ADT, 2, 7, false, true, 3, 0, 1, 3, 0, 2>(st, p, max_tiles);
        }
        return true;
    }
    if (split) {
        if (gated) *rc = launch_w4e_t<WF, ADT, 2, 7, true, true, 3, 0, 1, 2>(st, p, max_tiles);
        else *rc = launch_w4e_t<WF, ADT, 2, 7, false, true, 3, 0, 1, 2>(st, p, max_tiles);
        return true;
    }
    if (sets == 2) {
        if (gated) *rc = launch_w4e_t<WF, ADT, 2, 7, true, true, 2, 0, 2>(st, p, max_tiles);
        else *rc = launch_w4e_t<WF, ADT, 2, 7, false, true, 2, 0, 2>(st, p, max_tiles);
        return true;
    }
    }
    if (is_g1 && gated) { LKM_W4E_ALL(true, true) }
    else if (is_g1) { LKM_W4E_ALL(false, true) }
    else { LKM_W4E_ALL(false, false) }
#undef LKM_W4E_ALL
#undef LKM_W4E_DV
#undef LKM_W4E_1
    return false;
}

// one launcher per (format, dtype) translation unit: "pf" = 6 -> the loader / consumer kernel, 5 -> gemm_w4x_kernel
#define LKM_DEFINE_W4X_LAUNCHER(SUFFIX, WF, ADT)                                                              \
    bool launch_w4x_##SUFFIX(hipStream_t st, const LaunchCfg& cfg, const GemmParams& p, bool gated, bool is_g1, \
                             int max_tiles, int* rc) {                                                        \
        if (launch_w4e_if<WF, ADT>(st, cfg, p, gated, is_g1, max_tiles, rc)) return true;                     \
        return launch_w4x_if<WF, ADT>(st, cfg, p, gated, is_g1, max_tiles, rc);                               \
    }
// ... and with "pf" = 7 the two-queue kernel (gemm_w4s.h) first: the translation units that include gemm_w4s.h use this one
#define LKM_DEFINE_W4S_LAUNCHER(SUFFIX, WF, ADT)                                                              \
    bool launch_w4x_##SUFFIX(hipStream_t st, const LaunchCfg& cfg, const GemmParams& p, bool gated, bool is_g1, \
                             int max_tiles, int* rc) {                                                        \
        if (launch_w4s_if<WF, ADT>(st, cfg, p, gated, is_g1, max_tiles, rc)) return true;                     \
        if (launch_w4e_if<WF, ADT>(st, cfg, p, gated, is_g1, max_tiles, rc)) return true;                     \
        return launch_w4x_if<WF, ADT>(st, cfg, p, gated, is_g1, max_tiles, rc);                               \
    }

}  // namespace lkm
