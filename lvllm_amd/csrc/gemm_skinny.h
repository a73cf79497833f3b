// gemm_skinny.hip -- per-expert grouped GEMMs for the decode regime (few rows per expert).
//
// Computes what the reference's routed-expert operator computes between the router and the final
// sum:   act = ACT(x_e . W13[e]^T)   and   y = act . W2[e]^T
//   math spec : vllm/model_executor/layers/fused_moe/fused_moe.py:298-610 (fp32 accumulate),
//               :64-295 (int4: b = T((nib-8)*scale)), csrc/cpu/cpu_fused_moe.cpp:229-522,
//               activation: csrc/libtorch_stable/activation_kernels.cu:57-75,401-408,
//               vllm/model_executor/layers/fused_moe/activation.py:208-210 (relu2).
//
// MI355X design (this is NOT how the reference does it):
//   * decode is HBM-bound (8 FLOP/B for Mixtral M=32): the kernel is a weight STREAMER.  Weights
//     are the MFMA *A* operand (16 weight rows x 32 k per mfma_f32_16x16x32), tokens are the *B*
//     operand (32 k x 16 tokens), so one 1-KiB wave-wide nontemporal global_load_dwordx4 of the
//     pre-shuffled layout (lkm_common.h) IS an A fragment: no LDS round trip, no shuffles, every
//     weight byte crosses HBM->VGPR exactly once and feeds TB MFMAs (TB = 16-token blocks).
//   * one wavefront owns NT 16-row tiles (x2 for gate+up) over a K range; loads are double-buffered
//     in registers one "unit" (64 or 128 k) ahead, so each wave keeps >= NT*2*LOADS KiB in flight;
//     thousands of waves => >= 32 KiB in flight per CU, the streaming regime of MI355X_MICROARCH.
//   * quantised formats are decoded in registers right before the MFMA (int4: nibble -> f32 ->
//     fma(q, s, -8s) (exact) -> RNE to the act dtype == the reference's T((q-8)*s); fp8: hardware
//     v_cvt_pk_f32_fp8 (exact in bf16/f16), block scale applied to the fp32 partial sums).
//   * GEMM1 fuses the activation (SiLU-mul / swigluoai / relu2) into the epilogue and writes the
//     act-dtype intermediate in expert-sorted row order; GEMM2 writes fp32 split-K partials that the
//     combine kernel (dispatch.hip) reduces together with the top-k weighting.
#pragma once
#include <type_traits>
#include <utility>

#include "lkm_kernels.h"

namespace lkm {

// ------------------------------------------------------------------ in-register weight decoders
template <int WF, int ADT>
struct Dec;

template <int ADT>
struct DecPlain {
    static constexpr int UNITK = 64, LOADS = 2, KSTEPS = 2;
    static constexpr bool UNIT_SCALE = false, A8 = false, XS = false;
    struct Aux {};
    static __device__ __forceinline__ void load_aux(Aux&, const void*, size_t, int, int) {}
    // pointer form (tiled kernels): aux_ptr(unit 0 of a tile) + u * aux_step(spu)
    static __device__ __forceinline__ const char* aux_ptr(const void*, size_t, int, int) { return nullptr; }
    static __device__ __forceinline__ int aux_step(int) { return 0; }
    static __device__ __forceinline__ void load_aux_at(Aux&, const char*) {}
    static __device__ __forceinline__ u32x4 frag(const u32x4 (&raw)[LOADS], const Aux&, int ks, int) {
        return raw[ks];
    }
};
template <>
struct Dec<LKM_W_BF16, LKM_DT_BF16> : DecPlain<LKM_DT_BF16> {};
template <>
struct Dec<LKM_W_F16, LKM_DT_F16> : DecPlain<LKM_DT_F16> {};

template <int ADT>
struct Dec<LKM_W_INT4_B8, ADT> {
    static constexpr int UNITK = 128, LOADS = 1, KSTEPS = 4;
    static constexpr bool UNIT_SCALE = false, A8 = false, XS = false;
    struct Aux {
        u32x2 raw;   // up to four act-dtype scales of this lane's weight row for the 128-k unit
    };
    typedef u32x2 __attribute__((aligned(2))) u32x2_unaligned;
    // scales: [tile][unit][16 rows][spu] act dtype; tu = tile*U + unit.  Branch-free: always fetch 8
    // bytes (the buffer is padded) and pick entry (kstep*spu)/4 when the fragment is decoded, so the
    // load has no dependent ALU work and stays in flight with the weight loads.
    static __device__ __forceinline__ void load_aux(Aux& a, const void* sbase, size_t tu, int lane,
                                                    int spu) {
        const unsigned short* p = (const unsigned short*)sbase + (tu * 16 + (lane & 15)) * spu;
        a.raw = *(const u32x2_unaligned*)p;
    }
    static __device__ __forceinline__ const char* aux_ptr(const void* sbase, size_t tu, int lane, int spu) {
        return (const char*)((const unsigned short*)sbase + (tu * 16 + (lane & 15)) * spu);
    }
    static __device__ __forceinline__ int aux_step(int spu) { return 32 * spu; }
    static __device__ __forceinline__ void load_aux_at(Aux& a, const char* p) {
        a.raw = *(const u32x2_unaligned*)p;
    }
    // the two multipliers of a (row, scale group): {512 s, 512 s} and {-8 s, -8 s}
    struct Mult {
        f32x2 s512, m8;
    };
    static __device__ __forceinline__ Mult mult(const Aux& a, int ks, int spu) {
        const int idx = (ks * spu) >> 2;   // 0 for g>=128, ks/2 for g=64, ks for g=32 (wave-uniform)
        // scale `idx` of the 8 fetched bytes -> f32 with one v_perm_b32 (bf16: the half-word moved to
        // the upper half IS the float; f16: to the lower half, then v_cvt_f32_f16)
        float s;
        if constexpr (ADT == LKM_DT_BF16) {
            s = __builtin_bit_cast(float, __builtin_amdgcn_perm(a.raw.y, a.raw.x, 0x01000c0cu + idx * 0x02020000u));
        } else {
            s = ActT<ADT>::to_f32((unsigned short)__builtin_amdgcn_perm(a.raw.y, a.raw.x, 0x0c0c0100u + idx * 0x0202u));
        }
        // The multiplier and the addend are two full pairs {512 s, 512 s} and {-8 s, -8 s} behind empty
        // asm statements: folded into ONE pair {512 s, -8 s} read through two op_sel swizzles (what the
        // compiler does by itself) the decode is wrong now and then -- see splat2_opaque, lkm_common.h.
        f32x2 sv = {s, s};
        asm("" : "+v"(sv));
        Mult m;
        m.s512 = sv * f32x2{512.0f, 512.0f};
        m.m8 = sv * f32x2{-8.0f, -8.0f};
        asm("" : "+v"(m.s512));
        asm("" : "+v"(m.m8));
        return m;
    }
    // A byte 0000vvvv read as OCP e4m3 is v * 2^-9 for EVERY v in 0..15 (subnormals and the first
    // binade are equally spaced), so v_cvt_pk_f32_fp8 turns two masked nibbles into two floats in
    // one instruction; fma(v * 2^-9, 512 s, -8 s) = (v - 8) s exactly (<= 15 significant bits),
    // two at a time on v_pk_fma_f32, then one RNE to the act dtype: 15 VALU per 8 weights (+4 for
    // the scale, which the tiled kernels pay once per 128-k unit when the group is >= 128), where
    // v_cvt_f32_ubyte + scalar fma took 23 (+5) and per-nibble v_bfe_u32 31.
    // Same bits as the reference's T((q-8)*s).
    static __device__ __forceinline__ u32x4 frag_m(const u32x4 (&raw)[LOADS], int ks, const Mult& m) {
        const unsigned w = raw[0][ks];
        const unsigned lo = w & 0x0f0f0f0fu, hi = (w >> 4) & 0x0f0f0f0fu;
        // byte b of the dword holds k=2b (low nibble) and k=2b+1 (high nibble)
        const f32x2 e01 = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8(lo, false), m.s512, m.m8);  // k = 0, 2
        const f32x2 e23 = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8(lo, true), m.s512, m.m8);   // k = 4, 6
        const f32x2 o01 = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8(hi, false), m.s512, m.m8);  // k = 1, 3
        const f32x2 o23 = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8(hi, true), m.s512, m.m8);   // k = 5, 7
        u32x4 o;
        o.x = ActT<ADT>::pack2(e01.x, o01.x);
        o.y = ActT<ADT>::pack2(e01.y, o01.y);
        o.z = ActT<ADT>::pack2(e23.x, o23.x);
        o.w = ActT<ADT>::pack2(e23.y, o23.y);
        return o;
    }
    static __device__ __forceinline__ u32x4 frag(const u32x4 (&raw)[LOADS], const Aux& a, int ks,
                                                 int spu) {
        return frag_m(raw, ks, mult(a, ks, spu));
    }
};

template <int ADT>
struct Dec<LKM_W_FP8_E4M3, ADT> {
    static constexpr int UNITK = 128, LOADS = 2, KSTEPS = 4;
    static constexpr bool UNIT_SCALE = true, A8 = false, XS = false;
    struct Aux {
        f32x4 s;  // block scale of this lane's 4 output rows (g*4 + r)
    };
    // scales: [tile][unit][16 rows] fp32
    static __device__ __forceinline__ void load_aux(Aux& a, const void* sbase, size_t tu, int lane,
                                                    int) {
        a.s = *(const f32x4*)((const float*)sbase + tu * 16 + (lane >> 4) * 4);
    }
    static __device__ __forceinline__ const char* aux_ptr(const void* sbase, size_t tu, int lane, int) {
        return (const char*)((const float*)sbase + tu * 16 + (lane >> 4) * 4);
    }
    static __device__ __forceinline__ int aux_step(int) { return 64; }
    static __device__ __forceinline__ void load_aux_at(Aux& a, const char* p) { a.s = *(const f32x4*)p; }
    static __device__ __forceinline__ u32x4 frag(const u32x4 (&raw)[LOADS], const Aux&, int ks, int) {
        const unsigned d0 = raw[ks >> 1][(ks & 1) * 2], d1 = raw[ks >> 1][(ks & 1) * 2 + 1];
        f32x2 p0 = __builtin_amdgcn_cvt_pk_f32_fp8(d0, false);
        f32x2 p1 = __builtin_amdgcn_cvt_pk_f32_fp8(d0, true);
        f32x2 p2 = __builtin_amdgcn_cvt_pk_f32_fp8(d1, false);
        f32x2 p3 = __builtin_amdgcn_cvt_pk_f32_fp8(d1, true);
        u32x4 o;
        o.x = ActT<ADT>::pack2(p0.x, p0.y);
        o.y = ActT<ADT>::pack2(p1.x, p1.y);
        o.z = ActT<ADT>::pack2(p2.x, p2.y);
        o.w = ActT<ADT>::pack2(p3.x, p3.y);
        return o;
    }
};

// uint4 with zero points (LKM_W_INT4_ZP, lkm_common.h): the decoder above with the addend -zp * s.  The scale image holds
// (scale, zero point) pairs in the activation dtype; a lane fetches the (up to four) pairs of its row for the 128-k unit with
// one 16-byte load.  fma(v * 2^-9, 512 s, -zp s) = (v - zp) s exactly, one RNE to the act dtype: the reference's
// T((q - zp) * s) (fused_moe.py:272-276), bit for bit.
template <int ADT>
struct Dec<LKM_W_INT4_ZP, ADT> {
    static constexpr int UNITK = 128, LOADS = 1, KSTEPS = 4;
    static constexpr bool UNIT_SCALE = false, A8 = false, XS = false;
    typedef Dec<LKM_W_INT4_B8, ADT> B8;
    typedef typename B8::Mult Mult;
    struct Aux {
        u32x4 raw;   // up to four (scale, zero point) pairs of this lane's weight row for the 128-k unit
    };
    typedef u32x4 __attribute__((aligned(4))) u32x4_unaligned;
    // pairs: [tile][unit][16 rows][spu][2] act dtype; 4 spu bytes per row.  Branch-free as in the symmetric decoder: always
    // fetch 16 bytes (the buffer is padded) and pick pair (kstep * spu) / 4 when the fragment is decoded.
    static __device__ __forceinline__ void load_aux(Aux& a, const void* sbase, size_t tu, int lane, int spu) {
        a.raw = *(const u32x4_unaligned*)((const unsigned*)sbase + (tu * 16 + (lane & 15)) * spu);
    }
    static __device__ __forceinline__ const char* aux_ptr(const void* sbase, size_t tu, int lane, int spu) {
        return (const char*)((const unsigned*)sbase + (tu * 16 + (lane & 15)) * spu);
    }
    static __device__ __forceinline__ int aux_step(int spu) { return 64 * spu; }
    static __device__ __forceinline__ void load_aux_at(Aux& a, const char* p) { a.raw = *(const u32x4_unaligned*)p; }
    static __device__ __forceinline__ Mult mult(const Aux& a, int ks, int spu) {
        const int idx = (ks * spu) >> 2;   // 0 for g>=128, ks/2 for g=64, ks for g=32 (wave-uniform)
        const unsigned pair = idx == 0 ? a.raw.x : (idx == 1 ? a.raw.y : (idx == 2 ? a.raw.z : a.raw.w));
        const float s = ActT<ADT>::to_f32((unsigned short)(pair & 0xffffu));
        const float z = ActT<ADT>::to_f32((unsigned short)(pair >> 16));
        f32x2 sv = {s, s};
        asm("" : "+v"(sv));
        Mult m;
        m.s512 = sv * f32x2{512.0f, 512.0f};
        m.m8 = sv * f32x2{-z, -z};
        asm("" : "+v"(m.s512));
        asm("" : "+v"(m.m8));
        return m;
    }
    static __device__ __forceinline__ u32x4 frag(const u32x4 (&raw)[LOADS], const Aux& a, int ks, int spu) {
        return B8::frag_m(raw, ks, mult(a, ks, spu));
    }
};

// uint4b8, fast mode (LKM_W_INT4_PS, lkm_common.h): 7 VALU per 16 x 32 fragment (three shifts, four v_and_or_b32)
// where the bit-exact decoder above needs 16-19.  A nibble v in the low bits of the mantissa of BIAS = 2^7 (bf16)
// / 2^10 (fp16) IS the value BIAS + v; the pair order of the re-packed dword (repack.hip) makes dword p of the
// fragment (k = 2p, 2p + 1) one mask away from the raw dword shifted by 4p.  The per-(token, unit) term
// (BIAS + 8) * sum_k x_k arrives as Streamer / tile `xs` (launch_rowsum128_rows), the group scale as fp32 per
// (row, unit) in the fp8 layout; groups of 128 k and multiples only.
template <int ADT>
struct Dec<LKM_W_INT4_PS, ADT> {
    static constexpr int UNITK = 128, LOADS = 1, KSTEPS = 4;
    static constexpr bool UNIT_SCALE = true, A8 = false, XS = true;
    static constexpr float BIAS8 = ADT == LKM_DT_BF16 ? 136.0f : 1032.0f;     // BIAS + 8
    struct Aux {
        f32x4 s;  // group scale of this lane's 4 output rows (g*4 + r) for the unit
    };
    static __device__ __forceinline__ void load_aux(Aux& a, const void* sbase, size_t tu, int lane, int) {
        a.s = *(const f32x4*)((const float*)sbase + tu * 16 + (lane >> 4) * 4);
    }
    static __device__ __forceinline__ const char* aux_ptr(const void* sbase, size_t tu, int lane, int) {
        return (const char*)((const float*)sbase + tu * 16 + (lane >> 4) * 4);
    }
    static __device__ __forceinline__ int aux_step(int) { return 64; }
    static __device__ __forceinline__ void load_aux_at(Aux& a, const char* p) { a.s = *(const f32x4*)p; }
    static __device__ __forceinline__ u32x4 frag(const u32x4 (&raw)[LOADS], const Aux&, int ks, int) {
        constexpr unsigned ONE = ADT == LKM_DT_BF16 ? 0x43004300u : 0x64006400u;
        const unsigned w = raw[0][ks];
        u32x4 o;
        o.x = (w & 0x000f000fu) | ONE;
        o.y = ((w >> 4) & 0x000f000fu) | ONE;
        o.z = ((w >> 8) & 0x000f000fu) | ONE;
        o.w = ((w >> 12) & 0x000f000fu) | ONE;
        return o;
    }
};

// OCP MXFP4 (E2M1 values, one E8M0 scale per 32 k = per MFMA k-step): gfx950 converts a packed pair
// of FP4 straight to the activation dtype AND applies the power-of-two scale in one instruction
// (v_cvt_scalef32_pk_{bf16,f16}_fp4; only the exponent of the f32 scale operand is used -- measured,
// tools/probe_cvt.hip), i.e. 4 VALU per 16x32 fragment against ~15 for uint4b8.  Exact: an E2M1 value
// has 2 significant bits (reference_mxfp4.py:91-117 computes the same product in the act dtype).
template <int ADT>
struct Dec<LKM_W_MXFP4, ADT> {
    static constexpr int UNITK = 128, LOADS = 1, KSTEPS = 4;
    static constexpr bool UNIT_SCALE = false, A8 = false, XS = false;
    struct Aux {
        unsigned raw;   // the four E8M0 scales (k-steps 0..3 of the unit) of this lane's weight row
    };
    // scales: [tile][unit][16 rows][4] bytes
    static __device__ __forceinline__ void load_aux(Aux& a, const void* sbase, size_t tu, int lane, int) {
        a.raw = ((const unsigned*)sbase)[tu * 16 + (lane & 15)];
    }
    static __device__ __forceinline__ const char* aux_ptr(const void* sbase, size_t tu, int lane, int) {
        return (const char*)((const unsigned*)sbase + tu * 16 + (lane & 15));
    }
    static __device__ __forceinline__ int aux_step(int) { return 64; }
    static __device__ __forceinline__ void load_aux_at(Aux& a, const char* p) { a.raw = *(const unsigned*)p; }
    static __device__ __forceinline__ u32x4 frag(const u32x4 (&raw)[LOADS], const Aux& a, int ks, int) {
        const unsigned w = raw[0][ks];
        // E8M0 e -> the f32 whose exponent field is e (2^(e-127))
        const float sc = __builtin_bit_cast(float, ((a.raw >> (8 * ks)) & 0xffu) << 23);
        u32x4 o;
        if constexpr (ADT == LKM_DT_BF16) {
            o.x = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 0));
            o.y = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 1));
            o.z = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 2));
            o.w = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 3));
        } else {
            o.x = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, sc, 0));
            o.y = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, sc, 1));
            o.z = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, sc, 2));
            o.w = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, sc, 3));
        }
        return o;
    }
};

// NVFP4 (E2M1 values, e4m3fn scale per 16 k, per-expert f32 multiplier): w = T(fp4 * (f32(sf) * gs))
// (nvfp4_utils.py:39-66).  The 8 k-values of a lane lie in ONE 16-k group (group 2*kstep + g/2), so a
// fragment needs one scale per lane: fp4 -> f32 pairs (exact), one packed multiply, one RNE pack.
// `gsbits` = the expert's multiplier as f32 bits (passed where uint4b8 passes scales-per-unit).
template <int ADT>
struct Dec<LKM_W_NVFP4, ADT> {
    static constexpr int UNITK = 128, LOADS = 1, KSTEPS = 4;
    static constexpr bool UNIT_SCALE = false, A8 = false, XS = false;
    struct Aux {
        u32x2 raw;   // the eight e4m3fn block scales of this lane's weight row for the 128-k unit
    };
    // scales: [tile][unit][16 rows][8] bytes
    static __device__ __forceinline__ void load_aux(Aux& a, const void* sbase, size_t tu, int lane, int) {
        a.raw = ((const u32x2*)sbase)[tu * 16 + (lane & 15)];
    }
    static __device__ __forceinline__ const char* aux_ptr(const void* sbase, size_t tu, int lane, int) {
        return (const char*)((const u32x2*)sbase + tu * 16 + (lane & 15));
    }
    static __device__ __forceinline__ int aux_step(int) { return 128; }
    static __device__ __forceinline__ void load_aux_at(Aux& a, const char* p) { a.raw = *(const u32x2*)p; }
    static __device__ __forceinline__ u32x4 frag(const u32x4 (&raw)[LOADS], const Aux& a, int ks, int gsbits) {
        const unsigned w = raw[0][ks];
        const int grp = 2 * ks + ((threadIdx.x >> 5) & 1);   // lane bit 5 = g/2
        const unsigned long long bits = ((unsigned long long)a.raw.y << 32 | a.raw.x) >> (8 * grp);
        const float s = __builtin_amdgcn_cvt_f32_fp8((int)(unsigned)bits, 0) * __builtin_bit_cast(float, gsbits);
        u32x4 o;
        f32x2 v;
        v = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 0);
        o.x = ActT<ADT>::pack2(v.x * s, v.y * s);
        v = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 1);
        o.y = ActT<ADT>::pack2(v.x * s, v.y * s);
        v = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 2);
        o.z = ActT<ADT>::pack2(v.x * s, v.y * s);
        v = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 3);
        o.w = ActT<ADT>::pack2(v.x * s, v.y * s);
        return o;
    }
};

// fp8 weights x fp8 activations on the native fp8 MFMA (W8A8): the A fragment is the raw 8 bytes,
// no decode at all; weight block scale x token block scale is applied to the fp32 partial sum of
// every 128-k unit (native_w8a8_block_matmul, tests/kernels/quant_utils.py:91-154).
template <int ADT>
struct Dec<LKM_W_FP8_A8, ADT> {
    static constexpr int UNITK = 128, LOADS = 2, KSTEPS = 4;
    static constexpr bool UNIT_SCALE = true, A8 = true, XS = true;
    struct Aux {
        f32x4 s;
    };
    static __device__ __forceinline__ void load_aux(Aux& a, const void* sbase, size_t tu, int lane,
                                                    int) {
        a.s = *(const f32x4*)((const float*)sbase + tu * 16 + (lane >> 4) * 4);
    }
    static __device__ __forceinline__ const char* aux_ptr(const void* sbase, size_t tu, int lane, int) {
        return (const char*)((const float*)sbase + tu * 16 + (lane >> 4) * 4);
    }
    static __device__ __forceinline__ int aux_step(int) { return 64; }
    static __device__ __forceinline__ void load_aux_at(Aux& a, const char* p) { a.s = *(const f32x4*)p; }
    static __device__ __forceinline__ long frag8(const u32x4 (&raw)[LOADS], int ks) {
        const u32x2 v = {raw[ks >> 1][(ks & 1) * 2], raw[ks >> 1][(ks & 1) * 2 + 1]};
        return __builtin_bit_cast(long, v);
    }
};

template <typename D, int NTT, int TB>
struct Stage {
    // token operand: 16-bit activations -> one u32x4 (8 elements) per k-step;
    //                fp8 activations    -> one u32x4 (16 bytes) per PAIR of k-steps
    static constexpr int XN = D::A8 ? D::KSTEPS / 2 : D::KSTEPS;
    u32x4 w[NTT][D::LOADS];
    u32x4 x[TB][XN];
    float xs[TB];          // XS: per-(token, unit) scalar of this lane's token (A8: activation scale; INT4_PS: k sum)
    typename D::Aux aux[NTT];
};

// Streams units [u0,u1) of NTT tiles against TB token blocks into acc[NTT][TB].
// xp[b]  : byte pointer to this lane's token row of block b (fp8 rows when D::A8, else 16-bit rows)
// xsp[b] : A8 only, pointer to that row's per-unit activation scales
// WSU (fp8 x fp8 only): every 16-row weight tile has ONE block scale per K unit (block-quantised checkpoints: 128 x 128),
// so the scales of a tile's whole K range are fetched once -- lane l holds unit l (wsu[t][0]) and unit 64 + l
// (wsu[t][1]) -- and a unit's scale is a v_readlane instead of one 16-byte vector load per tile per unit (2 of the 12
// load instructions of a unit, 16 staging registers).
template <int WF, int ADT, int NTT, int TB, bool WSU = false>
struct Streamer {
    typedef Dec<WF, ADT> D;
    typedef Stage<D, NTT, TB> St;
    static constexpr int XB = D::A8 ? 1 : 2;
    static_assert(!WSU || D::A8, "unit-scale registers: fp8 x fp8 only");

    // NTB = token blocks actually in use (compile time): the loop below contains no conditional load,
    // which is what lets s_waitcnt leave the next stage in flight -- vector-memory loads retire in
    // order and are counted, so a load under a branch anywhere in the loop turns every wait into
    // vmcnt(0).  STEADY: the unit is not the last one of the K range, i.e. never a ragged K tail.
    template <int NTB, bool STEADY>
    static __device__ __forceinline__ void load(St& st, const u32x4* const (&wp)[NTT],
                                                const char* const (&auxp)[NTT], int aux_step, int wstep,
                                                const unsigned char* const (&xp)[TB],
                                                const float* const (&xsp)[TB], int u, int Kreal, int gk) {
#pragma unroll
        for (int t = 0; t < NTT; ++t) {
#pragma unroll
            for (int l = 0; l < D::LOADS; ++l)
                st.w[t][l] = __builtin_nontemporal_load(wp[t] + (size_t)u * wstep + l * 64);
            if constexpr (!WSU) D::load_aux_at(st.aux[t], auxp[t] + (size_t)u * aux_step);
        }
        // Token rows beyond the expert's count point at a valid row (their D columns are never
        // stored, and a B column cannot contaminate another), so the loads are unconditional.
        const bool tail = !STEADY && (u + 1) * D::UNITK > Kreal;   // wave-uniform
#pragma unroll
        for (int b = 0; b < NTB; ++b) {
            if constexpr (D::XS) st.xs[b] = xsp[b][u];     // (ablated: worth 0.6 % of the fp8 x fp8 GEMM1)
#pragma unroll
            for (int i = 0; i < St::XN; ++i) {
                // 16-byte token loads: 8 x 16-bit = k-step i, or 16 x fp8 = the k-step pair i;
                // the four g-lanes of a row are adjacent (64 contiguous bytes per row per load)
                const int k = u * D::UNITK + i * (64 / XB) + gk;
                if (!tail) {
                    st.x[b][i] = *(const u32x4*)(xp[b] + (size_t)k * XB);
                } else {
                    u32x4 v = {0u, 0u, 0u, 0u};
                    if (k + 16 / XB <= Kreal) v = *(const u32x4*)(xp[b] + (size_t)k * XB);
                    st.x[b][i] = v;
                }
            }
        }
    }

    template <int NTB>
    static __device__ __forceinline__ void compute(const St& st, f32x4 (&acc)[NTT][TB], int spu, int u,
                                                   const float (&wsu)[NTT][2]) {
        if constexpr (D::UNIT_SCALE) {
            f32x4 part[NTT][NTB];
#pragma unroll
            for (int t = 0; t < NTT; ++t)
#pragma unroll
                for (int b = 0; b < NTB; ++b) part[t][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < D::KSTEPS; ++ks)
#pragma unroll
                for (int t = 0; t < NTT; ++t) {
                    if constexpr (D::A8) {
                        // the whole 128-k unit in ONE MX-scaled MFMA (unit E8M0 scales, C = 0; twice the rate of the
                        // legacy fp8 MFMA, a quarter of the instructions): the two 16-byte loads of a lane ARE its
                        // 32-byte operand -- the weight image (repack.hip, a8) and the token loads above pair
                        // k = ld*64 + g*16 + [0,16) on both sides.  Same instruction as the prefill kernel.
                        if (ks == 0) {
                            typedef __attribute__((ext_vector_type(8))) int i32x8_t;
                            const i32x8_t a = {(int)st.w[t][0].x, (int)st.w[t][0].y, (int)st.w[t][0].z, (int)st.w[t][0].w,
                                               (int)st.w[t][1].x, (int)st.w[t][1].y, (int)st.w[t][1].z, (int)st.w[t][1].w};
#pragma unroll
                            for (int b = 0; b < NTB; ++b) {
                                const i32x8_t bb = {(int)st.x[b][0].x, (int)st.x[b][0].y, (int)st.x[b][0].z, (int)st.x[b][0].w,
                                                    (int)st.x[b][1].x, (int)st.x[b][1].y, (int)st.x[b][1].z, (int)st.x[b][1].w};
                                part[t][b] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                                    a, bb, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                            }
                        }
                    } else {
                        const u32x4 a = D::frag(st.w[t], st.aux[t], ks, spu);
#pragma unroll
                        for (int b = 0; b < NTB; ++b) part[t][b] = ActT<ADT>::mfma(a, st.x[b][ks], part[t][b]);
                    }
                }
#pragma unroll
            for (int t = 0; t < NTT; ++t)
#pragma unroll
                for (int b = 0; b < NTB; ++b) {
                    if constexpr (WSU) {
                        const float ws = __builtin_bit_cast(float, __builtin_amdgcn_readlane(
                            __builtin_bit_cast(int, u < 64 ? wsu[t][0] : wsu[t][1]), u & 63));
                        acc[t][b] += scale4(f32x4{ws, ws, ws, ws}, splat2_opaque(st.xs[b])) * part[t][b];
                    } else if constexpr (D::A8)
                        acc[t][b] += scale4(st.aux[t].s, splat2_opaque(st.xs[b])) * part[t][b];
                    else if constexpr (D::XS)      // INT4_PS: s * (sum (BIAS + v) x - (BIAS + 8) sum x)
                        acc[t][b] += st.aux[t].s * sub4(part[t][b], splat2_opaque(D::BIAS8 * st.xs[b]));
                    else
                        acc[t][b] += st.aux[t].s * part[t][b];
                }
        } else {
#pragma unroll
            for (int ks = 0; ks < D::KSTEPS; ++ks)
#pragma unroll
                for (int t = 0; t < NTT; ++t) {
                    const u32x4 a = D::frag(st.w[t], st.aux[t], ks, spu);
#pragma unroll
                    for (int b = 0; b < NTB; ++b) acc[t][b] = ActT<ADT>::mfma(a, st.x[b][ks], acc[t][b]);
                }
        }
    }

    template <int NTB>
    static __device__ __forceinline__ void run_n(f32x4 (&acc)[NTT][TB], const u32x4* const (&wp)[NTT],
                                                 const char* const (&auxp)[NTT], int aux_step, int wstep, int spu,
                                                 const unsigned char* const (&xp)[TB],
                                                 const float* const (&xsp)[TB], int u0, int u1, int Kreal,
                                                 int lane, const float (&wsu)[NTT][2]) {
        const int gk = (lane >> 4) * (16 / XB);
        // PD register stages, loads PD - 1 units ahead of their MFMAs.  (Three stages for the fp8 x fp8 kernels --
        // +34 registers, twice the weight bytes in flight per wave -- changed nothing: GEMM1 150.8 vs 151.0 us.)
        constexpr int PD = 2;
        St st[PD];
        if (u0 >= u1) return;
#pragma unroll
        for (int s0 = 0; s0 < PD - 1; ++s0)
            if (u0 + s0 < u1) load<NTB, false>(st[s0], wp, auxp, aux_step, wstep, xp, xsp, u0 + s0, Kreal, gk);
        // steady groups of PD units: every look-ahead unit stays below u1 - 1 (never the possibly ragged last unit)
        const int n_steady = u1 - PD - u0 > 0 ? (u1 - PD - u0) / PD * PD : 0;
        int u = u0;
        for (; u < u0 + n_steady; u += PD) {
#pragma unroll
            for (int h = 0; h < PD; ++h) {
                // sched_barrier: the look-ahead stage's loads are ISSUED before this stage's MFMAs (the scheduler
                // otherwise sinks them below the MFMAs that last read those registers and the prefetch distance
                // shrinks from a stage to a few instructions)
                load<NTB, true>(st[(h + PD - 1) % PD], wp, auxp, aux_step, wstep, xp, xsp, u + h + PD - 1, Kreal, gk);
                __builtin_amdgcn_sched_barrier(0);
                compute<NTB>(st[h], acc, spu, u + h, wsu);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (; u < u1; u += PD) {
#pragma unroll
            for (int h = 0; h < PD; ++h) {
                const int uu = u + h;
                if (uu < u1) {
                    if (uu + PD - 1 < u1)
                        load<NTB, false>(st[(h + PD - 1) % PD], wp, auxp, aux_step, wstep, xp, xsp, uu + PD - 1, Kreal, gk);
                    compute<NTB>(st[h], acc, spu, uu, wsu);
                }
            }
        }
    }

    // ntb (wave-uniform) selects the statically sized loop; 3 blocks run as 4 (the 4th block's rows
    // alias valid rows and are never stored)
    static __device__ __forceinline__ void run(f32x4 (&acc)[NTT][TB], const u32x4* const (&wp)[NTT],
                                               const char* const (&auxp)[NTT], int aux_step, int wstep, int spu,
                                               const unsigned char* const (&xp)[TB],
                                               const float* const (&xsp)[TB], int u0, int u1, int Kreal,
                                               int lane, int ntb, const float (&wsu)[NTT][2]) {
        if constexpr (TB == 1) {
            run_n<1>(acc, wp, auxp, aux_step, wstep, spu, xp, xsp, u0, u1, Kreal, lane, wsu);
        } else if constexpr (TB == 2) {
            if (ntb <= 1)
                run_n<1>(acc, wp, auxp, aux_step, wstep, spu, xp, xsp, u0, u1, Kreal, lane, wsu);
            else
                run_n<2>(acc, wp, auxp, aux_step, wstep, spu, xp, xsp, u0, u1, Kreal, lane, wsu);
        } else {
            static_assert(TB == 4, "token blocks per wave: 1, 2 or 4");
            if (ntb <= 1)
                run_n<1>(acc, wp, auxp, aux_step, wstep, spu, xp, xsp, u0, u1, Kreal, lane, wsu);
            else if (ntb == 2)
                run_n<2>(acc, wp, auxp, aux_step, wstep, spu, xp, xsp, u0, u1, Kreal, lane, wsu);
            else
                run_n<4>(acc, wp, auxp, aux_step, wstep, spu, xp, xsp, u0, u1, Kreal, lane, wsu);
        }
    }
};

// SiLU with the sigmoid on the transcendental unit (v_exp_f32 + v_rcp_f32: ~6 VALU per element).  Until round 6 the decode /
// tile kernels shared the CPU restatement's polynomial exp and an IEEE division here -- ~70 instructions per element behind
// per-lane branches, 1200 of the ~5700 vector instructions a wave of the int4 decode kernel executes (profiles/
// r06_int4_pmc_raw.log: 179 VALU per wave and K unit against 141 in the loop), on kernels whose bound IS the vector port.
// Same rounding points; the fp32 sigmoid differs from the restatement in its last bits (<= 2 ulp of fp32), i.e. one ulp of
// the activation dtype on ~1e-4 of the intermediate elements -- what gemm_prefill.h / gemm_prefill_a8w.h have done since
// rounds 3-4, far inside the operator's tolerance (bf16 atol 2e-2, test_moe.py:233).  Routing keeps lkm_expf: its ids and
// weights are bit-exact against the CPU restatement.
__device__ __forceinline__ float act_sigmoid_fast(float g) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(g * -1.44269504088896341f)); }
__device__ __forceinline__ float act_silu(float g) { return g * act_sigmoid_fast(g); }
// (the polynomial form, kept for ONE call site: the swigluoai fallback inside gemm_prefill_a8w.h's epilogue.  That kernel runs on
//  a fixed register map with 40 registers left to the compiler; the shorter sigmoid there moved its allocation and put a spill + a
//  vmcnt(0) into the K loop -- tests/test_a8w_codegen.py caught it.  LEGACY = true compiles that call site as it was.)
__device__ __forceinline__ float act_silu_poly(float g) { return g / (1.0f + lkm_expf(-g)); }

// ------------------------------------------------------------------ epilogues shared by all GEMM kernels
// One D fragment = 4 consecutive output features n..n+3 of one routed row.
// GEMM1: activation (SiLU-mul / swigluoai / relu2; rounding points per GemmParams::round_gemm1) and ONE
// rounding to the activation dtype; the row is `out_row` of the expert-sorted intermediate.
template <int ADT, bool GATED, bool LEGACY = false>
__device__ __forceinline__ void gemm1_act4(const GemmParams& p, const f32x4& gate, const f32x4& upv, float (&v)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float a = gate[r];
        if (p.round_gemm1) a = ActT<ADT>::to_f32(ActT<ADT>::from_f32(a));
        if constexpr (GATED) {
            float up = upv[r];
            if (p.round_gemm1) up = ActT<ADT>::to_f32(ActT<ADT>::from_f32(up));
            if (p.act_type == LKM_ACT_SWIGLUOAI) {
                const float gg = fminf(a, p.limit);
                const float uu = fmaxf(fminf(up, p.limit), -p.limit);
                if constexpr (LEGACY) v[r] = (uu + 1.0f) * gg / (1.0f + lkm_expf(-gg * p.alpha));
                else v[r] = (uu + 1.0f) * gg * act_sigmoid_fast(gg * p.alpha);
            } else if (p.round_gemm1) {
                // T(silu_f32(g)) * u  (activation_kernels.cu:57-75,157-160)
                v[r] = ActT<ADT>::to_f32(ActT<ADT>::from_f32(LEGACY ? act_silu_poly(a) : act_silu(a))) * up;
            } else {
                v[r] = (LEGACY ? act_silu_poly(a) : act_silu(a)) * up;
            }
        } else {
            const float tt = a > 0.0f ? a : 0.0f;
            v[r] = tt * tt;
        }
    }
}
template <int ADT, bool GATED, bool LEGACY = false>
__device__ __forceinline__ void store_gemm1_frag(const GemmParams& p, const f32x4& gate, const f32x4& upv,
                                                 size_t out_row, int n) {
    float v[4];
    gemm1_act4<ADT, GATED, LEGACY>(p, gate, upv, v);
    unsigned short* o = (unsigned short*)p.out + out_row * p.ldo + n;
    if (n + 4 <= p.n_real) {
        u32x2 pk;
        pk.x = ActT<ADT>::pack2(v[0], v[1]);
        pk.y = ActT<ADT>::pack2(v[2], v[3]);
        *(u32x2*)o = pk;
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)   // static r: a runtime index would spill the accumulators
            if (n + r < p.n_real) o[r] = ActT<ADT>::from_f32(v[r]);
    }
}
// GEMM2: fp32 partial of split-K slab `slab`
__device__ __forceinline__ void store_gemm2_frag(const GemmParams& p, const f32x4& v, int slab, size_t out_row, int n) {
    float* o = (float*)p.out + (size_t)slab * p.sk_stride + out_row * p.ldo + n;
    if (n + 4 <= p.n_real) {
        *(f32x4*)o = v;
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n + r < p.n_real) o[r] = v[r];
    }
}

// single-token decode: the local expert of slot k, -1 = skip (negative, or not one of this engine's experts --
// the same rule as the scatter of the batched path, dispatch.hip SlotIds)
__device__ __forceinline__ int direct_expert(const GemmParams& p, int k) {
    int e = p.direct_ids[k];
    if (e >= 0) e -= p.direct_id_off;
    return (e < 0 || e >= p.direct_E) ? -1 : e;
}

// fp8 x fp8 with one block scale per (16-row tile, K unit): lane l <- the scales of units l and 64 + l of each tile
// (Streamer<..., WSU>).  WSU is a template parameter of the two decode kernels: a kernel that carries both scale paths
// needs 198 registers (two waves per SIMD), the specialised one 164 (three).
template <typename D, int N, bool WSU>
__device__ __forceinline__ void load_unit_scales(const GemmParams& p, const size_t (&tl)[N], int lane, float (&wsu)[N][2]) {
#pragma unroll
    for (int t = 0; t < N; ++t) wsu[t][0] = wsu[t][1] = 0.0f;
    if constexpr (WSU) {
#pragma unroll
        for (int t = 0; t < N; ++t) {
            const float* sb = (const float*)p.s + tl[t] * p.U * 16;      // row 0 of (tile, unit 0): 16 floats per unit
            const int u0 = lane < p.U ? lane : p.U - 1, u1 = lane + 64 < p.U ? lane + 64 : p.U - 1;
            wsu[t][0] = sb[(size_t)u0 * 16];
            wsu[t][1] = sb[(size_t)u1 * 16];
        }
    }
}
// the host side of the choice (launchers): fp8 x fp8, one block scale per 16-row tile, at most 128 K units
template <int WF>
static bool use_unit_scales(const GemmParams& p) {
    return WF == LKM_W_FP8_A8 && p.tile_uniform_scale && p.U <= 128 && !(p.dbg & 256);      // (dbg 256: A/B switch)
}
#define LKM_STREAM_RUN(NTT_, TB_, ...) Streamer<WF, ADT, NTT_, TB_, WSU>::run(__VA_ARGS__, wsu)

template <int I>
struct SlotsC {
    static constexpr int v = I;
};

// ------------------------------------------------------------------ GEMM1 + activation
// grid = (groups, max_active_experts); block = 64*KW threads: the KW waves of a workgroup split K
// and reduce through LDS (needed when an expert has too few tiles to fill the chip, e.g. M=1).
// DIRECT (single-token decode): blockIdx.y is the slot; expert = direct_ids[slot], one row, no sort output.
template <int WF, int ADT, int NT, int TB, bool GATED, bool DIRECT = false, bool WSU = false>
__global__ __launch_bounds__(512) void gemm1_act_kernel(GemmParams p) {
    typedef Dec<WF, ADT> D;
    constexpr int NTT = GATED ? 2 * NT : NT;
    extern __shared__ __attribute__((aligned(16))) float red[];  // [NTT*TB][64] f32x4
    const int ai = blockIdx.y;
    int e, m_e, off_e;
    if constexpr (DIRECT) {
        if (p.route_on) {
            // the router of the one row, by the first wavefront of EVERY workgroup (the same device code as the router
            // kernels, so the same bits); selection k sits in lane k
            __shared__ int32_t sh_id[64];
            __shared__ float sh_ch[kMaxSlots * 64];
            if ((threadIdx.x >> 6) == 0) {
                const int ln = threadIdx.x & 63;
                float w = 0.0f;
                int id = -1;
                const RouteArgs& ra = p.route;
                const int rrow = ai / ra.K;          // this workgroup's token (slot ai = token * K + k)
                auto go = [&](auto SC) __attribute__((always_inline)) {
                    constexpr int S = decltype(SC)::v;
                    if (ra.n_group > 0)
                        grouped_topk_row<S>(ra.src, ra.bias, rrow, ra.E, ra.K, ra.n_group, ra.topk_group, ra.scoring,
                                            ra.renorm, ra.rsf, ln, sh_ch, w, id);
                    else
                        topk_row<S, 64>(ra.src, ra.bias, rrow, ra.E, ra.K, ra.scoring, ra.renorm, ra.rsf, ln, w, id);
                };
                switch (route_slots(ra.E)) {
                case 1: go(SlotsC<1>{}); break;
                case 2: go(SlotsC<2>{}); break;
                case 4: go(SlotsC<4>{}); break;
                default: go(SlotsC<8>{}); break;
                }
                if (ln < ra.K) {
                    sh_id[ln] = id;
                    if (blockIdx.x == 0 && ai % ra.K == 0) {      // one workgroup per token publishes its routing
                        ra.out_ids[(size_t)rrow * ra.K + ln] = id;
                        ra.out_w[(size_t)rrow * ra.K + ln] = w;
                    }
                }
            }
            __syncthreads();
            e = sh_id[ai % p.route.K];
            if (e >= 0) e -= p.direct_id_off;
            if (e < 0 || e >= p.direct_E) e = -1;
        } else {
            e = direct_expert(p, ai);
        }
        if (e < 0) return;
        m_e = 1;
        off_e = ai;
    } else {
        if (ai >= p.meta[0]) return;
        e = p.active[ai];
        m_e = p.counts[e];
        off_e = p.offsets[e];
        if (p.max_rows > 0 && m_e > p.max_rows) return;   // workgroup-uniform: the tiled kernel owns this expert
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, KW = blockDim.x >> 6;
    const int g = lane >> 4, j = lane & 15;
    const int tile0 = blockIdx.x * NT;
    const int T_all = p.T_half * p.halves;

    const u32x4* wp[NTT];
    const char* auxp[NTT];
    size_t tlv[NTT];
    const int aux_step = D::aux_step(p.spu), wstep = (int)p.w_ustride;
#pragma unroll
    for (int t = 0; t < NTT; ++t) {
        const int tile = (GATED && t >= NT) ? p.T_half + tile0 + (t - NT) : tile0 + t;
        const size_t tl = (size_t)e * T_all + tile;
        tlv[t] = tl;
        wp[t] = (const u32x4*)p.w + (size_t)e * p.w_estride + (size_t)tile * p.w_tstride + lane;
        auxp[t] = D::aux_ptr(p.s, tl * p.U, lane, p.spu);
    }
    float wsu[NTT][2];
    load_unit_scales<D, NTT, WSU>(p, tlv, lane, wsu);
    const int u0 = (int)((long long)wave * p.U / KW), u1 = (int)((long long)(wave + 1) * p.U / KW);
    const int dparam = WF == LKM_W_NVFP4 ? __builtin_bit_cast(int, p.gs ? p.gs[e] : 1.0f) : p.spu;

    for (int sb = 0; sb < m_e; sb += 16 * TB) {
        const int rows = min(m_e - sb, 16 * TB);
        const int ntb = (rows + 15) >> 4;
        constexpr int XB = D::A8 ? 1 : 2;
        const unsigned char* xp[TB];
        const float* xsp[TB];
#pragma unroll
        for (int b = 0; b < TB; ++b) {
            const int r = sb + b * 16 + j;
            const int slot = DIRECT ? off_e : p.sorted_slot[off_e + (r < m_e ? r : 0)];
            const int tok = slot / p.top_k;
            xp[b] = (const unsigned char*)p.x + (size_t)tok * p.ldx * XB;
            xsp[b] = p.xscale + (size_t)tok * p.ld_xscale;
        }
        f32x4 acc[NTT][TB];
#pragma unroll
        for (int t = 0; t < NTT; ++t)
#pragma unroll
            for (int b = 0; b < TB; ++b) acc[t][b] = f32x4{0.f, 0.f, 0.f, 0.f};

        LKM_STREAM_RUN(NTT, TB, acc, wp, auxp, aux_step, wstep, dparam, xp, xsp, u0, u1, p.Kreal, lane, ntb);

        if (KW > 1) {
            // fixed-order cross-wave sum: wave KW-1 stores, KW-2 .. 1 add, wave 0 takes the total
            for (int w = KW - 1; w >= 1; --w) {
                if (wave == w) {
#pragma unroll
                    for (int t = 0; t < NTT; ++t)
#pragma unroll
                        for (int b = 0; b < TB; ++b) {
                            f32x4* q = (f32x4*)red + (t * TB + b) * 64 + lane;
                            if (w == KW - 1)
                                *q = acc[t][b];
                            else
                                *q = *q + acc[t][b];
                        }
                }
                __syncthreads();
            }
            if (wave == 0) {
#pragma unroll
                for (int t = 0; t < NTT; ++t)
#pragma unroll
                    for (int b = 0; b < TB; ++b)
                        acc[t][b] = *((f32x4*)red + (t * TB + b) * 64 + lane) + acc[t][b];
            }
            __syncthreads();
        }
        if (wave == 0) {
            // D layout: lane (g,j): rows tile*16 + g*4 + r (r=0..3), token column j
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                const int r_tok = sb + b * 16 + j;
                if (b < ntb && r_tok < m_e) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int n = (tile0 + t) * 16 + g * 4;
                        if (n < p.n_real)
                            store_gemm1_frag<ADT, GATED>(p, acc[t][b], acc[GATED ? NT + t : t][b], (size_t)(off_e + r_tok), n);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------ GEMM2 (split-K partials)
// grid = (ceil(groups*SK / 4), max_active_experts); block = 256 = 4 independent waves.
template <int WF, int ADT, int NT, int TB, bool WSU = false>
__global__ __launch_bounds__(256) void gemm2_kernel(GemmParams p) {
    typedef Dec<WF, ADT> D;
    const int ai = blockIdx.y;
    if (ai >= p.meta[0]) return;
    const int e = p.active[ai];
    const int m_e = p.counts[e], off_e = p.offsets[e];
    if (p.max_rows > 0 && m_e > p.max_rows) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, j = lane & 15;
    const int uid = blockIdx.x * 4 + wave;
    if (uid >= p.groups * p.SK) return;
    const int grp = uid / p.SK, sk = uid % p.SK;
    const int tile0 = grp * NT;

    const u32x4* wp[NT];
    const char* auxp[NT];
    size_t tlv[NT];
    const int aux_step = D::aux_step(p.spu), wstep = (int)p.w_ustride;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const size_t tl = (size_t)e * p.T_half + tile0 + t;
        tlv[t] = tl;
        wp[t] = (const u32x4*)p.w + (size_t)e * p.w_estride + (size_t)(tile0 + t) * p.w_tstride + lane;
        auxp[t] = D::aux_ptr(p.s, tl * p.U, lane, p.spu);
    }
    float wsu[NT][2];
    load_unit_scales<D, NT, WSU>(p, tlv, lane, wsu);
    const int u0 = (int)((long long)sk * p.U / p.SK), u1 = (int)((long long)(sk + 1) * p.U / p.SK);
    const int dparam = WF == LKM_W_NVFP4 ? __builtin_bit_cast(int, p.gs ? p.gs[e] : 1.0f) : p.spu;

    for (int sb = 0; sb < m_e; sb += 16 * TB) {
        const int rows = min(m_e - sb, 16 * TB);
        const int ntb = (rows + 15) >> 4;
        constexpr int XB = D::A8 ? 1 : 2;
        const unsigned char* xp[TB];
        const float* xsp[TB];
#pragma unroll
        for (int b = 0; b < TB; ++b) {
            const int r = sb + b * 16 + j;
            const size_t row = (size_t)(off_e + (r < m_e ? r : 0));
            xp[b] = (const unsigned char*)p.x + row * p.ldx * XB;
            xsp[b] = p.xscale + row * p.ld_xscale;
        }
        f32x4 acc[NT][TB];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int b = 0; b < TB; ++b) acc[t][b] = f32x4{0.f, 0.f, 0.f, 0.f};

        LKM_STREAM_RUN(NT, TB, acc, wp, auxp, aux_step, wstep, dparam, xp, xsp, u0, u1, p.Kreal, lane, ntb);

#pragma unroll
        for (int b = 0; b < TB; ++b) {
            const int r_tok = sb + b * 16 + j;
            if (b < ntb && r_tok < m_e) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int n = (tile0 + t) * 16 + g * 4;
                    store_gemm2_frag(p, acc[t][b], sk, (size_t)(off_e + r_tok), n);
                }
            }
        }
    }
}

// ------------------------------------------------------------------ GEMM2 + combine, single-token decode
// grid = tile groups; block = K*SK waves: wave (k, s) streams slice s of expert ids[k]'s rows against the
// slot's intermediate row; the workgroup then forms  out[h] = sum_k w[k] * sum_s partial[k][s][h]  through
// LDS in exactly the order of combine_kernel (s ascending, then k ascending).
template <int WF, int ADT, int NT, typename OutT>
__global__ __launch_bounds__(1024) void gemm2_direct_kernel(GemmParams p, int K) {
#pragma clang fp contract(off)
    typedef Dec<WF, ADT> D;
    extern __shared__ __attribute__((aligned(16))) float red[];  // [K*SK][NT][64] f32x4
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, j = lane & 15;
    const int SK = p.SK;
    const int k = wave / SK, sk = wave % SK;
    const int tok = blockIdx.y, slot0 = tok * K;          // a few tokens: one grid row each, slots tok*K .. tok*K+K-1
    const int e = direct_expert(p, slot0 + k);
    const int tile0 = blockIdx.x * NT;
    // slot k's routing weight, fetched under the weight stream (a first-touch global load after the barrier would add
    // its whole latency to this latency-bound launch); handed to the summing lanes through LDS
    float* redw = red + (size_t)blockDim.x * NT * 4;      // [K] behind the partials
    const float wk = (sk == 0 && e >= 0) ? p.direct_w[slot0 + k] : 0.0f;
    f32x4 acc[NT][1];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (e >= 0) {
        const u32x4* wp[NT];
        const char* auxp[NT];
        size_t tlv[NT];
        const int aux_step = D::aux_step(p.spu), wstep = (int)p.w_ustride;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const size_t tl = (size_t)e * p.T_half + tile0 + t;
            tlv[t] = tl;
            wp[t] = (const u32x4*)p.w + (size_t)e * p.w_estride + (size_t)(tile0 + t) * p.w_tstride + lane;
            auxp[t] = D::aux_ptr(p.s, tl * p.U, lane, p.spu);
        }
        float wsu[NT][2] = {};
        constexpr bool WSU = false;      // (128-register kernels: the per-row scale loads stay)
        (void)tlv;
        const int u0 = (int)((long long)sk * p.U / SK), u1 = (int)((long long)(sk + 1) * p.U / SK);
        const int dparam = WF == LKM_W_NVFP4 ? __builtin_bit_cast(int, p.gs ? p.gs[e] : 1.0f) : p.spu;
        constexpr int XB = D::A8 ? 1 : 2;
        const unsigned char* xp[1] = {(const unsigned char*)p.x + (size_t)(slot0 + k) * p.ldx * XB};
        const float* xsp[1] = {p.xscale + (size_t)(slot0 + k) * p.ld_xscale};
        LKM_STREAM_RUN(NT, 1, acc, wp, auxp, aux_step, wstep, dparam, xp, xsp, u0, u1, p.Kreal, lane, 1);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) ((f32x4*)red)[(wave * NT + t) * 64 + lane] = acc[t][0];
    if (sk == 0 && lane == 0) redw[k] = wk;
    __syncthreads();
    // wave 0, lanes with token column j == 0 hold the row: D layout lane (g,0): rows tile*16 + g*4 + r
    if (wave != 0 || j != 0) return;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = (tile0 + t) * 16 + g * 4;
        if (n >= p.n_real) continue;
        f32x4 out = {0.f, 0.f, 0.f, 0.f};
        for (int kk = 0; kk < K; ++kk) {
            if (direct_expert(p, slot0 + kk) < 0) continue;
            f32x4 v = ((const f32x4*)red)[((kk * SK) * NT + t) * 64 + lane];
            for (int s = 1; s < SK; ++s) v += ((const f32x4*)red)[((kk * SK + s) * NT + t) * 64 + lane];
            out += redw[kk] * v;
        }
        OutT* o = (OutT*)p.out + (size_t)tok * p.ldo + n;
        if (n + 4 <= p.n_real) {
            store4<OutT>(o, out);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < p.n_real) store1<OutT>(o + r, out[r]);
        }
    }
}

template <int WF, int ADT>
static int launch_g2_direct_t(hipStream_t st, const LaunchCfg& cfg, const GemmParams& p, int K) {
    const int waves = K * p.SK;
    LKM_REQUIRE(waves >= 1 && waves <= 16, "gemm2 direct: K*sk=%d waves do not fit one workgroup", waves);
    dim3 grid(p.T_half / cfg.nt, (unsigned)(p.x_rows / K)), block(64 * waves);      // x_rows = M * K slots
    const size_t lds = (size_t)waves * cfg.nt * 64 * sizeof(f32x4) + 16 * sizeof(float);
    typedef typename std::conditional<ADT == LKM_DT_BF16, bf16_out, f16_out>::type ActOut;
#define LKM_G2D(NT)                                                                                       \
    if (p.direct_out_dt == LKM_DT_F32)                                                                    \
        LKM_LAUNCH_GEMM((gemm2_direct_kernel<WF, ADT, NT, float>), grid, block, lds, st, p, K);        \
    else                                                                                                  \
        LKM_LAUNCH_GEMM((gemm2_direct_kernel<WF, ADT, NT, ActOut>), grid, block, lds, st, p, K);
    if (cfg.nt == 2) {
        LKM_G2D(2)
    } else {
        LKM_G2D(1)
    }
#undef LKM_G2D
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}

// ------------------------------------------------------------------ launchers (per format TU)
template <int WF, int ADT, int NT, int TB>
static int launch_g1_t(hipStream_t st, const GemmParams& p, bool gated, int kw, int max_active) {
    dim3 grid(p.groups, max_active), block(64 * kw);
    const int ntt = gated ? 2 * NT : NT;
    const size_t lds = kw > 1 ? (size_t)ntt * TB * 64 * sizeof(f32x4) : 0;
    // only register-resident variants are built (see -Rpass-analysis=kernel-resource-usage)
    if constexpr (TB == 1 && NT == 1) {
        if (p.direct_ids) {
            if (gated) LKM_LAUNCH_GEMM((gemm1_act_kernel<WF, ADT, 1, 1, true, true>), grid, block, lds, st, p);
            else LKM_LAUNCH_GEMM((gemm1_act_kernel<WF, ADT, 1, 1, false, true>), grid, block, lds, st, p);
            LKM_HIP_CHECK(hipGetLastError());
            return LKM_OK;
        }
    }
    if (gated) {
        if constexpr (NT <= 2 && NT * TB <= 4) {
            if (use_unit_scales<WF>(p)) LKM_LAUNCH_GEMM((gemm1_act_kernel<WF, ADT, NT, TB, true, false, WF == LKM_W_FP8_A8>), grid, block, lds, st, p);
            else LKM_LAUNCH_GEMM((gemm1_act_kernel<WF, ADT, NT, TB, true>), grid, block, lds, st, p);
        } else {
            set_error("gemm1: gated variant nt=%d tb=%d is not built (register budget)", NT, TB);
            return LKM_E_INVALID;
        }
    } else {
        if constexpr (NT * TB <= 8) {
            if (use_unit_scales<WF>(p)) LKM_LAUNCH_GEMM((gemm1_act_kernel<WF, ADT, NT, TB, false, false, WF == LKM_W_FP8_A8>), grid, block, lds, st, p);
            else LKM_LAUNCH_GEMM((gemm1_act_kernel<WF, ADT, NT, TB, false>), grid, block, lds, st, p);
        } else {
            set_error("gemm1: variant nt=%d tb=%d is not built (register budget)", NT, TB);
            return LKM_E_INVALID;
        }
    }
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}

template <int WF, int ADT, int NT, int TB>
static int launch_g2_t(hipStream_t st, const GemmParams& p, int max_active) {
    dim3 grid(ceil_div(p.groups * p.SK, 4), max_active), block(256);
    if constexpr (NT * TB <= 8) {
        if (use_unit_scales<WF>(p)) LKM_LAUNCH_GEMM((gemm2_kernel<WF, ADT, NT, TB, WF == LKM_W_FP8_A8>), grid, block, 0, st, p);
        else LKM_LAUNCH_GEMM((gemm2_kernel<WF, ADT, NT, TB>), grid, block, 0, st, p);
    } else {
        set_error("gemm2: variant nt=%d tb=%d is not built (register budget)", NT, TB);
        return LKM_E_INVALID;
    }
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}

#define LKM_DISPATCH_TB(FN, WF, ADT, NT, ...)                                    \
    switch (cfg.tb) {                                                            \
    case 1: return FN<WF, ADT, NT, 1>(__VA_ARGS__);                              \
    case 2: return FN<WF, ADT, NT, 2>(__VA_ARGS__);                              \
    case 4: return FN<WF, ADT, NT, 4>(__VA_ARGS__);                              \
    default: set_error("gemm: unsupported tb=%d", cfg.tb); return LKM_E_INVALID; \
    }
#define LKM_DISPATCH_NT(FN, WF, ADT, ...)                                        \
    switch (cfg.nt) {                                                            \
    case 1: LKM_DISPATCH_TB(FN, WF, ADT, 1, __VA_ARGS__)                         \
    case 2: LKM_DISPATCH_TB(FN, WF, ADT, 2, __VA_ARGS__)                         \
    case 4: LKM_DISPATCH_TB(FN, WF, ADT, 4, __VA_ARGS__)                         \
    default: set_error("gemm: unsupported nt=%d", cfg.nt); return LKM_E_INVALID; \
    }

// Each gemm_<fmt>.hip instantiates one (weight format, activation dtype) pair:
#define LKM_DEFINE_GEMM_LAUNCHERS(SUFFIX, WF, ADT)                                              \
    int launch_gemm1_##SUFFIX(hipStream_t st, const LaunchCfg& cfg, const GemmParams& p,        \
                              bool gated, int max_active) {                                     \
        LKM_DISPATCH_NT(launch_g1_t, WF, ADT, st, p, gated, cfg.kw, max_active)                 \
    }                                                                                           \
    int launch_gemm2_##SUFFIX(hipStream_t st, const LaunchCfg& cfg, const GemmParams& p,        \
                              int max_active) {                                                 \
        LKM_DISPATCH_NT(launch_g2_t, WF, ADT, st, p, max_active)                                \
    }                                                                                           \
    int launch_gemm2_direct_##SUFFIX(hipStream_t st, const LaunchCfg& cfg, const GemmParams& p, \
                                     int K) {                                                   \
        return launch_g2_direct_t<WF, ADT>(st, cfg, p, K);                                      \
    }

}  // namespace lkm
