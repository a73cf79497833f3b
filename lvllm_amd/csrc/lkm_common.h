// lkm_common.h -- shared types / device helpers for the gfx950 MoE expert path.
// Written for MI355X (gfx950, wave64) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lkm.h"

namespace lkm {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int kWave = 64;

// ------------------------------------------------------------------ host-side error plumbing
void set_error(const char* fmt, ...);
#define LKM_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            ::lkm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),      \
                             __FILE__, __LINE__);                                        \
            return LKM_E_HIP;                                                            \
        }                                                                                \
    } while (0)
#define LKM_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            ::lkm::set_error(__VA_ARGS__);     \
            return LKM_E_INVALID;              \
        }                                      \
    } while (0)

// ------------------------------------------------------------------ device math
// Deterministic expf: the operation sequence is shared bit-for-bit with
// oracle/lkm_oracle.c: lkm_or_expf (explicit fma, contraction off), so routing scores and
// therefore routing ids/weights are reproducible between the GPU and the CPU oracle.
// Stands in for the reference's expf (topk_softmax_kernels.cu:428, activation_kernels.cu).
__device__ __forceinline__ float lkm_expf(float x) {
#pragma clang fp contract(off)
    if (!(x == x)) return x;
    if (x > 88.72283f) return __builtin_inff();
    if (x < -103.97f) return 0.0f;
    const float LOG2E = 1.44269504088896341f;
    const float LN2_HI = 0.693145751953125f;
    const float LN2_LO = 1.42860682030941723e-6f;
    float n = __builtin_rintf(x * LOG2E);
    float r = __builtin_fmaf(-n, LN2_HI, x);
    r = __builtin_fmaf(-n, LN2_LO, r);
    float p = 1.9841270e-4f;
    p = __builtin_fmaf(p, r, 1.3888889e-3f);
    p = __builtin_fmaf(p, r, 8.3333338e-3f);
    p = __builtin_fmaf(p, r, 4.1666668e-2f);
    p = __builtin_fmaf(p, r, 1.6666667e-1f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    int ni = (int)n;
    int n1 = ni / 2, n2 = ni - n1;
    float s1 = __builtin_bit_cast(float, (unsigned)(n1 + 127) << 23);
    float s2 = __builtin_bit_cast(float, (unsigned)(n2 + 127) << 23);
    return (p * s1) * s2;
}

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short h) {
    return __builtin_bit_cast(float, (unsigned)h << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    return __builtin_bit_cast(unsigned short, (__bf16)f);  // v_cvt_pk_bf16_f32: RNE
}
__device__ __forceinline__ float f16_bits_to_f32(unsigned short h) {
    return (float)__builtin_bit_cast(_Float16, h);
}
__device__ __forceinline__ unsigned short f32_to_f16_bits(float f) {
    return __builtin_bit_cast(unsigned short, (_Float16)f);
}

// Activation-dtype traits: the MFMA flavour and the scalar conversions.
// {v, v} in a register pair the compiler cannot see through, and a 4-vector scaled by it.  Packed fp32
// instructions (v_pk_mul/fma/add_f32) then read the pair with the default op_sel.  MEASURED on MI355X
// (tools/probe_hazard.hip): a packed instruction whose LOW lane takes the HIGH dword of a VGPR pair
// (op_sel = 1) returns 0 for that operand in lanes 48..63, in ~0.05 % of the executions, when the same
// pair is also read through another swizzle (same instruction or one nearby) while MFMAs are in flight --
// which is what the compiler emits for `vec * scalar` when two scalars share a pair -- and in ~0.003 % even
// when that hi->lo read of src1 is the only one.  Wait states do not help; default selects and lo-broadcasts
// are always right.  tools/scan_pk_swizzle.py rejects any packed-fp32 VGPR source with op_sel = 1.
static __device__ __forceinline__ f32x2 splat2_opaque(float v) {
    f32x2 p = {v, v};
    asm("" : "+v"(p));
    return p;
}
static __device__ __forceinline__ f32x4 scale4(f32x4 a, f32x2 b) {
    const f32x2 lo = f32x2{a.x, a.y} * b, hi = f32x2{a.z, a.w} * b;
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}

static __device__ __forceinline__ f32x4 sub4(f32x4 a, f32x2 b) {
    const f32x2 lo = f32x2{a.x, a.y} - b, hi = f32x2{a.z, a.w} - b;
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}

template <int DT>
struct ActT;
template <>
struct ActT<LKM_DT_BF16> {
    typedef bf16x8 vec8;
    static __device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                       __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ float to_f32(unsigned short h) { return bf16_bits_to_f32(h); }
    static __device__ __forceinline__ unsigned short from_f32(float f) { return f32_to_bf16_bits(f); }
    // pack two f32 -> two act elements in one dword (lo = a)
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        const f32x2 v = {a, b};   // one v_cvt_pk_bf16_f32 (RNE)
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    }
};
template <>
struct ActT<LKM_DT_F16> {
    typedef f16x8 vec8;
    static __device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a),
                                                      __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ float to_f32(unsigned short h) { return f16_bits_to_f32(h); }
    static __device__ __forceinline__ unsigned short from_f32(float f) { return f32_to_f16_bits(f); }
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        const f32x2 v = {a, b};   // one v_cvt_pk_f16_f32 (RNE)
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
    }
};

__host__ __device__ constexpr inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ constexpr inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------ typed output stores (fp32 / act dtype)
template <typename OutT>
__device__ __forceinline__ void store4(OutT* p, f32x4 v);
template <>
__device__ __forceinline__ void store4<float>(float* p, f32x4 v) {
    *(f32x4*)p = v;
}
struct bf16_out { unsigned short v; };
struct f16_out { unsigned short v; };
template <>
__device__ __forceinline__ void store4<bf16_out>(bf16_out* p, f32x4 v) {
    u32x2 o;
    o.x = ActT<LKM_DT_BF16>::pack2(v.x, v.y);
    o.y = ActT<LKM_DT_BF16>::pack2(v.z, v.w);
    *(u32x2*)p = o;
}
template <>
__device__ __forceinline__ void store4<f16_out>(f16_out* p, f32x4 v) {
    u32x2 o;
    o.x = ActT<LKM_DT_F16>::pack2(v.x, v.y);
    o.y = ActT<LKM_DT_F16>::pack2(v.z, v.w);
    *(u32x2*)p = o;
}

// four consecutive values of a typed array as fp32
template <typename T>
__device__ __forceinline__ f32x4 load4(const T* p);
template <>
__device__ __forceinline__ f32x4 load4<float>(const float* p) {
    return *(const f32x4*)p;
}
template <>
__device__ __forceinline__ f32x4 load4<bf16_out>(const bf16_out* p) {
    const u32x2 v = *(const u32x2*)p;
    return f32x4{bf16_bits_to_f32((unsigned short)(v.x & 0xffffu)), bf16_bits_to_f32((unsigned short)(v.x >> 16)),
                 bf16_bits_to_f32((unsigned short)(v.y & 0xffffu)), bf16_bits_to_f32((unsigned short)(v.y >> 16))};
}
template <>
__device__ __forceinline__ f32x4 load4<f16_out>(const f16_out* p) {
    const u32x2 v = *(const u32x2*)p;
    return f32x4{f16_bits_to_f32((unsigned short)(v.x & 0xffffu)), f16_bits_to_f32((unsigned short)(v.x >> 16)),
                 f16_bits_to_f32((unsigned short)(v.y & 0xffffu)), f16_bits_to_f32((unsigned short)(v.y >> 16))};
}

template <typename OutT>
__device__ __forceinline__ void store1(OutT* p, float v);
template <>
__device__ __forceinline__ void store1<float>(float* p, float v) {
    *p = v;
}
template <>
__device__ __forceinline__ void store1<bf16_out>(bf16_out* p, float v) {
    p->v = f32_to_bf16_bits(v);
}
template <>
__device__ __forceinline__ void store1<f16_out>(f16_out* p, float v) {
    p->v = f32_to_f16_bits(v);
}

// ------------------------------------------------------------------ pre-shuffled weight geometry
// A weight matrix [N rows][K] (K contiguous, "B^T" form) is stored per expert as
//   [tile = n/16][unit = k/UNITK][load][lane 0..63][16 bytes]
// so that one wave-wide global_load_dwordx4 fetches 1 KiB of contiguous HBM that is already the
// MFMA A-operand fragment(s) of mfma_f32_16x16x32: lane l = g*16 + i holds row tile*16+i,
// k = unit*UNITK + kstep*32 + g*8 + (0..7): the four g-lanes of a row read 64 contiguous bytes of
// a 16-bit token row per load.  (An MFMA sums over its 32 k-values, so any A/B-consistent k
// assignment is valid.  Measured alternative -- each lane owning a contiguous 32-byte range -- was
// 4 % slower on GEMM2: the loads become 32-byte strided.)  With fp8 ACTIVATIONS (W8A8) one 16-byte
// token load feeds a PAIR of k-steps: k = unit*128 + pair*64 + g*16 + (0..15).
//   bf16/f16 : UNITK = 64,  LOADS = 2 (load = kstep),          one dwordx4 = 8 elements
//   fp8 e4m3 : UNITK = 128, LOADS = 2 (load l: .xy = kstep 2l, .zw = kstep 2l+1; 2 x 8 bytes)
//   uint4b8 / E2M1 (MXFP4, NVFP4) : UNITK = 128, LOADS = 1 (dword s = kstep s), one dwordx4 = 4 x 8 nibbles
// internal kernel-format code: fp8 weights consumed by the native fp8 MFMA against dynamically
// quantised fp8 activations (W8A8, the in-tree operator's block-fp8 semantics); same HBM layout as
// LKM_W_FP8_E4M3.
#define LKM_W_FP8_A8 100
// internal kernel-format code: uint4b8 weights in the FAST mode (LkmConfig.int4_mode = LKM_INT4_FAST): nibbles become
// the activation-dtype values BIAS + v with one v_and_or_b32 per pair (exact), the group scale is applied to the
// fp32 partial sum of each 128-k unit and the bias leaves through the per-(token, unit) activation sums:
//   sum_k s (v_k - 8) x_k  =  s (sum_k (BIAS + v_k) x_k  -  (BIAS + 8) sum_k x_k).
// Same bytes as LKM_W_INT4_B8 with the nibbles of a dword re-ordered (k -> position k/2 + 4 (k & 1)), fp32 scales in
// the fp8 unit layout.  NOT the reference's rounding (it rounds (q-8) s to the activation dtype first): opt-in.
#define LKM_W_INT4_PS 101
// internal kernel-format code: uint4 weights with ZERO POINTS (LkmConfig.int4_mode = LKM_INT4_ZP): the LKM_W_INT4_B8 weight
// image; the scale image holds (scale, zero point) PAIRS in the activation dtype -- [tile][unit][16 rows][spu][2] -- and the
// decoder's addend is -zp * s instead of -8 * s (exact: zp <= 15 times a 16-bit scale has <= 15 significant bits).
#define LKM_W_INT4_ZP 102

template <int WF>
struct WGeom;
template <>
struct WGeom<LKM_W_BF16> {
    static constexpr int UNITK = 64, LOADS = 2, KSTEPS = 2;
};
template <>
struct WGeom<LKM_W_F16> {
    static constexpr int UNITK = 64, LOADS = 2, KSTEPS = 2;
};
template <>
struct WGeom<LKM_W_FP8_E4M3> {
    static constexpr int UNITK = 128, LOADS = 2, KSTEPS = 4;
};
template <>
struct WGeom<LKM_W_INT4_B8> {
    static constexpr int UNITK = 128, LOADS = 1, KSTEPS = 4;
};

template <>
struct WGeom<LKM_W_INT4_PS> {
    static constexpr int UNITK = 128, LOADS = 1, KSTEPS = 4;
};

template <>
struct WGeom<LKM_W_INT4_ZP> {
    static constexpr int UNITK = 128, LOADS = 1, KSTEPS = 4;
};

template <>
struct WGeom<LKM_W_MXFP4> {   // 4-bit formats share the uint4b8 geometry (dword s = kstep s)
    static constexpr int UNITK = 128, LOADS = 1, KSTEPS = 4;
};
template <>
struct WGeom<LKM_W_NVFP4> {
    static constexpr int UNITK = 128, LOADS = 1, KSTEPS = 4;
};

// ---- x / s, correctly rounded, for the 1 x 128 fp8 activation quantiser's operands (dispatch.hip quant_fp8_rows_kernel,
// gemm_prefill_a8w.h fused epilogue): s = amax / 448 >= 2.2e-13 is normal and |x| <= amax, so |x / s| <= 448 and none of the
// range handling of the general division sequence can trigger; the reciprocal is refined ONCE per group and each quotient
// takes the two remainder corrections of the IEEE sequence.  Bit-exactness against x / s: tests/test_gpu_quant.py.
struct DivBy {
    float s, r;
};
__device__ __forceinline__ DivBy make_div_by(float s) {
    const float r0 = __builtin_amdgcn_rcpf(s);
    const float e = __builtin_fmaf(-s, r0, 1.0f);
    return DivBy{s, __builtin_fmaf(e, r0, r0)};
}
__device__ __forceinline__ float div_by(float x, const DivBy& d) {
    const float q0 = x * d.r;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-q0, d.s, x), d.r, q0);
    // (the corrections turn -0 / s into +0: the sign of the quotient is the sign of x)
    return __builtin_copysignf(__builtin_fmaf(__builtin_fmaf(-q1, d.s, x), d.r, q1), x);
}


inline bool wf_is_4bit(int wf) { return wf == LKM_W_INT4_B8 || wf == LKM_W_INT4_ZP || wf == LKM_W_MXFP4 || wf == LKM_W_NVFP4; }
inline int wf_unitk(int wf) { return (wf == LKM_W_BF16 || wf == LKM_W_F16) ? 64 : 128; }
inline int wf_loads(int wf) { return wf_is_4bit(wf) ? 1 : 2; }

}  // namespace lkm
