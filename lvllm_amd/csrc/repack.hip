// repack.hip -- one-time conversion of the reference's expert weight layouts (SURVEY 8(a5),
// routed_experts.py:1440-1616) into the engine's MFMA-native HBM layout (lkm_common.h: WGeom).
//
// The lk_moe contract says the constructor must copy the weights (the caller frees its tensors
// right after, routed_experts.py:1420-1432), so the pre-shuffle costs nothing extra: every 16-byte
// vector of the destination is produced by one thread from <= 2 contiguous source segments.
//
// Destination (per expert):  [tile][unit][load][lane 64][16 B]
//   rows: "halves" (gate rows then up rows for w13; one half for w2), each half padded to a
//         multiple of 16*kTilePad rows with zero tiles; interleaved gate/up (swigluoai,
//         activation_kernels.cu:401-440) is de-interleaved here.
//   K   : padded with zeros to a multiple of UNITK.
#include "lkm_kernels.h"

namespace lkm {

// element (row n, k0..k0+cnt) byte address helpers per format
// PLAIN16: 2 bytes/elem; FP8: 1 byte/elem; INT4: 1/2 byte/elem
template <int WF>
__global__ __launch_bounds__(256) void repack_w_kernel(const uint8_t* __restrict__ src,
                                                       u32x4* __restrict__ dst, RepackDims d) {
    typedef WGeom<WF> G;
    const size_t nvec = (size_t)d.E * d.halves * d.T_half * d.U * G::LOADS * 64;
    size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= nvec) return;
    const int lane = (int)(v & 63);
    size_t t = v >> 6;
    const int ld = (int)(t % G::LOADS);
    t /= G::LOADS;
    const int u = (int)(t % d.U);
    t /= d.U;
    const int tile = (int)(t % (d.halves * d.T_half));
    const int e = (int)(t / (d.halves * d.T_half));
    const int i = lane & 15, g = lane >> 4;
    const int half = tile / d.T_half;
    const int idx = (tile % d.T_half) * 16 + i;
    const bool row_ok = idx < d.n_half;
    const int N = d.n_half * d.halves;
    const int n = d.interleaved ? idx * d.halves + half : half * d.n_half + idx;

    unsigned out[4] = {0u, 0u, 0u, 0u};
    if (row_ok) {
        if (WF == LKM_W_BF16 || WF == LKM_W_F16) {
            const int k0 = u * 64 + ld * 32 + g * 8;
            const unsigned short* p = (const unsigned short*)src + ((size_t)e * N + n) * d.K;
            unsigned short h[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = (k0 + j < d.K) ? p[k0 + j] : (unsigned short)0;
#pragma unroll
            for (int j = 0; j < 4; ++j) out[j] = (unsigned)h[2 * j] | ((unsigned)h[2 * j + 1] << 16);
        } else if (WF == LKM_W_FP8_E4M3) {
            const uint8_t* p = src + ((size_t)e * N + n) * d.K;
#pragma unroll
            for (int q = 0; q < 2; ++q) {  // kstep = 2*ld + q
                // 16-bit activations: k-step 2ld+q takes k = ks*32 + g*8 (4 lanes = 64 contiguous
                // bytes of a token row).  fp8 activations (d.a8): one 16-byte token load feeds the
                // k-step PAIR ld, so lane g owns k = ld*64 + g*16 + [0,16): bytes 0-7 -> step 2ld, 8-15 -> 2ld+1
                const int k0 = d.a8 ? u * 128 + ld * 64 + g * 16 + q * 8 : u * 128 + (2 * ld + q) * 32 + g * 8;
                unsigned lo = 0, hi = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    lo |= (unsigned)((k0 + j < d.K) ? p[k0 + j] : 0) << (8 * j);
                    hi |= (unsigned)((k0 + 4 + j < d.K) ? p[k0 + 4 + j] : 0) << (8 * j);
                }
                out[2 * q] = lo;
                out[2 * q + 1] = hi;
            }
        } else {  // 4-bit: bytes [E][N][K/2]; zero padding must decode to 0 => uint4b8 nibble 8, E2M1 nibble 0
            constexpr unsigned PADB = (WF == LKM_W_INT4_B8 || WF == LKM_W_INT4_PS) ? 0x88u : 0x00u;
            const uint8_t* p = src + ((size_t)e * N + n) * (d.K / 2);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k0 = u * 128 + s * 32 + g * 8;
                unsigned w = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned b = (k0 + 2 * j < d.K) ? p[(k0 >> 1) + j] : PADB;
                    w |= b << (8 * j);
                }
                if (WF == LKM_W_INT4_PS) {   // k -> nibble position k/2 + 4 (k & 1): pair p = (w >> 4p) & 0x000f000f
                    unsigned r = 0;
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) r |= ((w >> (4 * kk)) & 0xfu) << (4 * ((kk >> 1) + 4 * (kk & 1)));
                    w = r;
                }
                out[s] = w;
            }
        }
    } else if (WF == LKM_W_INT4_B8 || WF == LKM_W_INT4_PS) {   // padded rows decode to 0 (E2M1: the zero fill already does)
        out[0] = out[1] = out[2] = out[3] = 0x88888888u;
    }
    u32x4 o;
    o.x = out[0];
    o.y = out[1];
    o.z = out[2];
    o.w = out[3];
    dst[v] = o;
}

// int4 group scales: src act-dtype [E][N][K/g]  ->  dst [E][tile][unit][16 rows][SPU] (same dtype)
__global__ __launch_bounds__(256) void repack_s_int4_kernel(const unsigned short* __restrict__ src,
                                                            unsigned short* __restrict__ dst,
                                                            RepackDims d, int group, int spu) {
    const size_t n_out = (size_t)d.E * d.halves * d.T_half * d.U * 16 * spu;
    size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= n_out) return;
    const int j = (int)(v % spu);
    size_t t = v / spu;
    const int i = (int)(t & 15);
    t >>= 4;
    const int u = (int)(t % d.U);
    t /= d.U;
    const int tile = (int)(t % (d.halves * d.T_half));
    const int e = (int)(t / (d.halves * d.T_half));
    const int half = tile / d.T_half;
    const int idx = (tile % d.T_half) * 16 + i;
    const int N = d.n_half * d.halves;
    const int n = d.interleaved ? idx * d.halves + half : half * d.n_half + idx;
    const int k = u * 128 + j * (128 / spu);
    unsigned short s = 0;
    if (idx < d.n_half && k < d.K) s = src[((size_t)e * N + n) * (d.K / group) + k / group];
    dst[v] = s;
}

// uint4 with zero points: the scale image holds (scale, zero point) PAIRS -- dst [E][tile][unit][16 rows][SPU][2] act dtype.
// Two passes over one destination (the hand-off stages one source at a time): which = 0 copies the act-dtype scales
// src [E][N][K/g] into the even half-words, which = 1 converts the zero points src uint8 [E][N][K/g] (0..15) to the act dtype
// (exact) into the odd ones.  Padding rows / units: scale 0 (their weights decode to (8 - zp) * 0 = 0).
template <int ADT>
__global__ __launch_bounds__(256) void repack_s_int4zp_kernel(const void* __restrict__ src, unsigned short* __restrict__ dst,
                                                              RepackDims d, int group, int spu, int which) {
    const size_t n_out = (size_t)d.E * d.halves * d.T_half * d.U * 16 * spu;
    size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= n_out) return;
    const int j = (int)(v % spu);
    size_t t = v / spu;
    const int i = (int)(t & 15);
    t >>= 4;
    const int u = (int)(t % d.U);
    t /= d.U;
    const int tile = (int)(t % (d.halves * d.T_half));
    const int e = (int)(t / (d.halves * d.T_half));
    const int half = tile / d.T_half;
    const int idx = (tile % d.T_half) * 16 + i;
    const int N = d.n_half * d.halves;
    const int n = d.interleaved ? idx * d.halves + half : half * d.n_half + idx;
    const int k = u * 128 + j * (128 / spu);
    unsigned short s = 0;
    if (idx < d.n_half && k < d.K) {
        const size_t at = ((size_t)e * N + n) * (d.K / group) + k / group;
        if (which == 0) s = ((const unsigned short*)src)[at];
        else s = ActT<ADT>::from_f32((float)(((const uint8_t*)src)[at] & 15));
    }
    dst[2 * v + which] = s;
}

int launch_repack_s_int4zp(hipStream_t st, const void* src, void* dst, const RepackDims& d, int group, int spu, int which, int adt) {
    const size_t n_out = (size_t)d.E * d.halves * d.T_half * d.U * 16 * spu;
    dim3 grid((unsigned)ceil_div64((int64_t)n_out, 256)), block(256);
    if (adt == LKM_DT_BF16)
        hipLaunchKernelGGL(repack_s_int4zp_kernel<LKM_DT_BF16>, grid, block, 0, st, src, (unsigned short*)dst, d, group, spu, which);
    else
        hipLaunchKernelGGL(repack_s_int4zp_kernel<LKM_DT_F16>, grid, block, 0, st, src, (unsigned short*)dst, d, group, spu, which);
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}

// int4 fast mode: src act-dtype group scales [E][N][K/g] (g a multiple of 128) -> dst fp32 [E][tile][unit][16 rows]
template <int ADT>
__global__ __launch_bounds__(256) void repack_s_int4ps_kernel(const unsigned short* __restrict__ src,
                                                              float* __restrict__ dst, RepackDims d, int group) {
    const size_t n_out = (size_t)d.E * d.halves * d.T_half * d.U * 16;
    size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= n_out) return;
    size_t t = v;
    const int i = (int)(t & 15);
    t >>= 4;
    const int u = (int)(t % d.U);
    t /= d.U;
    const int tile = (int)(t % (d.halves * d.T_half));
    const int e = (int)(t / (d.halves * d.T_half));
    const int half = tile / d.T_half;
    const int idx = (tile % d.T_half) * 16 + i;
    const int N = d.n_half * d.halves;
    const int n = d.interleaved ? idx * d.halves + half : half * d.n_half + idx;
    const int k = u * 128;
    float s = 0.0f;
    if (idx < d.n_half && k < d.K) s = ActT<ADT>::to_f32(src[((size_t)e * N + n) * (d.K / group) + k / group]);
    dst[v] = s;
}

int launch_repack_s_int4ps(hipStream_t st, const void* src, void* dst, const RepackDims& d, int group, int adt) {
    const size_t n_out = (size_t)d.E * d.halves * d.T_half * d.U * 16;
    dim3 grid((unsigned)ceil_div64((int64_t)n_out, 256)), block(256);
    if (adt == LKM_DT_BF16)
        hipLaunchKernelGGL(repack_s_int4ps_kernel<LKM_DT_BF16>, grid, block, 0, st, (const unsigned short*)src, (float*)dst, d, group);
    else
        hipLaunchKernelGGL(repack_s_int4ps_kernel<LKM_DT_F16>, grid, block, 0, st, (const unsigned short*)src, (float*)dst, d, group);
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}

// 4-bit float block scales (one byte each): src [E][N][K/g] (MXFP4: E8M0, g=32; NVFP4: e4m3fn, g=16)
//   -> dst [E][tile][unit][16 rows][128/g bytes]: the 4 (8) scales of a row for one 128-k unit are
//   one dword (two) per lane.  Padding rows/units get `pad` (their weights are zero anyway).
__global__ __launch_bounds__(256) void repack_s_fp4_kernel(const uint8_t* __restrict__ src,
                                                           uint8_t* __restrict__ dst, RepackDims d,
                                                           int group, int pad) {
    const int spu = 128 / group;
    const size_t n_out = (size_t)d.E * d.halves * d.T_half * d.U * 16 * spu;
    size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= n_out) return;
    const int j = (int)(v % spu);
    size_t t = v / spu;
    const int i = (int)(t & 15);
    t >>= 4;
    const int u = (int)(t % d.U);
    t /= d.U;
    const int tile = (int)(t % (d.halves * d.T_half));
    const int e = (int)(t / (d.halves * d.T_half));
    const int half = tile / d.T_half;
    const int idx = (tile % d.T_half) * 16 + i;
    const int N = d.n_half * d.halves;
    const int n = d.interleaved ? idx * d.halves + half : half * d.n_half + idx;
    const int k = u * 128 + j * group;
    uint8_t s = (uint8_t)pad;
    if (idx < d.n_half && k < d.K) s = src[((size_t)e * N + n) * (d.K / group) + k / group];
    dst[v] = s;
}

// fp8 block scales: src fp32 [E][ceil(N/gN)][ceil(K/gK)] -> dst fp32 [E][tile][unit][16 rows]
__global__ __launch_bounds__(256) void repack_s_fp8_kernel(const float* __restrict__ src,
                                                           float* __restrict__ dst, RepackDims d,
                                                           int gN, int gK) {
    const size_t n_out = (size_t)d.E * d.halves * d.T_half * d.U * 16;
    size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= n_out) return;
    size_t t = v;
    const int i = (int)(t & 15);
    t >>= 4;
    const int u = (int)(t % d.U);
    t /= d.U;
    const int tile = (int)(t % (d.halves * d.T_half));
    const int e = (int)(t / (d.halves * d.T_half));
    const int half = tile / d.T_half;
    const int idx = (tile % d.T_half) * 16 + i;
    const int N = d.n_half * d.halves;
    const int n = d.interleaved ? idx * d.halves + half : half * d.n_half + idx;
    const int k = u * 128;
    const int NB = ceil_div(N, gN), KB = ceil_div(d.K, gK);
    float s = 0.0f;
    if (idx < d.n_half && k < d.K) s = src[((size_t)e * NB + n / gN) * KB + k / gK];
    dst[v] = s;
}

int launch_repack_w(hipStream_t st, int wf, const void* src, void* dst, const RepackDims& d) {
    const int loads = wf_loads(wf == LKM_W_INT4_PS ? LKM_W_INT4_B8 : wf);
    const size_t nvec = (size_t)d.E * d.halves * d.T_half * d.U * loads * 64;
    dim3 grid((unsigned)ceil_div64((int64_t)nvec, 256)), block(256);
    switch (wf) {
    case LKM_W_BF16:
        hipLaunchKernelGGL(repack_w_kernel<LKM_W_BF16>, grid, block, 0, st, (const uint8_t*)src, (u32x4*)dst, d);
        break;
    case LKM_W_F16:
        hipLaunchKernelGGL(repack_w_kernel<LKM_W_F16>, grid, block, 0, st, (const uint8_t*)src, (u32x4*)dst, d);
        break;
    case LKM_W_FP8_E4M3:
        hipLaunchKernelGGL(repack_w_kernel<LKM_W_FP8_E4M3>, grid, block, 0, st, (const uint8_t*)src, (u32x4*)dst, d);
        break;
    case LKM_W_INT4_B8:
        hipLaunchKernelGGL(repack_w_kernel<LKM_W_INT4_B8>, grid, block, 0, st, (const uint8_t*)src, (u32x4*)dst, d);
        break;
    case LKM_W_INT4_PS:
        hipLaunchKernelGGL(repack_w_kernel<LKM_W_INT4_PS>, grid, block, 0, st, (const uint8_t*)src, (u32x4*)dst, d);
        break;
    case LKM_W_MXFP4:
    case LKM_W_NVFP4:
        hipLaunchKernelGGL(repack_w_kernel<LKM_W_MXFP4>, grid, block, 0, st, (const uint8_t*)src, (u32x4*)dst, d);
        break;
    default:
        set_error("repack: unsupported weight format %d", wf);
        return LKM_E_UNSUPPORTED;
    }
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}

int launch_repack_s_int4(hipStream_t st, const void* src, void* dst, const RepackDims& d, int group,
                         int spu) {
    const size_t n_out = (size_t)d.E * d.halves * d.T_half * d.U * 16 * spu;
    hipLaunchKernelGGL(repack_s_int4_kernel, dim3((unsigned)ceil_div64((int64_t)n_out, 256)),
                       dim3(256), 0, st, (const unsigned short*)src, (unsigned short*)dst, d, group,
                       spu);
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}

int launch_repack_s_fp4(hipStream_t st, const void* src, void* dst, const RepackDims& d, int group,
                        int pad) {
    const size_t n_out = (size_t)d.E * d.halves * d.T_half * d.U * 16 * (128 / group);
    hipLaunchKernelGGL(repack_s_fp4_kernel, dim3((unsigned)ceil_div64((int64_t)n_out, 256)),
                       dim3(256), 0, st, (const uint8_t*)src, (uint8_t*)dst, d, group, pad);
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}

int launch_repack_s_fp8(hipStream_t st, const void* src, void* dst, const RepackDims& d, int gN,
                        int gK) {
    const size_t n_out = (size_t)d.E * d.halves * d.T_half * d.U * 16;
    hipLaunchKernelGGL(repack_s_fp8_kernel, dim3((unsigned)ceil_div64((int64_t)n_out, 256)),
                       dim3(256), 0, st, (const float*)src, (float*)dst, d, gN, gK);
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}

// ---- weight-only integer experts with zero points / 8 bits (the in-tree operator's int4_w4a16 / int8_w8a16 schemes,
// fused_moe.py:207-276) -> 16-bit weights, T((q - zp) * s) with ONE rounding: exactly what that kernel feeds tl.dot.
// One thread produces 8 consecutive k of one row (16 bytes out); q: [R][K/2] nibbles (low = even k) or [R][K] bytes;
// scales [R][K/group] act dtype; zp: 4-bit [R/2][K/group] (low nibble = even row, rows within an expert: N even) or
// 8-bit [R][K/group]; null = symmetric (8 / 128).  R = E*N rows.
template <int ADT, int BITS>
__global__ __launch_bounds__(256) void wna16_expand_kernel(const uint8_t* __restrict__ q, const unsigned short* __restrict__ sc,
                                                           const uint8_t* __restrict__ zp, u32x4* __restrict__ out,
                                                           int64_t R, int K, int group) {
    typedef ActT<ADT> A;
    const int K8 = K >> 3, KG = K / group;
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= R * K8) return;
    const int64_t r = v / K8;
    const int k0 = (int)(v - r * K8) * 8;
    const int g = k0 / group;                       // group is a multiple of 8: the 8 weights share scale and zp
    const float s = A::to_f32(sc[r * KG + g]);
    float z = BITS == 4 ? 8.0f : 128.0f;
    if (zp) z = BITS == 4 ? (float)((zp[(r >> 1) * KG + g] >> ((r & 1) * 4)) & 0xF) : (float)zp[r * KG + g];
    float w[8];
    if (BITS == 4) {
        const unsigned b = *(const unsigned*)(q + r * (K >> 1) + (k0 >> 1));
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = (float)((b >> (4 * j)) & 0xF);
    } else {
        const uint2 b = *(const uint2*)(q + r * K + k0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            w[j] = (float)((b.x >> (8 * j)) & 0xFF);
            w[4 + j] = (float)((b.y >> (8 * j)) & 0xFF);
        }
    }
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = A::pack2((w[2 * j] - z) * s, (w[2 * j + 1] - z) * s);
    out[v] = o;
}

int launch_wna16_expand(hipStream_t st, const void* q, const void* scales, const void* zp, void* out, int64_t rows,
                        int K, int group, int bits, int adt) {
    const int64_t nvec = rows * (K >> 3);
    if (nvec == 0) return LKM_OK;
    const dim3 grid((unsigned)ceil_div64(nvec, 256)), block(256);
#define LKM_WNA16_CASE(ADT_, BITS_)                                                                                     \
    hipLaunchKernelGGL((wna16_expand_kernel<ADT_, BITS_>), grid, block, 0, st, (const uint8_t*)q,                       \
                       (const unsigned short*)scales, (const uint8_t*)zp, (u32x4*)out, rows, K, group)
    if (adt == LKM_DT_BF16 && bits == 4) LKM_WNA16_CASE(LKM_DT_BF16, 4);
    else if (adt == LKM_DT_BF16) LKM_WNA16_CASE(LKM_DT_BF16, 8);
    else if (bits == 4) LKM_WNA16_CASE(LKM_DT_F16, 4);
    else LKM_WNA16_CASE(LKM_DT_F16, 8);
#undef LKM_WNA16_CASE
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}

}  // namespace lkm
