// probe_cvt.hip -- development probe: semantics of the gfx950 scaled conversion instructions
// (v_cvt_scalef32_pk_bf16_fp8 / _fp4 / pk32_bf16_fp6): is the f32 scale applied with its mantissa, and
// is the result one RNE rounding of the exact product?  Decides whether sub-8-bit weights can be
// dequantised to the reference's T(value * scale) in one instruction.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
typedef __attribute__((ext_vector_type(32))) __bf16 bf32;
typedef __attribute__((ext_vector_type(6))) unsigned u6;

__global__ void k_fp8(const unsigned* src, const float* scale, unsigned* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bf2 lo = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(src[i], scale[i], false);
    bf2 hi = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(src[i], scale[i], true);
    out[2 * i] = __builtin_bit_cast(unsigned, lo);
    out[2 * i + 1] = __builtin_bit_cast(unsigned, hi);
}
__global__ void k_fp4(const unsigned* src, const float* scale, unsigned* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[4 * i + 0] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(src[i], scale[i], 0));
    out[4 * i + 1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(src[i], scale[i], 1));
    out[4 * i + 2] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(src[i], scale[i], 2));
    out[4 * i + 3] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(src[i], scale[i], 3));
}
__global__ void k_fp6(const u6* src, const float* scale, bf32* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = __builtin_amdgcn_cvt_scalef32_pk32_bf16_fp6(src[i], scale[i]);
}

static uint16_t bf16_rne(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf16_f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static float fp8_f(uint8_t b) {   // e4m3fn
    int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -v : v;
}
static const uint8_t F8[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4A, 0x4C, 0x4E, 0x50};
static uint8_t fp8_of_int(int v) { return (uint8_t)((v < 0 ? 0x80 : 0) | F8[v < 0 ? -v : v]); }
static uint8_t fp6_of_half_int(int v) {   // e2m3 code of v/2, v in [-8,7]
    int h = v < 0 ? -v : v;
    int c = h <= 4 ? 4 * h : 2 * h + 8;
    return (uint8_t)((v < 0 ? 32 : 0) | c);
}
static const float F4[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};

int main() {
    const int N = 4096;
    std::vector<float> sc(N);
    uint32_t rng = 12345;
    auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return rng; };
    for (int i = 0; i < N; ++i) {
        // bf16-valued scales over a wide exponent range, both signs
        uint16_t b = (uint16_t)(rnd() >> 16);
        int e = 100 + (rnd() >> 8) % 50;
        b = (uint16_t)((b & 0x807f) | (e << 7));
        sc[i] = bf16_f(b);
    }
    sc[0] = 1.5f; sc[1] = 1.0f; sc[2] = 0.01171875f; sc[3] = 3.0f;
    float* d_sc; hipMalloc(&d_sc, N * 4); hipMemcpy(d_sc, sc.data(), N * 4, hipMemcpyHostToDevice);
    // ---- fp8
    {
        std::vector<unsigned> src(N);
        for (int i = 0; i < N; ++i) {
            int v0 = (int)(rnd() % 16) - 8, v1 = (int)(rnd() % 16) - 8, v2 = (int)(rnd() % 16) - 8, v3 = (int)(rnd() % 16) - 8;
            if (i < 4) { v0 = 3; v1 = -8; v2 = 7; v3 = -5; }
            src[i] = fp8_of_int(v0) | fp8_of_int(v1) << 8 | fp8_of_int(v2) << 16 | (unsigned)fp8_of_int(v3) << 24;
        }
        unsigned *d_s, *d_o; hipMalloc(&d_s, N * 4); hipMalloc(&d_o, N * 8);
        hipMemcpy(d_s, src.data(), N * 4, hipMemcpyHostToDevice);
        k_fp8<<<N / 256, 256>>>(d_s, d_sc, d_o, N);
        std::vector<unsigned> out(2 * N); hipMemcpy(out.data(), d_o, N * 8, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < 4; ++j) {
                float v = fp8_f((src[i] >> (8 * j)) & 0xff);
                uint16_t want = bf16_rne(v * sc[i]);
                uint16_t got = (uint16_t)(out[2 * i + j / 2] >> (16 * (j & 1)));
                if (want != got && !(((want | got) & 0x7fff) == 0)) {
                    if (bad < 5) printf("  fp8 mismatch i=%d j=%d v=%g s=%g want=%04x (%g) got=%04x (%g)\n", i, j, v, sc[i], want, bf16_f(want), got, bf16_f(got));
                    ++bad;
                }
            }
        printf("fp8->bf16 scaled: sample v=3,s=1.5 -> %g ; mismatches vs RNE(v*s): %d / %d\n", bf16_f((uint16_t)out[0]), bad, 4 * N);
    }
    // ---- fp4
    {
        std::vector<unsigned> src(N);
        for (int i = 0; i < N; ++i) src[i] = rnd();
        unsigned *d_s, *d_o; hipMalloc(&d_s, N * 4); hipMalloc(&d_o, N * 16);
        hipMemcpy(d_s, src.data(), N * 4, hipMemcpyHostToDevice);
        k_fp4<<<N / 256, 256>>>(d_s, d_sc, d_o, N);
        std::vector<unsigned> out(4 * N); hipMemcpy(out.data(), d_o, N * 16, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < 8; ++j) {
                unsigned c = (src[i] >> (4 * j)) & 15;
                float v = (c & 8 ? -1.f : 1.f) * F4[c & 7];
                uint16_t want = bf16_rne(v * sc[i]);
                uint16_t got = (uint16_t)(out[4 * i + j / 2] >> (16 * (j & 1)));
                if (want != got && !(((want | got) & 0x7fff) == 0)) {
                    if (bad < 5) printf("  fp4 mismatch i=%d j=%d v=%g s=%g want=%04x got=%04x\n", i, j, v, sc[i], want, got);
                    ++bad;
                }
            }
        printf("fp4->bf16 scaled: mismatches vs RNE(v*s) with nibble j of byte sel=j/2: %d / %d\n", bad, 8 * N);
    }
    // ---- fp6 (e2m3) pk32
    {
        std::vector<unsigned> src(6 * N, 0u);
        std::vector<int> vals(32 * N);
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < 32; ++j) {
                int v = (int)(rnd() % 16) - 8;
                vals[32 * i + j] = v;
                unsigned long long c = fp6_of_half_int(v);
                int bit = 6 * j;
                src[6 * i + bit / 32] |= (unsigned)(c << (bit % 32));
                if (bit % 32 > 26) src[6 * i + bit / 32 + 1] |= (unsigned)(c >> (32 - bit % 32));
            }
        unsigned* d_s; bf32* d_o; hipMalloc(&d_s, N * 24); hipMalloc(&d_o, N * 64);
        hipMemcpy(d_s, src.data(), N * 24, hipMemcpyHostToDevice);
        k_fp6<<<N / 256, 256>>>((const u6*)d_s, d_sc, d_o, N);
        std::vector<uint16_t> out(32 * N); hipMemcpy(out.data(), d_o, N * 64, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < 32; ++j) {
                float v = vals[32 * i + j] * 0.5f;
                uint16_t want = bf16_rne(v * sc[i]);
                uint16_t got = out[32 * i + j];
                if (want != got && !(((want | got) & 0x7fff) == 0)) {
                    if (bad < 5) printf("  fp6 mismatch i=%d j=%d v=%g s=%g want=%04x got=%04x (%g)\n", i, j, v, sc[i], want, got, bf16_f(got));
                    ++bad;
                }
            }
        printf("fp6(e2m3)->bf16 pk32 scaled: mismatches vs RNE(v*s): %d / %d\n", bad, 32 * N);
    }
    hipError_t e = hipDeviceSynchronize();
    printf("status %s\n", hipGetErrorString(e));
    return 0;
}
