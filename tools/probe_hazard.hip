// probe_hazard.hip -- development probe: does gfx950 interlock a v_cvt_pk_f32_fp8 (64-bit result)
// against a dependent v_pk_fma_f32 / VALU issued in the very next slot?  The int4 decoder
// (gemm_skinny.h Dec<LKM_W_INT4_B8>) came out wrong in the tiled kernels exactly where the compiler had
// scheduled those two back to back.  Each mode runs the same dependent chain with a different gap.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int MODE>
__global__ void k(const unsigned* win, const float* sin_, f32x2* out, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned w = win[i];
    const float s = sin_[i];
    const f32x2 sm = {512.0f * s, -8.0f * s};
    f32x4 acc = {1000.f, 1000.f, 1000.f, 1000.f};   // stays 1000: the MFMA operands are zero
    bf16x8 za = {}, zb = {};
    for (int q = 0; q < 8; ++q) za[q] = (__bf16)3.0f;   // A operand 3.0 everywhere, B zero
    for (int it = 0; it < iters; ++it) {
        f32x2 r;
        unsigned t;
        if (MODE == 0) {   // reference: compiler-scheduled C
            const f32x2 c = __builtin_amdgcn_cvt_pk_f32_fp8(w & 0x0f0f0f0fu, false);
            r = f32x2{__builtin_fmaf(c.x, sm.x, sm.y), __builtin_fmaf(c.y, sm.x, sm.y)};
        } else if (MODE == 1) {   // cvt e32 -> pk_fma next slot
            asm volatile("v_and_b32 %1, 0x0f0f0f0f, %2\n s_nop 4\n v_cvt_pk_f32_fp8_e32 %0, %1\n"
                         "v_pk_fma_f32 %0, %0, %3, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n"
                         : "=&v"(r), "=&v"(t) : "v"(w), "v"(sm));
        } else if (MODE == 2) {   // one wait state between
            asm volatile("v_and_b32 %1, 0x0f0f0f0f, %2\n s_nop 4\n v_cvt_pk_f32_fp8_e32 %0, %1\n s_nop 0\n"
                         "v_pk_fma_f32 %0, %0, %3, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n"
                         : "=&v"(r), "=&v"(t) : "v"(w), "v"(sm));
        } else if (MODE == 3) {   // and -> cvt next slot (source forwarding), long gap before the use
            asm volatile("v_and_b32 %1, 0x0f0f0f0f, %2\n v_cvt_pk_f32_fp8_e32 %0, %1\n s_nop 4\n"
                         "v_pk_fma_f32 %0, %0, %3, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n"
                         : "=&v"(r), "=&v"(t) : "v"(w), "v"(sm));
        } else if (MODE == 4) {   // cvt -> scalar v_fma_f32 on each half next slot
            float x, y;
            asm volatile("v_and_b32 %2, 0x0f0f0f0f, %3\n s_nop 4\n v_cvt_pk_f32_fp8_e32 v[100:101], %2\n"
                         "v_fma_f32 %0, v100, %4, %5\n v_fma_f32 %1, v101, %4, %5\n"
                         : "=&v"(x), "=&v"(y), "=&v"(t) : "v"(w), "v"(sm.x), "v"(sm.y) : "v100", "v101");
            r = f32x2{x, y};
        } else if (MODE == 5) {   // an MFMA in flight right before the pair
            asm volatile("v_and_b32 %[t], 0x0f0f0f0f, %[w]\n s_nop 4\n"
                         "v_mfma_f32_16x16x32_bf16 %[acc], %[za], %[zb], %[acc]\n"
                         "v_cvt_pk_f32_fp8_e32 %[r], %[t]\n"
                         "v_pk_fma_f32 %[r], %[r], %[sm], %[sm] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n"
                         : [r] "=&v"(r), [t] "=&v"(t), [acc] "+v"(acc) : [sm] "v"(sm), [za] "v"(za), [zb] "v"(zb), [w] "v"(w));
        } else if (MODE == 6) {   // MFMA in flight, one wait state between cvt and use
            asm volatile("v_and_b32 %[t], 0x0f0f0f0f, %[w]\n s_nop 4\n"
                         "v_mfma_f32_16x16x32_bf16 %[acc], %[za], %[zb], %[acc]\n"
                         "v_cvt_pk_f32_fp8_e32 %[r], %[t]\n s_nop 0\n"
                         "v_pk_fma_f32 %[r], %[r], %[sm], %[sm] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n"
                         : [r] "=&v"(r), [t] "=&v"(t), [acc] "+v"(acc) : [sm] "v"(sm), [za] "v"(za), [zb] "v"(zb), [w] "v"(w));
        } else if (MODE == 7) {   // SDWA form (upper two bytes) -> pk_fma next slot
            asm volatile("v_and_b32 %1, 0x0f0f0f0f, %2\n s_nop 4\n v_cvt_pk_f32_fp8_sdwa %0, %1 src0_sel:WORD_1\n"
                         "v_pk_fma_f32 %0, %0, %3, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n"
                         : "=&v"(r), "=&v"(t) : "v"(w), "v"(sm));
        } else if (MODE >= 9 && MODE <= 32) {
            const f32x2 s512 = {sm.x, sm.x}, m8 = {sm.y, sm.y};
#define PRE "v_and_b32 %[t], 0x0f0f0f0f, %[w]\n s_nop 4\n v_mfma_f32_16x16x32_bf16 %[acc], %[za], %[zb], %[acc]\n"
#define OPS : [r] "=&v"(r), [t] "=&v"(t), [acc] "+v"(acc) : [sm] "v"(sm), [s512] "v"(s512), [m8] "v"(m8), [za] "v"(za), [zb] "v"(zb), [w] "v"(w)
#define FMA_SEL "v_pk_fma_f32 %[r], %[r], %[sm], %[sm] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n"
            if (MODE == 9) asm volatile(PRE "v_cvt_pk_f32_fp8_e32 %[r], %[t]\n v_pk_fma_f32 %[r], %[r], %[s512], %[m8]\n" OPS);
            if (MODE == 10) asm volatile(PRE "v_cvt_pk_f32_fp8_e32 %[r], %[t]\n v_pk_mul_f32 %[r], %[r], %[sm] op_sel_hi:[1,0]\n v_pk_add_f32 %[r], %[r], %[m8]\n" OPS);
            if (MODE == 11) asm volatile(PRE "s_nop 3\n v_cvt_pk_f32_fp8_e32 %[r], %[t]\n" FMA_SEL OPS);
            if (MODE == 12) asm volatile(PRE "s_nop 7\n v_cvt_pk_f32_fp8_e32 %[r], %[t]\n" FMA_SEL OPS);
            if (MODE == 13) asm volatile(PRE "s_nop 15\n v_cvt_pk_f32_fp8_e32 %[r], %[t]\n" FMA_SEL OPS);
            if (MODE == 14) asm volatile(PRE "v_cvt_pk_f32_fp8_e32 %[r], %[t]\n s_nop 7\n" FMA_SEL OPS);
            if (MODE == 15) asm volatile(PRE "v_cvt_pk_f32_fp8_e32 %[r], %[t]\n s_nop 15\n" FMA_SEL OPS);
            {
                // which ingredient of `v_pk_fma_f32 r, r, sm, sm op_sel:[0,0,1] op_sel_hi:[1,0,1]` breaks?
                f32x2 sm2 = sm;            // same values, different registers
                asm volatile("" : "+v"(sm2));
                const f32x2 ms = {sm.y, sm.x};   // {m8, s512}
#define OPS2 : [r] "=&v"(r), [t] "=&v"(t), [acc] "+v"(acc) : [sm] "v"(sm), [sm2] "v"(sm2), [ms] "v"(ms), [s512] "v"(s512), [m8] "v"(m8), [za] "v"(za), [zb] "v"(zb), [w] "v"(w)
#define CV "v_cvt_pk_f32_fp8_e32 %[r], %[t]\n"
                if (MODE == 17) asm volatile(PRE CV "v_pk_fma_f32 %[r], %[r], %[sm], %[sm2] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n" OPS2);   // distinct regs, same selects
                if (MODE == 18) asm volatile(PRE CV "v_pk_fma_f32 %[r], %[r], %[s512], %[sm] op_sel:[0,0,1] op_sel_hi:[1,1,1]\n" OPS2);   // src2 hi-broadcast only
                if (MODE == 19) asm volatile(PRE CV "v_pk_fma_f32 %[r], %[r], %[sm], %[m8] op_sel_hi:[1,0,1]\n" OPS2);                    // src1 lo-broadcast only
                if (MODE == 20) asm volatile(PRE CV "v_pk_fma_f32 %[r], %[r], %[sm], %[ms] op_sel_hi:[1,0,0]\n" OPS2);                    // both lo-broadcast, distinct regs
                if (MODE == 21) asm volatile(PRE CV "v_pk_fma_f32 %[r], %[r], %[ms], %[ms] op_sel:[0,1,0] op_sel_hi:[1,1,0]\n" OPS2);     // same reg: src1 hi-bcast, src2 lo-bcast
                if (MODE == 22) asm volatile(PRE CV "v_pk_mul_f32 %[r], %[r], %[sm] op_sel_hi:[1,0]\n v_pk_add_f32 %[r], %[r], %[sm] op_sel:[0,1] op_sel_hi:[1,1]\n" OPS2);
                if (MODE == 23) asm volatile(PRE CV "v_pk_fma_f32 %[r], %[sm], %[r], %[sm] op_sel:[0,0,1] op_sel_hi:[0,1,1]\n" OPS2);     // same reg as src0 and src2
                if (MODE == 24) asm volatile(PRE CV "v_pk_fma_f32 %[r], %[r], %[s512], %[s512] op_sel_hi:[1,1,1]\n" OPS2);                // same reg, no selects: c*s512 + s512
                // how far apart must the two swizzles of one pair be?  (mode 22 = adjacent = wrong)
                unsigned d0 = w, d1 = w;
#define OPS3 : [r] "=&v"(r), [t] "=&v"(t), [acc] "+v"(acc), [d0] "+v"(d0), [d1] "+v"(d1) : [sm] "v"(sm), [s512] "v"(s512), [za] "v"(za), [zb] "v"(zb), [w] "v"(w)
#define MUL_LO "v_pk_mul_f32 %[r], %[r], %[sm] op_sel_hi:[1,0]\n"
#define ADD_HI "v_pk_add_f32 %[r], %[r], %[sm] op_sel:[0,1] op_sel_hi:[1,1]\n"
#define U1 "v_add_u32 %[d0], %[d0], %[d1]\n"
                if (MODE == 25) asm volatile(PRE CV MUL_LO U1 ADD_HI OPS3);
                if (MODE == 26) asm volatile(PRE CV MUL_LO U1 U1 ADD_HI OPS3);
                if (MODE == 27) asm volatile(PRE CV MUL_LO U1 U1 U1 U1 U1 ADD_HI OPS3);
                if (MODE == 29) asm volatile(PRE CV MUL_LO "s_nop 7\n" ADD_HI OPS3);
                if (MODE == 31) asm volatile(PRE CV ADD_HI OPS3);                                   // hi-broadcast src1 alone: c + m8
                if (MODE == 32) asm volatile(PRE CV "v_pk_mul_f32 %[r], %[r], %[sm] op_sel:[0,1] op_sel_hi:[1,1]\n" OPS3);   // c * m8
                if (d0 == 0x12345 && d1 == 0x777) out[1] = f32x2{1.f, 2.f};
            }
            if (MODE == 16) {   // conversion finished long before the MFMA: only pk_fma follows it
                asm volatile("v_and_b32 %[t], 0x0f0f0f0f, %[w]\n v_cvt_pk_f32_fp8_e32 %[r], %[t]\n s_nop 7\n"
                             "v_mfma_f32_16x16x32_bf16 %[acc], %[za], %[zb], %[acc]\n" FMA_SEL OPS);
            }
        } else if (MODE == 8) {   // pk_fma -> v_cvt_pk_bf16_f32 next slot (result returned as bits)
            unsigned b;
            asm volatile("v_and_b32 %1, 0x0f0f0f0f, %2\n s_nop 4\n v_cvt_pk_f32_fp8_e32 v[100:101], %1\n s_nop 4\n"
                         "v_pk_fma_f32 v[100:101], v[100:101], %3, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n"
                         "v_cvt_pk_bf16_f32 %0, v100, v101\n"
                         : "=&v"(b), "=&v"(t) : "v"(w), "v"(sm) : "v100", "v101");
            r = f32x2{__builtin_bit_cast(float, b << 16), __builtin_bit_cast(float, b & 0xffff0000u)};
        }
        out[(size_t)it * gridDim.x * blockDim.x + i] = r;
        w = (w >> 4) | (w << 28);
    }
    if (acc.x == 123.f) out[0] = f32x2{acc.x, acc.y};
}
static float fp8_nib(unsigned b) { return ldexpf((float)(b & 15), -9); }
static float bf16r(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u); u &= 0xffff0000u;
    memcpy(&f, &u, 4); return f;
}
template <int MODE>
static void run(const char* name, const unsigned* dw, const float* ds, f32x2* dout, const std::vector<unsigned>& w,
                const std::vector<float>& s, int n, int iters) {
    (void)hipMemset(dout, 0xff, sizeof(f32x2) * (size_t)n * iters);
    hipLaunchKernelGGL(k<MODE>, dim3(n / 256), dim3(256), 0, 0, dw, ds, dout, iters);
    hipError_t e = hipDeviceSynchronize();
    std::vector<f32x2> out((size_t)n * iters);
    (void)hipMemcpy(out.data(), dout, sizeof(f32x2) * out.size(), hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (int i = 0; i < n; ++i) {
        unsigned ww = w[i];
        for (int it = 0; it < iters; ++it) {
            const unsigned m = ww & 0x0f0f0f0fu;
            const unsigned b0 = MODE == 7 ? (m >> 16) & 0xff : m & 0xff, b1 = MODE == 7 ? m >> 24 : (m >> 8) & 0xff;
            float x = fmaf(fp8_nib(b0), 512.0f * s[i], -8.0f * s[i]), y = fmaf(fp8_nib(b1), 512.0f * s[i], -8.0f * s[i]);
            if (MODE == 8) { x = bf16r(x); y = bf16r(y); }
            if (MODE == 31) { x = fp8_nib(b0) + -8.0f * s[i]; y = fp8_nib(b1) + -8.0f * s[i]; }
            if (MODE == 32) { x = fp8_nib(b0) * (-8.0f * s[i]); y = fp8_nib(b1) * (-8.0f * s[i]); }
            if (MODE == 24) { x = fmaf(fp8_nib(b0), 512.0f * s[i], 512.0f * s[i]); y = fmaf(fp8_nib(b1), 512.0f * s[i], 512.0f * s[i]); }
            const f32x2 g = out[(size_t)it * n + i];
            const float gx = g.x, gy = g.y;
            if (memcmp(&gx, &x, 4) || memcmp(&gy, &y, 4)) {
                if (bad < 4) printf("   %s i=%d (lane %d) it=%d got (%g, %g) want (%g, %g) | v=(%u,%u) s=%g 512s=%g -8s=%g\n", name, i, i & 63, it, gx, gy, x, y, b0 & 15, b1 & 15, s[i], 512.f * s[i], -8.f * s[i]);
                ++bad;
            }
            ww = (ww >> 4) | (ww << 28);
        }
    }
    printf("%-52s %s: %zu / %zu wrong\n", name, e == hipSuccess ? "ok" : hipGetErrorString(e), bad, out.size());
}
int main() {
    const int n = 256 * 2048, iters = 16;
    std::vector<unsigned> w(n);
    std::vector<float> s(n);
    uint64_t st = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        w[i] = (unsigned)st;
        s[i] = ldexpf(1.0f + (float)((st >> 40) & 127) / 128.0f, -6 + (int)((st >> 50) & 7));
    }
    unsigned* dw; float* ds; f32x2* dout;
    (void)hipMalloc(&dw, n * 4); (void)hipMalloc(&ds, n * 4); (void)hipMalloc(&dout, sizeof(f32x2) * (size_t)n * iters);
    (void)hipMemcpy(dw, w.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(ds, s.data(), n * 4, hipMemcpyHostToDevice);
    run<0>("0 compiler-scheduled C", dw, ds, dout, w, s, n, iters);
    run<1>("1 cvt_pk_f32_fp8 -> pk_fma next slot", dw, ds, dout, w, s, n, iters);
    run<2>("2 cvt -> s_nop 0 -> pk_fma", dw, ds, dout, w, s, n, iters);
    run<3>("3 and -> cvt next slot, long gap before use", dw, ds, dout, w, s, n, iters);
    run<4>("4 cvt -> v_fma_f32 (each half) next slot", dw, ds, dout, w, s, n, iters);
    run<5>("5 mfma; cvt -> pk_fma next slot", dw, ds, dout, w, s, n, iters);
    run<6>("6 mfma; cvt -> s_nop 0 -> pk_fma", dw, ds, dout, w, s, n, iters);
    run<7>("7 cvt sdwa WORD_1 -> pk_fma next slot", dw, ds, dout, w, s, n, iters);
    run<8>("8 pk_fma -> cvt_pk_bf16 next slot", dw, ds, dout, w, s, n, iters);
    run<9>("9 mfma; cvt; pk_fma plain pairs (no op_sel)", dw, ds, dout, w, s, n, iters);
    run<10>("10 mfma; cvt; pk_mul; pk_add", dw, ds, dout, w, s, n, iters);
    run<11>("11 mfma; s_nop 3; cvt; pk_fma(sel)", dw, ds, dout, w, s, n, iters);
    run<12>("12 mfma; s_nop 7; cvt; pk_fma(sel)", dw, ds, dout, w, s, n, iters);
    run<13>("13 mfma; s_nop 15; cvt; pk_fma(sel)", dw, ds, dout, w, s, n, iters);
    run<14>("14 mfma; cvt; s_nop 7; pk_fma(sel)", dw, ds, dout, w, s, n, iters);
    run<15>("15 mfma; cvt; s_nop 15; pk_fma(sel)", dw, ds, dout, w, s, n, iters);
    run<16>("16 cvt; s_nop 7; mfma; pk_fma(sel)", dw, ds, dout, w, s, n, iters);
    run<17>("17 distinct regs, selects [0,0,1]/[1,0,1]", dw, ds, dout, w, s, n, iters);
    run<18>("18 src1 plain, src2 = sm hi-broadcast", dw, ds, dout, w, s, n, iters);
    run<19>("19 src1 = sm lo-broadcast, src2 plain", dw, ds, dout, w, s, n, iters);
    run<20>("20 both lo-broadcast, distinct regs", dw, ds, dout, w, s, n, iters);
    run<21>("21 same reg {m8,s512}: src1 hi-, src2 lo-broadcast", dw, ds, dout, w, s, n, iters);
    run<22>("22 pk_mul(sm lo) ; pk_add(sm hi-broadcast)", dw, ds, dout, w, s, n, iters);
    run<23>("23 same reg as src0 and src2", dw, ds, dout, w, s, n, iters);
    run<24>("24 same reg src1 = src2, no selects", dw, ds, dout, w, s, n, iters);
    run<25>("25 pk_mul(sm lo); 1 VALU; pk_add(sm hi)", dw, ds, dout, w, s, n, iters);
    run<26>("26 pk_mul(sm lo); 2 VALU; pk_add(sm hi)", dw, ds, dout, w, s, n, iters);
    run<27>("27 pk_mul(sm lo); 5 VALU; pk_add(sm hi)", dw, ds, dout, w, s, n, iters);
    run<29>("29 pk_mul(sm lo); s_nop 7; pk_add(sm hi)", dw, ds, dout, w, s, n, iters);
    run<31>("31 pk_add src1 hi-broadcast alone", dw, ds, dout, w, s, n, iters);
    run<32>("32 pk_mul src1 hi-broadcast alone", dw, ds, dout, w, s, n, iters);
    return 0;
}
