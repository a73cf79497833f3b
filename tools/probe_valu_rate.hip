// probe_valu_rate.hip -- stand-alone MI355X probe: the ISSUE cost of the vector instructions the 4-bit decoders are made of
// (gemm_skinny.h: Dec<LKM_W_INT4_B8>::frag_m and its candidates), one instruction kind at a time, 16 independent chains per
// wave, 1 / 2 / 4 waves per SIMD.  Printed: ns per wave-instruction per SIMD and the ratio to v_and_b32 (a full-rate
// instruction: 4 cycles per wave on a 16-lane SIMD).  DESIGN.md 4 "4-bit decode, round 6" measured "~5 cycles per vector
// instruction whatever the wave count" on the whole decoder; this says which of its instructions are not full rate.
// build: hipcc -O2 --offload-arch=gfx950 tools/probe_valu_rate.hip -o tools/_bin/probe_valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x)                                                     \
    do {                                                          \
        hipError_t e_ = (x);                                      \
        if (e_ != hipSuccess) {                                   \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_)); \
            return 1;                                             \
        }                                                         \
    } while (0)

constexpr int CHAINS = 16, REPS = 8, ITERS = 2000;   // instructions per wave = CHAINS x REPS x ITERS

// one instruction on chain registers: d (dword or pair, written), a / b (read; never written: no dependency between chains'
// instructions except through d of the same chain REPS instructions later)
#define K1(NAME, TXT)                                                                                  \
    __global__ __launch_bounds__(1024) void NAME(unsigned* out, unsigned seed) {                        \
        unsigned d[CHAINS], a = seed + threadIdx.x, b = seed * 3 + 1, c = seed ^ 0x3c003c00u;           \
        unsigned long long q[CHAINS];                                                                   \
        const unsigned long long aa = ((unsigned long long)b << 32) | a, cc = ((unsigned long long)c << 32) | c; \
        for (int i = 0; i < CHAINS; ++i) { d[i] = seed + i; q[i] = seed * 7ull + i; }                   \
        for (int it = 0; it < ITERS; ++it) {                                                            \
            _Pragma("unroll") for (int r = 0; r < REPS; ++r)                                            \
                _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) {                                    \
                    asm volatile(TXT : "+v"(d[i]), "+v"(q[i]) : "v"(aa), "v"(cc), "v"(a), "v"(b));      \
                }                                                                                       \
        }                                                                                               \
        unsigned s = 0;                                                                                 \
        for (int i = 0; i < CHAINS; ++i) s ^= d[i] ^ (unsigned)q[i] ^ (unsigned)(q[i] >> 32);                                        \
        if (s == 0x12345678u) out[threadIdx.x] = s;                                                     \
    }

// %0 = d (dword), %1 = q (64-bit pair), %2 = {a, b} pair, %3 = {c, c} pair, %4 = a, %5 = b
K1(k_and, "v_and_b32 %0, %4, %0")
K1(k_lshr, "v_lshrrev_b32 %0, 4, %0")
K1(k_and_or, "v_and_or_b32 %0, %0, %4, %5")
K1(k_perm, "v_perm_b32 %0, %0, %4, %5")
K1(k_bfi, "v_bfi_b32 %0, %4, %0, %5")
K1(k_cvt_f32_fp8, "v_cvt_pk_f32_fp8 %1, %4")
K1(k_cvt_f32_fp8_hi, "v_cvt_pk_f32_fp8_sdwa %1, %4 src0_sel:WORD_1")
K1(k_cvt_f32_ub, "v_cvt_f32_ubyte1 %0, %4")
K1(k_pk_fma, "v_pk_fma_f32 %1, %2, %3, %1")
K1(k_pk_mul, "v_pk_mul_f32 %1, %2, %1")
K1(k_fma, "v_fma_f32 %0, %4, %5, %0")
K1(k_cvt_bf16, "v_cvt_pk_bf16_f32 %0, %0, %4")
K1(k_cvt_f16, "v_cvt_pkrtz_f16_f32 %0, %0, %4")
K1(k_pk_mul_f16, "v_pk_mul_f16 %0, %0, %4")
K1(k_pk_fma_f16, "v_pk_fma_f16 %0, %4, %5, %0")
K1(k_pk_add_f16, "v_pk_add_f16 %0, %0, %4")
K1(k_scale_bf16_fp8, "v_cvt_scalef32_pk_bf16_fp8 %0, %4, %5")
K1(k_scale_f32_fp8, "v_cvt_scalef32_pk_f32_fp8 %1, %4, %5")
K1(k_scale_bf16_fp4, "v_cvt_scalef32_pk_bf16_fp4 %0, %4, %5")
K1(k_exp, "v_exp_f32 %0, %0")
K1(k_dot2_bf16, "v_dot2c_f32_bf16 %0, %4, %5")
K1(k_mov, "v_mov_b32 %0, %4")
K1(k_pk_mov, "v_pk_mov_b32 %1, %2, %3")
K1(k_lshl_or, "v_lshl_or_b32 %0, %0, 4, %4")
K1(k_bfe, "v_bfe_u32 %0, %4, 4, 4")
K1(k_mul_u24, "v_mul_u32_u24 %0, %0, %4")
K1(k_mad_u24, "v_mad_u32_u24 %0, %4, %5, %0")
K1(k_pk_add_u16, "v_pk_add_u16 %0, %0, %4")
K1(k_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %4")
K1(k_pk_mad_u16, "v_pk_mad_u16 %0, %4, %5, %0")
K1(k_pk_lshr_b16, "v_pk_lshrrev_b16 %0, 4, %0")

struct Entry {
    const char* name;
    void (*k)(unsigned*, unsigned);
};

int main() {
    unsigned* out;
    CK(hipMalloc(&out, 4096 * 4));
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", pr.gcnArchName, cus, pr.clockRate);
    std::vector<Entry> ks = {
        {"v_and_b32", k_and}, {"v_lshrrev_b32", k_lshr}, {"v_and_or_b32", k_and_or}, {"v_perm_b32", k_perm}, {"v_bfi_b32", k_bfi},
        {"v_lshl_or_b32", k_lshl_or}, {"v_bfe_u32", k_bfe}, {"v_mov_b32", k_mov}, {"v_pk_mov_b32", k_pk_mov},
        {"v_cvt_pk_f32_fp8 (word 0)", k_cvt_f32_fp8}, {"v_cvt_pk_f32_fp8 (word 1)", k_cvt_f32_fp8_hi}, {"v_cvt_f32_ubyte1", k_cvt_f32_ub},
        {"v_pk_fma_f32", k_pk_fma}, {"v_pk_mul_f32", k_pk_mul}, {"v_fma_f32", k_fma},
        {"v_cvt_pk_bf16_f32", k_cvt_bf16}, {"v_cvt_pkrtz_f16_f32", k_cvt_f16},
        {"v_pk_mul_f16", k_pk_mul_f16}, {"v_pk_fma_f16", k_pk_fma_f16}, {"v_pk_add_f16", k_pk_add_f16},
        {"v_cvt_scalef32_pk_bf16_fp8", k_scale_bf16_fp8}, {"v_cvt_scalef32_pk_f32_fp8", k_scale_f32_fp8},
        {"v_cvt_scalef32_pk_bf16_fp4", k_scale_bf16_fp4}, {"v_exp_f32", k_exp}, {"v_dot2c_f32_bf16", k_dot2_bf16},
        {"v_mul_u32_u24", k_mul_u24}, {"v_mad_u32_u24", k_mad_u24}, {"v_pk_add_u16", k_pk_add_u16},
        {"v_pk_mul_lo_u16", k_pk_mul_lo_u16}, {"v_pk_mad_u16", k_pk_mad_u16}, {"v_pk_lshrrev_b16", k_pk_lshr_b16},
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double n_inst = (double)CHAINS * REPS * ITERS;
    double base[3] = {0, 0, 0};
    printf("%-34s %10s %10s %10s   (ns per wave-instruction per SIMD; x = ratio to v_and_b32)\n", "instruction", "1 wave", "2 waves", "4 waves");
    for (auto& en : ks) {
        double ns[3];
        for (int wi = 0; wi < 3; ++wi) {
            const int wps = 1 << wi, threads = 256 * wps;
            en.k<<<cus, threads>>>(out, 1u);                  // warm-up (and clock ramp)
            CK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                en.k<<<cus, threads>>>(out, 1u + rep);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            ns[wi] = best * 1e6 / (n_inst * wps);
            if (en.k == k_and) base[wi] = ns[wi];
        }
        printf("%-34s %7.3f %4.2fx %7.3f %4.2fx %7.3f %4.2fx\n", en.name, ns[0], ns[0] / base[0], ns[1], ns[1] / base[1], ns[2], ns[2] / base[2]);
    }
    return 0;
}
