// probe_w4_unit32.hip -- stand-alone MI355X probe (round 6): what a SIMD sustains on ONE K unit (128 k) of the uint4b8 decode
// kernels' inner loop in its 32 x 32 x 16 MFMA form (gemm_w4x.h / gemm_w4e.h / gemm_w4s.h: a lane decodes one weight row, 8
// fragments of 8 weights per unit, CB token column blocks) with no memory traffic at all, as a function of the waves per SIMD,
// of the decoder (packed-fp32 / plain) and of the column blocks.  2 KiB of weights per wave-unit: 229 k wave-units per GEMM1
// of Mixtral int4 (configs[2]) on 1024 SIMDs = 224 per SIMD.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I lvllm_amd/csrc -I include tools/probe_w4_unit32.hip -o /tmp/probe_w4_unit32
#include "gemm_w4x.h"

#include <cstdarg>
#include <cstdio>

namespace lkm {
void set_error(const char*, ...) {}
void note_gemm_launch(const void*) {}
}  // namespace lkm
using namespace lkm;

#define CK(x)                                                     \
    do {                                                          \
        hipError_t e_ = (x);                                      \
        if (e_ != hipSuccess) {                                   \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_)); \
            return 1;                                             \
        }                                                         \
    } while (0)

// accumulators in the AGPR half of the register file (round 6, second probe: does the C / D traffic of a 32 x 32 MFMA -- 16
// registers read and written per instruction -- stop competing with the decoder's VALU operand reads when it is there?)
__device__ __forceinline__ void mfma_agpr(f32x16& acc, u32x4 a, u32x4 b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

// MODE bit 0: decode, bit 1: MFMA, bit 2: accumulators in AGPRs, bit 3: the column blocks' MFMAs spread between the next
// fragment's decode (sched_group_barrier) instead of back to back; DECV: 0 packed-fp32 decoder, 1 plain
template <int CB, int MODE, int DECV, int THREADS>
__global__ __launch_bounds__(THREADS) void probe(const u32x4* __restrict__ w, const u32x4* __restrict__ x, float* __restrict__ out, int iters) {
    typedef Dec<LKM_W_INT4_B8, LKM_DT_BF16> D;
    typedef W4Int4<LKM_W_INT4_B8, LKM_DT_BF16> W;
    const int lane = threadIdx.x & 63;
    u32x4 raw[2][1];
    raw[0][0] = w[threadIdx.x];
    raw[1][0] = w[threadIdx.x + 1024];
    typename D::Aux aux;
    aux.raw = u32x2{0x3c00u + lane, 0u};
    u32x4 bf[4][2][CB];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int c = 0; c < CB; ++c) bf[ks][q][c] = x[((ks * 2 + q) * 2 + c) * 64 + lane];
    f32x16 acc[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
        raw[0][0] += u32x4{1u, 3u, 5u, 7u};      // (a new unit's weights: perturbed so that nothing is hoisted)
        raw[1][0] += u32x4{2u, 4u, 6u, 8u};
        aux.raw.x += 1u;
        typename W::M mu;
        if constexpr (MODE & 1) mu = W::mult(aux);
        auto dec = [&](int s_, int q_) __attribute__((always_inline)) {
            if constexpr (!(MODE & 1)) return raw[q_][0] + u32x4{(unsigned)s_, 0u, 0u, 0u};
            else return W::template frag<DECV>(raw[q_], s_, mu);
        };
        u32x4 a = dec(0, 0);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int s_ = t >> 1, q_ = t & 1;
            u32x4 an = a;
            if (t + 1 < 8) an = dec((t + 1) >> 1, (t + 1) & 1);
            if constexpr ((MODE & 6) == 6) {
#pragma unroll
                for (int c = 0; c < CB; ++c) mfma_agpr(acc[c], a, bf[s_][q_][c]);
            } else if constexpr (MODE & 2) {
#pragma unroll
                for (int c = 0; c < CB; ++c) acc[c] = Mfma32<LKM_DT_BF16>::run(a, bf[s_][q_][c], acc[c]);
                if constexpr ((MODE & 8) != 0 && CB == 2) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                }
            } else {
#pragma unroll
                for (int c = 0; c < CB; ++c) acc[c][0] += __builtin_bit_cast(float, a.x ^ a.y ^ a.z ^ a.w);
            }
            a = an;
        }
    }
    if constexpr ((MODE & 6) == 6) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[c][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CB, int MODE, int DECV, int WPS>
static int run(const char* what, const u32x4* w, const u32x4* x, float* out) {
    constexpr int threads = 256 * WPS;
    const int iters = 4000, blocks = 256;
    hipLaunchKernelGGL((probe<CB, MODE, DECV, threads>), dim3(blocks), dim3(threads), 0, 0, w, x, out, iters);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<CB, MODE, DECV, threads>), dim3(blocks), dim3(threads), 0, 0, w, x, out, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double ns_unit_simd = ms * 1e6 / iters / WPS;          // one wave-unit = 2 KiB of weights; the waves of a SIMD share it
    printf("%-46s CB=%d %d waves/SIMD  %6.1f ns per wave-unit per SIMD -> GEMM1 of configs[2] (224 wave-units per SIMD): %6.1f us\n", what, CB, WPS,
           ns_unit_simd, ns_unit_simd * 224 / 1e3);
    return 0;
}

template <int WPS>
static int sweep(const u32x4* w, const u32x4* x, float* out) {
    run<1, 1, 0, WPS>("decode only (packed fp32)", w, x, out);
    run<1, 1, 1, WPS>("decode only (plain)", w, x, out);
    run<1, 2, 0, WPS>("MFMA only", w, x, out);
    run<2, 2, 0, WPS>("MFMA only", w, x, out);
    run<1, 3, 0, WPS>("decode (packed fp32) + MFMA", w, x, out);
    run<2, 3, 0, WPS>("decode (packed fp32) + MFMA", w, x, out);
    run<1, 3, 1, WPS>("decode (plain) + MFMA", w, x, out);
    run<2, 3, 1, WPS>("decode (plain) + MFMA", w, x, out);
    run<1, 7, 0, WPS>("decode (packed fp32) + MFMA, AGPR acc", w, x, out);
    run<2, 7, 0, WPS>("decode (packed fp32) + MFMA, AGPR acc", w, x, out);
    run<2, 6, 0, WPS>("MFMA only, AGPR acc", w, x, out);
    run<2, 11, 0, WPS>("decode (packed fp32) + MFMA, spread", w, x, out);
    return 0;
}

int main() {
    u32x4 *w, *x;
    float* out;
    CK(hipMalloc(&w, 2048 * 16));
    CK(hipMalloc(&x, 32 * 64 * 16));
    CK(hipMalloc(&out, 256 * 1024 * 4));
    CK(hipMemset(w, 0x57, 2048 * 16));
    CK(hipMemset(x, 0x3c, 32 * 64 * 16));
    sweep<1>(w, x, out);
    sweep<2>(w, x, out);
    sweep<3>(w, x, out);
    sweep<4>(w, x, out);
    return 0;
}
