// probe_int4_unit.hip -- stand-alone MI355X probe: what a SIMD sustains on ONE K unit (128 k) of the gated uint4b8 decode
// kernel's inner loop -- the exact T((q - 8) * s) decode of two 16-row weight tiles (the product's own Dec<>::frag_m:
// 15 VALU per 8 weights) and their MFMAs against NB token blocks held in registers -- with no memory traffic at all,
// as a function of the token blocks in use and of the waves per SIMD.  The HBM budget it has to fit: 2 KiB of weights per
// unit and wave; at 8 TB/s a SIMD must finish one wave-unit every ~630 cycles (2.1 GHz, 1024 SIMDs).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I lvllm_amd/csrc tools/probe_int4_unit.hip -o /tmp/probe_int4_unit
#include "gemm_skinny.h"

#include <cstdarg>
#include <cstdio>

namespace lkm {
void set_error(const char*, ...) {}
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

// MODE bit 0: decode, bit 1: MFMA
template <int NB, int MODE, int FAST>
__global__ __launch_bounds__(1024) void probe(const u32x4* __restrict__ w, const u32x4* __restrict__ x, float* __restrict__ out,
                                              long long* __restrict__ cycles, int iters) {
    typedef Dec<LKM_W_INT4_B8, LKM_DT_BF16> D;
    typedef Dec<LKM_W_INT4_PS, LKM_DT_BF16> DF;
    const int lane = threadIdx.x & 63;
    u32x4 raw[2][1];
    raw[0][0] = w[threadIdx.x];
    raw[1][0] = w[threadIdx.x + 1024];
    typename D::Aux aux[2];
    aux[0].raw = u32x2{0x3c003c00u + lane, 0x3c003c00u};
    aux[1].raw = u32x2{0x3c803c80u + lane, 0x3c803c80u};
    u32x4 bf[4][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int b = 0; b < 4; ++b) bf[ks][b] = x[(ks * 4 + b) * 64 + lane];
    f32x4 acc[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[t][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        // (a new unit's weights: perturbed so that nothing is hoisted)
        raw[0][0] += u32x4{1u, 3u, 5u, 7u};
        raw[1][0] += u32x4{2u, 4u, 6u, 8u};
        typename D::Mult mu[2];
        if constexpr (!FAST && (MODE & 1)) {
#pragma unroll
            for (int t = 0; t < 2; ++t) mu[t] = D::mult(aux[t], 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            u32x4 a[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if constexpr (!(MODE & 1)) a[t] = raw[t][0] + u32x4{(unsigned)ks, 0u, 0u, 0u};
                else if constexpr (FAST) a[t] = DF::frag(raw[t], typename DF::Aux{}, ks, 0);
                else a[t] = D::frag_m(raw[t], ks, mu[t]);
            }
            if constexpr (MODE & 2) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int b = 0; b < NB; ++b) acc[t][b] = ActT<LKM_DT_BF16>::mfma(a[t], bf[ks][b], acc[t][b]);
            } else {
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t][0] += f32x4{__builtin_bit_cast(float, a[t].x), __builtin_bit_cast(float, a[t].y),
                                                               __builtin_bit_cast(float, a[t].z), __builtin_bit_cast(float, a[t].w)};
            }
        }
    }
    const long long t1 = clock64();
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int b = 0; b < 4; ++b) s += acc[t][b];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z + s.w;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int NB, int MODE, int FAST>
static int run(const char* what, int waves_per_simd, const u32x4* w, const u32x4* x, float* out, long long* cyc) {
    const int iters = 4000, threads = 256 * waves_per_simd, blocks = 256;
    hipLaunchKernelGGL((probe<NB, MODE, FAST>), dim3(blocks), dim3(threads), 0, 0, w, x, out, cyc, iters);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<NB, MODE, FAST>), dim3(blocks), dim3(threads), 0, 0, w, x, out, cyc, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    long long c[256];
    CK(hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost));
    double mean = 0;
    for (int i = 0; i < blocks; ++i) mean += (double)c[i] / blocks;
    // one unit of one wave = 2 KiB of weights; all waves of a SIMD share it
    const double ns_unit_simd = ms * 1e6 / iters / waves_per_simd;
    const double tbps = 2048.0 / ns_unit_simd * 1024 / 1e3;        // bytes per ns per SIMD x 1024 SIMDs -> TB/s
    printf("%-58s %d waves/SIMD  %7.1f ticks/unit/wave  %6.1f ns per wave-unit per SIMD  -> weight stream it could feed: %5.2f TB/s\n", what,
           waves_per_simd, mean / iters, ns_unit_simd, tbps);
    return 0;
}

int main() {
    u32x4 *w, *x;
    float* out;
    long long* cyc;
    CK(hipMalloc(&w, 2048 * 16));
    CK(hipMalloc(&x, 16 * 64 * 16));
    CK(hipMalloc(&out, 256 * 1024 * 4));
    CK(hipMalloc(&cyc, 256 * 8));
    CK(hipMemset(w, 0x57, 2048 * 16));
    CK(hipMemset(x, 0x3c, 16 * 64 * 16));
    for (int wps = 1; wps <= 4; ++wps) {
        if (wps == 4) printf("(four waves per SIMD need <= 128 registers: the compiler decides whether this build still fits)\n");
        run<2, 1, 0>("exact decode only", wps, w, x, out, cyc);
        run<2, 2, 0>("MFMA only, 2 token blocks", wps, w, x, out, cyc);
        run<4, 2, 0>("MFMA only, 4 token blocks", wps, w, x, out, cyc);
        run<2, 3, 0>("exact decode + MFMA, 2 token blocks", wps, w, x, out, cyc);
        run<3, 3, 0>("exact decode + MFMA, 3 token blocks", wps, w, x, out, cyc);
        run<4, 3, 0>("exact decode + MFMA, 4 token blocks", wps, w, x, out, cyc);
        run<2, 1, 1>("fast-mode decode only", wps, w, x, out, cyc);
        run<3, 3, 1>("fast-mode decode + MFMA, 3 token blocks", wps, w, x, out, cyc);
    }
    return 0;
}
