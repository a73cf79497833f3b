// probe_mfma_issue.hip -- stand-alone MI355X probe: what a SIMD sustains on the inner block of the round-3 fp8 prefill
// kernel (gemm_prefill_a8w.h): two v_mfma_scale_f32_16x16x128_f8f6f4 + ten fp32 VALU (+ three LDS reads) per 16-token
// block, two waves per SIMD, fixed register map -- as a function of WHERE the operands sit in the register file
// (VGPR banks = index mod 4) and of what shares the block with the MFMAs.
// build: hipcc -O2 --offload-arch=gfx950 tools/probe_mfma_issue.hip -o /tmp/probe_mfma_issue
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x)                                                                        \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                    \
            return 1;                                                                \
        }                                                                            \
    } while (0)

struct Cfg {
    int a, b0, b1, p, acc, f, one;   // register bases: A (16), B parities (8 each), P (16), acc (128), f (4), scale
    int mfma, valu, ds, fma3;        // what a block contains; fma3: v_fma_f32 (VOP3) instead of v_fmac_f32 (VOP2)
};

#define MFMA(P, A, B) \
    "v_mfma_scale_f32_16x16x128_f8f6f4 v[" P ":" P "+3], v[" A ":" A "+7], v[" B ":" B "+7], 0, v[%c[one]], v[%c[one]] op_sel_hi:[0,0,0]\n\t"
#define FMAC4(ACC, F, P)                                 \
    ".if %c[fma3] == 2\n\t"                              \
    "v_pk_fma_f32 v[" ACC "+0:" ACC "+1], v[" P "+0:" P "+1], v[" F ":" F "+1], v[" ACC "+0:" ACC "+1] op_sel_hi:[1,0,1]\n\t" \
    "v_pk_fma_f32 v[" ACC "+2:" ACC "+3], v[" P "+2:" P "+3], v[" F ":" F "+1], v[" ACC "+2:" ACC "+3] op_sel_hi:[1,0,1]\n\t" \
    ".elseif %c[fma3]\n\t"                               \
    "v_fma_f32 v[" ACC "+0], v[" F "], v[" P "+0], v[" ACC "+0]\n\t" \
    "v_fma_f32 v[" ACC "+1], v[" F "], v[" P "+1], v[" ACC "+1]\n\t" \
    "v_fma_f32 v[" ACC "+2], v[" F "], v[" P "+2], v[" ACC "+2]\n\t" \
    "v_fma_f32 v[" ACC "+3], v[" F "], v[" P "+3], v[" ACC "+3]\n\t" \
    ".else\n\t"                                          \
    "v_fmac_f32 v[" ACC "+0], v[" F "], v[" P "+0]\n\t"  \
    "v_fmac_f32 v[" ACC "+1], v[" F "], v[" P "+1]\n\t"  \
    "v_fmac_f32 v[" ACC "+2], v[" F "], v[" P "+2]\n\t"  \
    "v_fmac_f32 v[" ACC "+3], v[" F "], v[" P "+3]\n\t"  \
    ".endif\n\t"

template <int A, int B0, int B1, int P, int ACC, int F, int ONE, int DO_MFMA, int DO_VALU, int DO_DS, int FMA3, int BLK>
__device__ __forceinline__ void block(int vaddr, int ws) {
    constexpr int par = BLK & 1;
    constexpr int BC = par ? B1 : B0, BN = par ? B0 : B1;
    constexpr int PC = P + par * 8, PP = P + (par ^ 1) * 8;
    constexpr int FC = FMA3 == 2 ? F + par * 4 : F + par * 2, FP = FMA3 == 2 ? F + (par ^ 1) * 4 : F + (par ^ 1) * 2;
    constexpr int FS = FMA3 == 2 ? 2 : 1;      // register distance between the two tiles' scale products
    constexpr int AC = ACC + ((BLK + 15) & 15) * 8;
    asm volatile(
        ".if %c[ds]\n\t"
        "ds_read_b128 v[%c[bn]:%c[bn]+3], %[va] offset:%c[boff]\n\t"
        "ds_read_b128 v[%c[bn]+4:%c[bn]+7], %[va] offset:%c[boff]+1024\n\t"
        "ds_read_b32 v[%c[x]], %[va] offset:%c[boff]\n\t"
        ".endif\n\t"
        ".if %c[valu]\n\t"
        "v_mul_f32 v[%c[fc]], %[ws], v[%c[x]]\n\t"
        "v_mul_f32 v[%c[fc]+%c[fs]], %[ws], v[%c[x]]\n\t"
        ".endif\n\t"
        ".if %c[mfma]\n\t" MFMA("%c[pc]", "%c[a]", "%c[bc]") ".endif\n\t"
        ".if %c[valu]\n\t" FMAC4("%c[acc]", "%c[fp]", "%c[pp]") ".endif\n\t"
        ".if %c[mfma]\n\t" MFMA("%c[pc]+4", "%c[a]+8", "%c[bc]") ".endif\n\t"
        ".if %c[valu]\n\t" FMAC4("%c[acc]+4", "%c[fp]+%c[fs]", "%c[pp]+4") ".endif\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        :
        : [bn] "i"(BN), [bc] "i"(BC), [pc] "i"(PC), [pp] "i"(PP), [fc] "i"(FC), [fp] "i"(FP), [a] "i"(A), [acc] "i"(AC),
          [one] "i"(ONE), [x] "i"(ONE + 1), [boff] "i"(BLK * 2048), [mfma] "i"(DO_MFMA), [valu] "i"(DO_VALU), [ds] "i"(DO_DS),
          [fma3] "i"(FMA3), [fs] "i"(FS), [va] "v"(vaddr), [ws] "s"(ws)
        : "memory", "v255");
}

template <int A, int B0, int B1, int P, int ACC, int F, int ONE, int DO_MFMA, int DO_VALU, int DO_DS, int FMA3>
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(32))) void probe(long long* out, int iters, int ws_in) {
    extern __shared__ char lds[];
    const int vaddr = (threadIdx.x & 63) * 16;
    const int ws = __builtin_amdgcn_readfirstlane(ws_in);
    // operands: small finite patterns (random-ish so that the data path toggles)
#pragma unroll
    for (int r = 32; r < 256; r += 8)
        asm volatile("v_mov_b32 v[%c0+0], %1\n\tv_mov_b32 v[%c0+1], %1\n\tv_mov_b32 v[%c0+2], %1\n\tv_mov_b32 v[%c0+3], %1\n\t"
                     "v_mov_b32 v[%c0+4], %1\n\tv_mov_b32 v[%c0+5], %1\n\tv_mov_b32 v[%c0+6], %1\n\tv_mov_b32 v[%c0+7], %1" ::"i"(r),
                     "v"((int)(0x3c383430u + threadIdx.x * 0x01010101u + r)) : "memory");
    asm volatile("v_mov_b32 v[%c0], 0x7f7f7f7f\n\tv_mov_b32 v[%c0+1], 1.0" ::"i"(ONE) : "memory");
    for (int i = threadIdx.x; i < 40960 / 4; i += 512) ((int*)lds)[i] = 0x38383838;
    __syncthreads();
    long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#define BLK(n) block<A, B0, B1, P, ACC, F, ONE, DO_MFMA, DO_VALU, DO_DS, FMA3, n>(vaddr, ws);
        BLK(0) BLK(1) BLK(2) BLK(3) BLK(4) BLK(5) BLK(6) BLK(7) BLK(8) BLK(9) BLK(10) BLK(11) BLK(12) BLK(13) BLK(14) BLK(15)
#undef BLK
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <typename K>
static int run(const char* name, K kern, int threads, int mfma_per_block) {
    const int iters = 2000, nwg = 256;
    long long* d;
    CK(hipMalloc(&d, nwg * 8 * sizeof(long long)));
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(threads), 65536, 0, d, 50, 0x3f800000);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(threads), 65536, 0, d, iters, 0x3f800000);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const int nw = nwg * threads / 64;
    std::vector<long long> h(nw);
    CK(hipMemcpy(h.data(), d, nw * sizeof(long long), hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double ticks = (double)h[nw / 2] / iters / 16.0;          // s_memtime ticks per block, one wave
    const double ns_blk = ms * 1e6 / iters / 16.0;                     // wall ns per block (all waves run concurrently)
    const int waves_per_simd = threads / 256;
    printf("%-44s %7.1f ticks/block/wave  %6.1f ns/block  -> %5.1f ns per MFMA per SIMD (%d waves/SIMD)\n", name, ticks, ns_blk,
           mfma_per_block ? ns_blk / (mfma_per_block * waves_per_simd) : 0.0, waves_per_simd);
    CK(hipFree(d));
    return 0;
}

int main() {
    //                         A   B0  B1  P   ACC  F   ONE mfma valu ds fma3
    run("mfma only, kernel map (A=80,B=40/48)", probe<80, 40, 48, 56, 128, 74, 78, 1, 0, 0, 0>, 512, 2);
    run("mfma only, B at 42/50 (A,B banks differ)", probe<80, 42, 50, 58, 128, 76, 74, 1, 0, 0, 0>, 512, 2);
    run("mfma only, one wave per SIMD", probe<80, 40, 48, 56, 128, 74, 78, 1, 0, 0, 0>, 256, 2);
    run("valu only, kernel map (acc,P same bank)", probe<80, 40, 48, 56, 128, 74, 78, 0, 1, 0, 0>, 512, 0);
    run("valu only, P at 58 (banks differ)", probe<80, 40, 48, 58, 128, 76, 74, 0, 1, 0, 0>, 512, 0);
    run("valu only, v_fma_f32 (VOP3), kernel map", probe<80, 40, 48, 56, 128, 74, 78, 0, 1, 0, 1>, 512, 0);
    run("valu only, one wave per SIMD", probe<80, 40, 48, 56, 128, 74, 78, 0, 1, 0, 0>, 256, 0);
    run("mfma + valu, kernel map", probe<80, 40, 48, 56, 128, 74, 78, 1, 1, 0, 0>, 512, 2);
    run("mfma + valu, P at 58", probe<80, 40, 48, 58, 128, 76, 74, 1, 1, 0, 0>, 512, 2);
    run("mfma + valu, P at 58, B at 42/50", probe<80, 42, 50, 58, 128, 76, 74, 1, 1, 0, 0>, 512, 2);
    run("mfma + valu + lds, kernel map", probe<80, 40, 48, 56, 128, 74, 78, 1, 1, 1, 0>, 512, 2);
    run("mfma + valu + lds, P at 58, B at 42/50", probe<80, 42, 50, 58, 128, 76, 74, 1, 1, 1, 0>, 512, 2);
    run("mfma + lds, kernel map", probe<80, 40, 48, 56, 128, 74, 78, 1, 0, 1, 0>, 512, 2);
    run("lds only", probe<80, 40, 48, 56, 128, 74, 78, 0, 0, 1, 0>, 512, 0);
    //  v_pk_fma_f32 with a lo-broadcast scale pair: f pairs at v64..v71 (P at 48.., B at 32.. would collide: own map)
    run("valu only, v_pk_fma_f32 (6 VALU / block)", probe<80, 32, 40, 48, 128, 64, 72, 0, 1, 0, 2>, 512, 0);
    run("mfma + valu(pk)", probe<80, 32, 40, 48, 128, 64, 72, 1, 1, 0, 2>, 512, 2);
    run("mfma + valu(pk) + lds", probe<80, 32, 40, 48, 128, 64, 72, 1, 1, 1, 2>, 512, 2);
    run("mfma + valu(pk) + lds, one wave per SIMD", probe<80, 32, 40, 48, 128, 64, 72, 1, 1, 1, 2>, 256, 2);
    run("mfma + valu + lds, one wave per SIMD", probe<80, 40, 48, 56, 128, 74, 78, 1, 1, 1, 0>, 256, 2);
    return 0;
}
