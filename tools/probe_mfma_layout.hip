// probe_mfma_layout.hip -- stand-alone MI355X probe: where the operands of v_mfma_f32_32x32x64_f8f6f4 (fp8 e4m3 x e4m3) sit
// in the lanes' registers, found with one-hot operands (development; feeds DESIGN.md section 10: the 32-token block layout).
//   A / B: 8 VGPRs = 32 bytes per lane.  D: 16 VGPRs per lane.
//   (1) A one-hot at (lane L, byte b), B all ones -> the non-zero D positions are one ROW of the result: which register i
//       and which lane half (l / 32) a weight row loaded into lane L comes out in, and whether all 32 bytes of a lane are
//       one row;
//   (2) B one-hot at (L, b), A all ones -> one COLUMN: which lanes;
//   (3) k alignment: hypothesis k(L, b) = 32 * (L / 32) + b for both operands; A = indicator{k == kh}, B = indicator{k == kh'}
//       must give D == 1 everywhere for kh' == kh and 0 everywhere for kh' != kh.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probe_mfma_layout.hip -o /tmp/probe_mfma_layout
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x)                                                     \
    do {                                                          \
        hipError_t e_ = (x);                                      \
        if (e_ != hipSuccess) {                                   \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_)); \
            return 1;                                             \
        }                                                         \
    } while (0)

// one wave; a, b: [64 lanes][32 bytes]; d: [64 lanes][16 floats]
__global__ __launch_bounds__(64) void mfma_once(const unsigned char* __restrict__ a, const unsigned char* __restrict__ b,
                                                float* __restrict__ d) {
    const int lane = threadIdx.x;
    i32x8 av, bv;
    memcpy(&av, a + lane * 32, 32);
    memcpy(&bv, b + lane * 32, 32);
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // cbsz = blgp = 0: both operands fp8 (e4m3); unit scales
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int i = 0; i < 16; ++i) d[lane * 16 + i] = acc[i];
}

static unsigned char *da, *db;
static float* dd;
static unsigned char ha[2048], hb[2048];
static float hd[1024];

static int run() {
    CK(hipMemcpy(da, ha, 2048, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb, 2048, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mfma_once, dim3(1), dim3(64), 0, 0, da, db, dd);
    CK(hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost));
    return 0;
}

int main() {
    const unsigned char ONE = 0x38;      // e4m3 1.0
    CK(hipMalloc(&da, 2048));
    CK(hipMalloc(&db, 2048));
    CK(hipMalloc(&dd, sizeof(hd)));
    // (1) rows: A one-hot
    printf("== A one-hot at (lane, byte) -> D positions (register i, lane half h) that are non-zero, and how many lanes\n");
    memset(hb, ONE, 2048);
    for (int L = 0; L < 64; ++L) {
        int sig0 = -1, same = 1;
        for (int b = 0; b < 32; ++b) {
            memset(ha, 0, 2048);
            ha[L * 32 + b] = ONE;
            if (run()) return 1;
            int sig = -1, lanes = 0;
            for (int l = 0; l < 64; ++l)
                for (int i = 0; i < 16; ++i)
                    if (hd[l * 16 + i] != 0.f) {
                        const int s = i * 2 + (l >> 5);
                        if (sig < 0) sig = s;
                        else if (sig != s) sig = 1000 + s;      // (more than one (i, h): not a single row)
                        ++lanes;
                    }
            if (b == 0) {
                sig0 = sig;
                printf("A lane %2d byte  0: register %2d, half %d, %2d lanes\n", L, sig / 2, sig & 1, lanes);
            } else if (sig != sig0) same = 0;
        }
        if (!same) printf("A lane %2d: its 32 bytes do NOT all land in one row\n", L);
    }
    // (2) columns: B one-hot
    printf("== B one-hot at (lane, byte 0) -> lanes of D that are non-zero (all 16 registers expected)\n");
    memset(ha, ONE, 2048);
    for (int L = 0; L < 64; ++L) {
        memset(hb, 0, 2048);
        hb[L * 32] = ONE;
        if (run()) return 1;
        int l0 = -1, l1 = -1, n = 0;
        for (int l = 0; l < 64; ++l) {
            int any = 0;
            for (int i = 0; i < 16; ++i) any |= hd[l * 16 + i] != 0.f;
            if (any) {
                if (l0 < 0) l0 = l;
                else l1 = l;
                ++n;
            }
        }
        printf("B lane %2d: D lanes %d and %d (%d lanes)\n", L, l0, l1, n);
    }
    // (3) k alignment under the hypothesis k = 32 * (lane / 32) + byte
    int bad = 0;
    for (int kh = 0; kh < 64; ++kh)
        for (int dk = 0; dk < 2; ++dk) {
            const int kb = (kh + dk) & 63;
            memset(ha, 0, 2048);
            memset(hb, 0, 2048);
            for (int l = 0; l < 64; ++l)
                for (int b = 0; b < 32; ++b) {
                    const int k = 32 * (l >> 5) + b;
                    if (k == kh) ha[l * 32 + b] = ONE;
                    if (k == kb) hb[l * 32 + b] = ONE;
                }
            if (run()) return 1;
            const float want = dk ? 0.f : 1.f;
            for (int q = 0; q < 1024; ++q) bad += hd[q] != want;
        }
    printf("== k hypothesis (k = 32 * (lane / 32) + byte, both operands): %s (%d mismatching result entries)\n", bad ? "WRONG" : "confirmed", bad);
    return 0;
}
