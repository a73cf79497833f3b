// probe_wave_place.hip -- stand-alone MI355X probe: on which SIMD of its CU does wave w of a workgroup land, for workgroups of
// 5 / 8 / 9 / 15 waves with two (or three) workgroups resident per CU?  gemm_w4e.h gives wave 0 the loader role and waves 1..NC
// the decode + MFMA work: the SIMD that hosts the loaders hosts fewer consumers, and the kernel's step time is the time of the
// most loaded SIMD.  Each wave records HW_ID (SIMD, CU, SH, SE) and XCC_ID; the host prints, per workgroup size, the histogram
// of "consumer waves per SIMD" over the CUs that held the expected number of workgroups.
// build: hipcc -O2 --offload-arch=gfx950 tools/probe_wave_place.hip -o tools/_bin/probe_wave_place
#include <hip/hip_runtime.h>

#include <cstdio>
#include <map>
#include <vector>

#define CK(x)                                                     \
    do {                                                          \
        hipError_t e_ = (x);                                      \
        if (e_ != hipSuccess) {                                   \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_)); \
            return 1;                                             \
        }                                                         \
    } while (0)

__global__ void k_place(unsigned* out, int spin) {
    extern __shared__ char lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * nw + wave) * 2 + 0] = hw;
        out[(blockIdx.x * nw + wave) * 2 + 1] = xcc;
    }
    // stay resident long enough for every workgroup of the grid to be placed beside this one
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
    if (spin < 0) lds[threadIdx.x] = 1;
}

int main() {
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    struct Case { int waves, wgs_per_cu, lds; };
    const Case cases[] = {{5, 2, 76800}, {5, 3, 51200}, {8, 2, 65024}, {9, 2, 69632}, {15, 1, 120000}, {7, 2, 60000}, {4, 4, 40000}, {8, 1, 97536}};
    unsigned* out;
    CK(hipMalloc(&out, 1 << 22));
    CK(hipFuncSetAttribute((const void*)k_place, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (const Case& c : cases) {
        const int grid = cus * c.wgs_per_cu;
        CK(hipMemset(out, 0, 1 << 22));
        k_place<<<grid, c.waves * 64, c.lds>>>(out, 2000);
        CK(hipDeviceSynchronize());
        std::vector<unsigned> h((size_t)grid * c.waves * 2);
        CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
        // CU key: xcc, se, sh, cu
        std::map<unsigned, std::vector<std::vector<int>>> per_cu;   // key -> list of workgroups -> SIMD of each wave
        for (int b = 0; b < grid; ++b) {
            std::vector<int> simds;
            unsigned key = 0;
            for (int w = 0; w < c.waves; ++w) {
                const unsigned hw = h[(size_t)(b * c.waves + w) * 2], xcc = h[(size_t)(b * c.waves + w) * 2 + 1] & 0xf;
                simds.push_back((hw >> 4) & 3);
                key = (xcc << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 8) | ((hw >> 8) & 0xf);
            }
            per_cu[key].push_back(simds);
        }
        std::map<std::vector<int>, int> hist;      // consumer waves (wave index >= 1) per SIMD, sorted by SIMD id
        std::map<std::vector<int>, int> first;     // SIMD of wave 0 of each co-resident workgroup
        int full = 0;
        for (auto& kv : per_cu) {
            if ((int)kv.second.size() != c.wgs_per_cu) continue;
            ++full;
            std::vector<int> cons(4, 0), f;
            for (auto& wg : kv.second) {
                f.push_back(wg[0]);
                for (size_t w = 1; w < wg.size(); ++w) cons[wg[w]]++;
            }
            hist[cons]++;
            first[f]++;
        }
        printf("== %d waves per workgroup, %d per CU asked (LDS %d): %zu CUs seen, %d held exactly %d\n", c.waves, c.wgs_per_cu, c.lds, per_cu.size(), full, c.wgs_per_cu);
        {
            auto& wg = per_cu.begin()->second[0];
            printf("   SIMD of waves 0..%d of one workgroup:", c.waves - 1);
            for (int s : wg) printf(" %d", s);
            printf("\n");
        }
        for (auto& kv : hist) printf("   consumers per SIMD [%d %d %d %d] (max %d of %d): %d CUs\n", kv.first[0], kv.first[1], kv.first[2], kv.first[3],
                                     std::max(std::max(kv.first[0], kv.first[1]), std::max(kv.first[2], kv.first[3])), kv.first[0] + kv.first[1] + kv.first[2] + kv.first[3], kv.second);
        int shown = 0;
        for (auto& kv : first) {
            if (shown++ >= 6) break;
            printf("   SIMD of wave 0 of the co-resident workgroups:");
            for (int s : kv.first) printf(" %d", s);
            printf("  x %d CUs\n", kv.second);
        }
    }
    return 0;
}
