// probe_stream.hip -- development probe: does the HBM read rate depend on HOW the weight bytes of a decode step are
// laid out?  The grouped-GEMM streamers read, per wavefront, one (16-row tile, K unit) chunk per step: 4 KiB for
// 16-bit weights, 2 KiB for fp8, 1 KiB for the 4-bit formats, the units of a tile contiguous in memory ("tile-major":
// every wavefront walks its own stream, the chip touches thousands of chunks a stream length apart).  The step time
// of the formats orders exactly like the chunk size (bf16 6.7 TB/s, fp8 6.3, 4-bit ~5) whatever the prefetch depth,
// which smells of DRAM page locality, not latency.  This probe reads the SAME bytes with the same instruction mix
// (CH/1024 x 16-byte loads per lane per step, ST steps in flight) in both orders:
//   tile-major : address = (stream * U + step) * CH          (what repack.hip writes today)
//   unit-major : address = (step * S + stream) * CH          (all streams' step-u chunks contiguous)
// build: hipcc -O3 --offload-arch=gfx950 tools/probe_stream.hip -o tools/_bin/probe_stream
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// one wavefront = one stream (64 lanes x 16 B = 1 KiB per load); CHK = KiB per step; ST = steps in flight
template <int CHK, int ST, bool UNIT_MAJOR>
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ buf, int U, int S, unsigned* out) {
    const int lane = threadIdx.x & 63;
    const int stream = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (stream >= S) return;
    auto addr = [&](int u, int c) {
        const size_t chunk = UNIT_MAJOR ? ((size_t)u * S + stream) : ((size_t)stream * U + u);
        return buf + (chunk * CHK + c) * 64 + lane;
    };
    u32x4 v[ST][CHK];
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < ST - 1; ++s)
#pragma unroll
        for (int c = 0; c < CHK; ++c) v[s][c] = __builtin_nontemporal_load(addr(s, c));
    int u = 0;
    for (; u + ST <= U; u += ST) {
#pragma unroll
        for (int s = 0; s < ST; ++s) {
            const int un = u + s + ST - 1;
#pragma unroll
            for (int c = 0; c < CHK; ++c) v[(s + ST - 1) % ST][c] = __builtin_nontemporal_load(addr(un < U ? un : U - 1, c));
#pragma unroll
            for (int c = 0; c < CHK; ++c) acc ^= v[s][c];
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = acc.x;
}

template <int CHK, int ST, bool UM>
static void run(const u32x4* buf, size_t bytes, int U, unsigned* out, int ldsb = 0) {
    const int S = (int)(bytes / ((size_t)U * CHK * 1024));
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const dim3 grid((S + 3) / 4), block(256);
    hipLaunchKernelGGL((k<CHK, ST, UM>), grid, block, ldsb, 0, buf, U, S, out);
    (void)hipEventRecord(a, 0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<CHK, ST, UM>), grid, block, ldsb, 0, buf, U, S, out);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    const double moved = (double)S * U * CHK * 1024 * reps;
    printf("%s chunk %d KiB  steps in flight %d  %-10s  streams %6d x %3d steps (%6.1f MB): %7.1f GB/s  (%.1f us per pass)\n",
           ldsb ? "[3 WG/CU]" : "[full occ]", CHK, ST - 1, UM ? "unit-major" : "tile-major", S, U, moved / reps / 1e6, moved / (ms * 1e-3) / 1e9,
           ms * 1e3 / reps);
}

int main() {
    const size_t bytes = (size_t)1880 << 20;      // Mixtral-8x7B bf16 w13: 1.88 GB; the 4-bit image is a quarter of it
    u32x4* buf;
    unsigned* out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, bytes);
    // the byte counts of one GEMM1 launch: bf16 1.88 GB in 4 KiB chunks x 64 steps; fp8 0.94 GB, 2 KiB x 32;
    // 4-bit 0.47 GB, 1 KiB x 32
    printf("-- 16-bit geometry (4 KiB per step, 64 steps per stream)\n");
    run<4, 2, false>(buf, bytes, 64, out);
    run<4, 2, true>(buf, bytes, 64, out);
    printf("-- fp8 geometry (2 KiB per step, 32 steps)\n");
    run<2, 2, false>(buf, bytes / 2, 32, out);
    run<2, 2, true>(buf, bytes / 2, 32, out);
    run<2, 4, false>(buf, bytes / 2, 32, out);
    run<2, 4, true>(buf, bytes / 2, 32, out);
    printf("-- 4-bit geometry (1 KiB per step, 32 steps)\n");
    run<1, 2, false>(buf, bytes / 4, 32, out);
    run<1, 2, true>(buf, bytes / 4, 32, out);
    run<1, 4, false>(buf, bytes / 4, 32, out);
    run<1, 4, true>(buf, bytes / 4, 32, out);
    run<1, 8, false>(buf, bytes / 4, 32, out);
    run<1, 8, true>(buf, bytes / 4, 32, out);
    printf("-- the same at the occupancy of the tile kernels (3 workgroups of 4 wavefronts per CU)\n");
    run<1, 2, false>(buf, bytes / 4, 32, out, 49152);
    run<1, 2, true>(buf, bytes / 4, 32, out, 49152);
    run<1, 4, false>(buf, bytes / 4, 32, out, 49152);
    run<1, 4, true>(buf, bytes / 4, 32, out, 49152);
    run<2, 2, false>(buf, bytes / 2, 32, out, 49152);
    run<2, 2, true>(buf, bytes / 2, 32, out, 49152);
    run<4, 2, false>(buf, bytes, 64, out, 49152);
    run<4, 2, true>(buf, bytes, 64, out, 49152);
    printf("-- 4-bit bytes, 4 steps of a stream fused into one 4 KiB chunk (8 steps)\n");
    run<4, 2, false>(buf, bytes / 4, 8, out);
    run<4, 2, true>(buf, bytes / 4, 8, out);
    return 0;
}
