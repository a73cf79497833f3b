// probe_l2.hip -- development probe: how many bytes per clock can a CU pull out of an L2-RESIDENT buffer?
// (The prefill kernels deliver their operands at ~13.5 B/clk/CU whatever the L2 hit rate; is that the
// L2 -> CU ceiling or the kernels' own load structure?)  Every workgroup streams the same `window` bytes
// `reps` times with `U` 16-byte loads per lane in flight; windows of 256 KiB .. 64 MiB walk from L2-resident
// to Infinity-Cache/HBM-resident.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int U, bool LDS>
__global__ __launch_bounds__(256) void k(const u32x4* buf, size_t n_vec, int reps, unsigned* out) {
    __shared__ u32x4 stage[LDS ? U * 256 : 1];
    u32x4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)256 * U;
    // workgroups start at different offsets so that they do not all hit the same channel at once
    size_t base = ((size_t)blockIdx.x * 7919 * stride) % n_vec;
    for (int r = 0; r < reps; ++r) {
        for (size_t i = 0; i < n_vec; i += stride) {
            size_t o = base + i;
            if (o >= n_vec) o -= n_vec;
            u32x4 v[U];
#pragma unroll
            for (int j = 0; j < U; ++j) v[j] = buf[o + (size_t)j * 256 + threadIdx.x];
            if (LDS) {
#pragma unroll
                for (int j = 0; j < U; ++j) stage[j * 256 + threadIdx.x] = v[j];
                __syncthreads();
                acc ^= stage[(threadIdx.x * 7 + r) % (U * 256)];
                __syncthreads();
            } else {
#pragma unroll
                for (int j = 0; j < U; ++j) acc ^= v[j];
            }
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = acc.x;
}

template <int U, bool LDS>
static void run(const char* name, const u32x4* buf, size_t bytes, int blocks, unsigned* out) {
    const size_t n_vec = bytes / 16 / (256 * U) * (256 * U);
    const double target = 24e9;   // bytes moved per measurement
    int reps = (int)(target / ((double)blocks * n_vec * 16));
    if (reps < 1) reps = 1;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<U, LDS>), dim3(blocks), dim3(256), 0, 0, buf, n_vec, 1, out);
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL((k<U, LDS>), dim3(blocks), dim3(256), 0, 0, buf, n_vec, reps, out);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    const double moved = (double)blocks * n_vec * 16 * reps;
    const double gbs = moved / (ms * 1e-3) / 1e9;
    printf("%-34s window %7.2f MiB blocks %5d: %8.1f GB/s = %5.1f B/clk/CU at 2.1 GHz (%.2f ms)\n", name,
           bytes / 1048576.0, blocks, gbs, gbs * 1e9 / 256 / 2.1e9, ms);
}

int main() {
    const size_t cap = (size_t)256 << 20;
    u32x4* buf; unsigned* out;
    (void)hipMalloc(&buf, cap); (void)hipMalloc(&out, 64);
    (void)hipMemset(buf, 1, cap);
    for (size_t w : {(size_t)256 << 10, (size_t)1 << 20, (size_t)2 << 20, (size_t)8 << 20, (size_t)32 << 20, (size_t)128 << 20}) {
        run<4, false>("regs, 4 x 16 B in flight / lane", buf, w, 2048, out);
        run<8, false>("regs, 8 x 16 B in flight / lane", buf, w, 2048, out);
        run<16, false>("regs, 16 x 16 B in flight / lane", buf, w, 1024, out);
        run<8, true>("via LDS + 2 barriers, 8 x 16 B", buf, w, 2048, out);
    }
    return 0;
}
