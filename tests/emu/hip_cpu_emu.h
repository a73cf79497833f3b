// hip_cpu_emu.h -- TEST INFRASTRUCTURE: runs a HIP kernel's source on the CPU, one std::thread per lane, one
// workgroup at a time (so `__shared__` is a function-level static), `__syncthreads()` = a std::barrier.  It exists
// to check indexing / LDS-histogram / barrier logic of the small integer kernels in a container without a GPU; it
// is never part of the product and says nothing about performance.  Only what those kernels use is provided.
#pragma once
#include <barrier>
#include <cstdint>
#include <thread>
#include <vector>

struct emu_dim3 {
    unsigned x = 1, y = 1, z = 1;
};
static thread_local emu_dim3 threadIdx, blockIdx;
static emu_dim3 blockDim, gridDim;
static std::barrier<>* emu_barrier = nullptr;

#define __global__
#define __device__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

inline void __syncthreads() { emu_barrier->arrive_and_wait(); }
template <class T>
inline T atomicAdd(T* p, T v) {
    return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}

// launch<kernel>(grid, block, args...): workgroups run one after the other, the lanes of a workgroup concurrently
template <class K, class... A>
void emu_launch(K kernel, unsigned grid, unsigned block, A... args) {
    gridDim.x = grid;
    blockDim.x = block;
    for (unsigned b = 0; b < grid; ++b) {
        std::barrier<> bar(block);
        emu_barrier = &bar;
        std::vector<std::thread> lanes;
        lanes.reserve(block);
        for (unsigned t = 0; t < block; ++t)
            lanes.emplace_back([=] {
                threadIdx.x = t;
                blockIdx.x = b;
                kernel(args...);
            });
        for (auto& l : lanes) l.join();
    }
}
