// oracle/_ref build shim (test infrastructure only).  Force-included (-include) in front of the reference's
// own csrc/cpu sources, which are compiled WHERE THEY LIE under /root/reference -- nothing of them is copied.
// It supplies the two things this image's toolchain lacks for those sources:
//   1. at::cpu::get_cpu_capabilities() -- called by csrc/cpu/utils.hpp:84 for the L2 size, absent from the
//      ATen/cpu/Utils.h of torch 2.10;
//   2. _mm512_extracti32x8_epi32 with a run-time 0/1 selector -- csrc/cpu/cpu_types_x86.hpp:536 passes a
//      function argument; GCC folds it after inlining, clang's macro form demands an immediate.  (GCC 11, the
//      only gcc here, lacks __bfloat16 / _mm_cvtness_sbh used at cpu_types_x86.hpp:988, hence ROCm's clang.)
#pragma once
#include <immintrin.h>
#include <unistd.h>

#include <string>
#include <unordered_map>

#include <ATen/core/ivalue.h>

namespace at {
namespace cpu {
inline std::unordered_map<std::string, c10::IValue> get_cpu_capabilities() {
  long l2 = sysconf(_SC_LEVEL2_CACHE_SIZE);
  if (l2 <= 0) l2 = 1024 * 1024;
  std::unordered_map<std::string, c10::IValue> caps;
  caps.emplace("l2_cache_size", c10::IValue(static_cast<int64_t>(l2)));
  return caps;
}
}  // namespace cpu
}  // namespace at

#undef _mm512_extracti32x8_epi32
static inline __m256i _mm512_extracti32x8_epi32(__m512i a, int upper) {
  const __v8si und = (__v8si)_mm256_undefined_si256();
  return upper ? (__m256i)__builtin_ia32_extracti32x8_mask((__v16si)a, 1, und, (__mmask8)-1)
               : (__m256i)__builtin_ia32_extracti32x8_mask((__v16si)a, 0, und, (__mmask8)-1);
}
