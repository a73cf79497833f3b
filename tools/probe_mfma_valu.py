#!/usr/bin/env python3
"""MI355X probe: how many VALU instructions hide beside one MFMA on a SIMD (development; feeds DESIGN.md section 4).

Writes a HIP file of hand-placed instruction streams (inline asm, nothing left to the compiler's scheduler), builds it with
hipcc and runs it on the GPU of the box:
  * MFMA only / VALU only / one MFMA followed by K VALU instructions, for the 16x16x32 bf16, 32x32x16 bf16 and
    16x16x128 f8f6f4 MFMAs and for the VALU instructions the 4-bit decoders and the fp8 prefill kernel are made of;
  * one to three waves per SIMD running the same stream;
  * two waves per SIMD with the roles split: one wave only MFMAs, its partner only VALU;
  * clusters of MFMAs followed by their VALU instead of the interleave;
  * the fp8 prefill kernel's block stream (gemm_prefill_a8w.h: LDS reads, scale products, MFMAs, accumulator updates, the
    lgkmcnt wait) per quantum of 32 tokens x 32 weight rows x 128 k, in today's 16-token layout (16x16x128 MFMAs) and in a
    32-token layout on the 32x32x64 MFMA (quantum_body).
Output: ns per group (one MFMA + K VALU) per SIMD, and the same in cycles at the clock the MFMA-only stream implies.
  python tools/probe_mfma_valu.py [out.log] [--quick] [--build-only]
"""
import subprocess
import sys
import tempfile
from pathlib import Path

GROUPS = 16          # groups per asm block (accumulators rotate over 4, VALU temporaries over 8)

MFMA = {
    0: None,
    1: "v_mfma_f32_16x16x32_bf16 %[acc{a}], %[A4], %[B4], %[acc{a}]",
    2: "v_mfma_f32_32x32x16_bf16 %[big{b}], %[A4], %[B4], %[big{b}]",
    3: "v_mfma_f32_16x16x128_f8f6f4 %[acc{a}], %[A8], %[B8], %[acc{a}]",
    4: "v_mfma_f32_32x32x64_f8f6f4 %[big{b}], %[A8], %[B8], %[big{b}]",
}
MFMA_NAME = {0: "-", 1: "16x16x32 bf16", 2: "32x32x16 bf16", 3: "16x16x128 fp8", 4: "32x32x64 fp8"}
MFMA_CYC = {1: 16, 2: 32, 3: 32, 4: 64}          # nominal pipe cycles per instruction

VALU = {
    "fma": "v_fma_f32 %[t{j}], %[t{j}], %[c], %[c]",
    "pkfma": "v_pk_fma_f32 %[p{q}], %[p{q}], %[pc], %[pc]",
    "pkmul": "v_pk_mul_f32 %[p{q}], %[p{q}], %[pc]",
    "cvt_fp8": "v_cvt_pk_f32_fp8 %[p{q}], %[t{j}]",
    "cvt_bf16": "v_cvt_pk_bf16_f32 %[t{j}], %[t{j}], %[c]",
    "perm": "v_perm_b32 %[t{j}], %[t{j}], %[c], %[c]",
    "andor": "v_and_or_b32 %[t{j}], %[t{j}], %[c], %[c]",
    "mul": "v_mul_f32 %[t{j}], %[t{j}], %[c]",
    "pkadd": "v_pk_add_f32 %[p{q}], %[p{q}], %[pc]",
    "sc_bf16_fp4": "v_cvt_scalef32_pk_bf16_fp4 %[t{j}], %[t{j}], %[c]",
    "sc_f32_fp4": "v_cvt_scalef32_pk_f32_fp4 %[p{q}], %[t{j}], 1.0",
    "exp": "v_exp_f32 %[t{j}], %[t{j}]",
    "pkmul_f16": "v_pk_mul_f16 %[t{j}], %[t{j}], %[c]",
    "pkfma_f16": "v_pk_fma_f16 %[t{j}], %[t{j}], %[c], %[c]",
}


def body(mf, k, op, cl=1):
    """GROUPS groups of (one MFMA + k VALU); cl > 1: cl MFMAs back to back, then their cl * k VALU"""
    lines = []
    j = 0
    for g0 in range(0, GROUPS, cl):
        if MFMA[mf]:
            for g in range(g0, g0 + cl):
                lines.append(MFMA[mf].format(a=g % 4, b=g % 2))
        for _ in range(k * cl):
            lines.append(VALU[op].format(j=j % 8, q=j % 4))
            j += 1
    return "\\n\\t".join(lines)


def quantum_body(layout, gated=True, quanta=8):
    """One K unit's worth of the fp8 prefill kernel's block stream, per QUANTUM = 32 tokens x 32 weight rows x 128 k
    (128 matrix-pipe cycles either way).  layout 16: today's blocks -- per 16-token block 2 B-operand reads + 1 scale read,
    the scale products, 2 MFMAs 16x16x128 with the previous block's 2 x 4 accumulator updates between them, one lgkmcnt
    wait; layout 32: one 32-token block -- 4 B-operand reads + 1 scale read, the scale products, 2 chained MFMAs 32x32x64
    with the previous block's 16 accumulator updates between / behind them, one wait.  (The updates read registers that
    no MFMA of the stream writes: issue cost only, as in the kernel, where they read the PREVIOUS block's results.)"""
    L = []
    j = 0
    def valu(n):
        nonlocal j
        for _ in range(n):
            L.append(VALU["fma"].format(j=j % 8, q=j % 4))
            j += 1
    for _ in range(quanta):
        if layout == 16:
            for blk in range(2):
                L.append("ds_read_b128 %[acc2], %[la] offset:" + str(blk * 2048))
                L.append("ds_read_b128 %[acc3], %[la] offset:" + str(blk * 2048 + 1024))
                L.append("ds_read_b32 %[t7], %[la] offset:8192")
                valu(2 if gated else 1)
                L.append(MFMA[3].format(a=0))
                valu(4)
                L.append(MFMA[3].format(a=1))
                valu(4)
                L.append("s_waitcnt lgkmcnt(0)")
        else:
            for k in range(4):
                L.append("ds_read_b128 %[acc" + str(k) + "], %[la] offset:" + str(k * 1024))
            L.append("ds_read_b32 %[t7], %[la] offset:8192")
            valu(2 if gated else 1)
            L.append("v_mfma_f32_32x32x64_f8f6f4 %[big0], %[A8], %[B8], 0")
            valu(8)
            L.append("v_mfma_f32_32x32x64_f8f6f4 %[big0], %[A8], %[B8], %[big0]")
            valu(8)
            L.append("s_waitcnt lgkmcnt(0)")
    return "\\n\\t".join(L)


OPERANDS = """: [acc0] "+v"(acc[0]), [acc1] "+v"(acc[1]), [acc2] "+v"(acc[2]), [acc3] "+v"(acc[3]), [big0] "+v"(big[0]), [big1] "+v"(big[1]),
              [t0] "+v"(t[0]), [t1] "+v"(t[1]), [t2] "+v"(t[2]), [t3] "+v"(t[3]), [t4] "+v"(t[4]), [t5] "+v"(t[5]), [t6] "+v"(t[6]), [t7] "+v"(t[7]),
              [p0] "+v"(p[0]), [p1] "+v"(p[1]), [p2] "+v"(p[2]), [p3] "+v"(p[3])
            : [A4] "v"(a4), [B4] "v"(b4), [A8] "v"(a8), [B8] "v"(b8), [c] "v"(c), [pc] "v"(pc), [la] "v"(la)"""

HEAD = r"""
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define PROLOGUE \
    const int lane = threadIdx.x & 63; \
    u32x4 a4 = in[lane], b4 = in[64 + lane]; \
    u32x8 a8, b8; \
    for (int i = 0; i < 8; ++i) { a8[i] = in[128 + lane][i & 3] + i; b8[i] = in[192 + lane][i & 3] + i; } \
    f32x4 acc[4]; f32x16 big[2]; unsigned t[8]; u32x2 p[4]; \
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; \
    for (int i = 0; i < 2; ++i) for (int q = 0; q < 16; ++q) big[i][q] = 0.f; \
    for (int i = 0; i < 8; ++i) t[i] = in[256 + lane][i & 3] + i; \
    for (int i = 0; i < 4; ++i) p[i] = u32x2{in[320 + lane][i], in[320 + lane][(i + 1) & 3]}; \
    unsigned c = in[384 + lane].x; u32x2 pc = u32x2{in[384 + lane].y, in[384 + lane].z}; \
    __shared__ u32x4 ldsbuf[640]; \
    ldsbuf[threadIdx.x & 511] = in[lane]; ldsbuf[512 + (threadIdx.x & 127)] = in[64 + lane]; \
    const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ldsbuf + lane * 16; \
    __syncthreads(); \
    const long long t0 = __builtin_readcyclecounter();
#define EPILOGUE \
    const long long t1 = __builtin_readcyclecounter(); \
    float s = 0.f; \
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w; \
    for (int i = 0; i < 2; ++i) for (int q = 0; q < 16; ++q) s += big[i][q]; \
    for (int i = 0; i < 8; ++i) s += __builtin_bit_cast(float, t[i]); \
    for (int i = 0; i < 4; ++i) s += __builtin_bit_cast(float, p[i].x) + __builtin_bit_cast(float, p[i].y); \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s; \
    if (lane == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
"""


def kernel(name, mf, k, op, cl=1):
    return f"""
__global__ __launch_bounds__(1024) void {name}(const u32x4* __restrict__ in, float* __restrict__ out, long long* __restrict__ cyc, int iters) {{
    PROLOGUE
    for (int it = 0; it < iters; ++it) {{
        asm volatile("{body(mf, k, op, cl)}"
            {OPERANDS});
    }}
    EPILOGUE
}}
"""


def custom_kernel(name, body_text):
    return f"""
__global__ __launch_bounds__(1024) void {name}(const u32x4* __restrict__ in, float* __restrict__ out, long long* __restrict__ cyc, int iters) {{
    PROLOGUE
    for (int it = 0; it < iters; ++it) {{
        asm volatile("{body_text}"
            {OPERANDS} : "memory");
    }}
    EPILOGUE
}}
"""


def split_kernel(name, mf, k, op):
    """waves 0-3 of a 512-thread workgroup (first wave of each SIMD): MFMAs only; waves 4-7: VALU only"""
    return f"""
__global__ __launch_bounds__(1024) void {name}(const u32x4* __restrict__ in, float* __restrict__ out, long long* __restrict__ cyc, int iters) {{
    PROLOGUE
    if ((threadIdx.x >> 8) & 1) {{
        for (int it = 0; it < iters; ++it) {{
            asm volatile("{body(0, k, op)}"
                {OPERANDS});
        }}
    }} else {{
        for (int it = 0; it < iters; ++it) {{
            asm volatile("{body(mf, 0, op)}"
                {OPERANDS});
        }}
    }}
    EPILOGUE
}}
"""


def main():
    quick = "--quick" in sys.argv      # only the cases added last (packed / scaled conversions, clustered streams)
    cases = []           # (name, mf, k, op, split, cluster)
    def add(mf, k, op="fma", split=False, cl=1, new=False):
        if quick and not new:
            return
        name = f"k_{'s' if split else 'n'}_{mf}_{k}_{op}_{cl}"
        cases.append((name, mf, k, op, split, cl))
    for k in (2, 4):
        add(0, k)
    for mf, ks in ((1, (0, 1, 2, 3, 4, 6)), (2, (0, 2, 4, 5, 6, 8, 10)), (3, (0, 2, 4, 5, 6, 8, 10))):
        for k in ks:
            add(mf, k)
    for op in ("pkfma", "pkmul", "cvt_fp8", "cvt_bf16", "perm", "andor", "mul"):
        add(0, 4, op)
        add(1, 2, op)
        add(2, 5, op)
        add(3, 5, op)
    for mf in (1, 2, 3):
        for k in (2, 4, 8):
            add(mf, k, "fma", True)
    for op in ("pkadd", "sc_bf16_fp4", "sc_f32_fp4", "exp", "pkmul_f16", "pkfma_f16"):
        add(0, 4, op, new=True)
        add(1, 2, op, new=True)
        add(2, 5, op, new=True)
    for k in (0, 8, 10, 12, 13, 14, 16):
        add(4, k, new=True)
    for op in ("pkfma", "fma"):          # 4 or 8 MFMAs back to back, then their VALU
        for cl in (4, 8, 16):
            add(1, 2, op, cl=cl, new=True)
            add(1, 4, op, cl=cl, new=True)
    src = [HEAD]
    for name, mf, k, op, split, cl in cases:
        src.append(split_kernel(name, mf, k, op) if split else kernel(name, mf, k, op, cl))
    customs = [("q16g", "block stream, 16-token blocks (today), gated: 4 MFMA 16x16x128 + 20 VALU + 6 ds_read per quantum", quantum_body(16, True)),
               ("q16p", "block stream, 16-token blocks (today), one scale: 4 MFMA 16x16x128 + 18 VALU + 6 ds_read per quantum", quantum_body(16, False)),
               ("q32g", "block stream, 32-token blocks, gated: 2 MFMA 32x32x64 + 18 VALU + 5 ds_read per quantum", quantum_body(32, True)),
               ("q32p", "block stream, 32-token blocks, one scale: 2 MFMA 32x32x64 + 17 VALU + 5 ds_read per quantum", quantum_body(32, False))]
    for name, what, body_text in customs:
        src.append(custom_kernel("k_" + name, body_text))
    src.append(r"""
struct Case { const char* what; void (*fn)(const u32x4*, float*, long long*, int); int mf, k, split; };
int main() {
    u32x4* in; float* out; long long* cyc;
    CK(hipMalloc(&in, 448 * 16)); CK(hipMalloc(&out, 256 * 1024 * 4)); CK(hipMalloc(&cyc, 256 * 16 * 8));
    CK(hipMemset(in, 0x3c, 448 * 16));
    const Case cases[] = {
""")
    for name, what, body_text in customs:
        src.append(f'        {{"{what}", k_{name}, -1, 0, 0}},\n')
    for name, mf, k, op, split, cl in cases:
        what = (f"split: wave A {MFMA_NAME[mf]} only | wave B {k} {op} per group" if split
                else f"{MFMA_NAME[mf]:14s} + {k:2d} {op}" + (f" (clusters of {cl} MFMAs)" if cl > 1 else ""))
        src.append(f'        {{"{what}", {name}, {mf}, {k}, {1 if split else 0}}},\n')
    src.append(r"""    };
    const int iters = 2000, groups = %d;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Case& cs : cases) {
        for (int wps = 1; wps <= 3; ++wps) {
            if (cs.split && wps != 2) continue;
            const int threads = 256 * wps;
            hipLaunchKernelGGL(cs.fn, dim3(256), dim3(threads), 0, 0, in, out, cyc, iters);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(cs.fn, dim3(256), dim3(threads), 0, 0, in, out, cyc, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            long long c[16]; CK(hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost));
            const double n = (double)iters * (cs.mf < 0 ? 8 : groups);      // (block streams: 8 quanta per asm block)
            // per SIMD: wps waves each ran n groups (split: one wave the MFMAs, one the VALU of n groups)
            const double ns_group = ms * 1e6 / n / (cs.split ? 1 : wps);
            printf("%%-62s %%d waves/SIMD  %%7.2f ns per group per SIMD   wave0 %%6.1f  last wave %%6.1f ticks/group\n", cs.what, wps, ns_group,
                   c[0] / n, c[4 * wps - 1] / n);
        }
    }
    return 0;
}
""" % GROUPS)
    d = Path(tempfile.mkdtemp(prefix="probe_mv_"))
    (d / "p.hip").write_text("".join(src))
    exe = d / "p"
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", str(d / "p.hip"), "-o", str(exe)])
    if "--build-only" in sys.argv:
        print("built", exe)
        return
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    if r.returncode:
        print(r.stderr[-2000:])
    outs = [a for a in sys.argv[1:] if not a.startswith("--")]
    if outs:
        Path(outs[0]).write_text(r.stdout)


if __name__ == "__main__":
    main()
