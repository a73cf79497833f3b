#!/usr/bin/env python3
"""CPU-side ISA statistics of a compiled kernel's loops (no GPU needed): instruction-class counts (MFMA / VALU /
packed VALU / conversions / LDS / VMEM / waitcnt / scratch) per loop body of a gfx950 assembly listing.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -Iinclude lvllm_amd/csrc/gemm_tiled_int4_bf16.hip -o /tmp/k.s
    grep -n "^_ZN3lkm17gemm_tiled_kernel.*:" /tmp/k.s            # pick the kernel: first and last line of its body
    python tools/isa_loop_stats.py /tmp/k.s FIRST LAST [min_instructions]

Used for the int4 decode budget (10 VALU per MFMA at 32 rows per expert; the scale multipliers
hoisted out of the k-step loop: v_pk_mul_f32 32 -> 8 and v_perm_b32 16 -> 4 per two 128-k units)."""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split("\n")
start = int(sys.argv[2]); end = int(sys.argv[3])
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: labels[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i, m.group(1)))
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_pk_"): return "valu_pk"
    if op.startswith("v_cvt"): return "valu_cvt"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("buffer_"): return "vmem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    if op.startswith("scratch_"): return "scratch"
    return "other"
for a, b, lab in loops:
    c = Counter(); ops = Counter()
    for l in body[a:b+1]:
        m = re.match(r"\s+([a-z_0-9]+)", l)
        if m and not l.strip().startswith((".", ";")):
            c[cls(m.group(1))] += 1
            if cls(m.group(1)).startswith("valu"): ops[m.group(1)] += 1
    n = sum(c.values())
    if n > (int(sys.argv[4]) if len(sys.argv) > 4 else 60):
        print(f"loop {lab} lines {start+a}-{start+b} n={n}", dict(c))
        print("   ", ops.most_common(14))
