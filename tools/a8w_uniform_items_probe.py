#!/usr/bin/env python3
"""Is the fp8 x fp8 prefill kernel's L2-miss traffic (2.7 x algorithmic on GLM-4.5-Air M=8192) a consequence of UNEQUAL items?
The persistent workgroups of an XCD walk their item list without synchronising; items of experts with different row counts
last differently, so the workgroups that share a weight panel / a token tile drift apart.  This probe runs the same
layer on (a) the bench's router output (rows per expert 512 +- 22) and (b) a crafted routing in which EVERY expert gets
exactly 512 rows (two full 256-row tiles: all items identical) and prints the per-kernel HIP-event times; run it under
`rocprofv3 --kernel-trace --pmc FETCH_SIZE` for the bytes.
  python tools/a8w_uniform_items_probe.py [balanced|router]      (GPU box)"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from bench import WORKLOADS, build_engine  # noqa: E402
from lvllm_amd import ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "both"
wl_name = sys.argv[2] if len(sys.argv) > 2 else "glm45air_fp8w8a8_prefill_m8192"     # (any GLM-4.5-Air prefill workload of bench.py)
wl = WORKLOADS[wl_name]
E, K, H, I, M = wl["E"], wl["K"], wl["H"], wl["I"], wl["M"]
dev = torch.device("cuda", 0)
eng = build_engine(ops, wl, E, 0, dev, max_num_seqs=8192, max_batch_size=8192)[0]
gen = torch.Generator(device=dev).manual_seed(7)
x = (torch.randn((M, H), generator=gen, device=dev) / 10).to(torch.bfloat16)
logits = torch.randn((M, E), generator=gen, device=dev)
bias = torch.randn((E,), generator=torch.Generator(device=dev).manual_seed(99), device=dev) * 0.1
tw_r, ids_r = ops.grouped_topk(x, logits, K, True, 1, 1, "sigmoid", 1.0, bias)
m = torch.arange(M, device=dev, dtype=torch.int32)[:, None]
k = torch.arange(K, device=dev, dtype=torch.int32)[None, :]
ids_b = ((m * K + k) % E).to(torch.int32).contiguous()          # token m -> experts 8m .. 8m+7 (mod 128): 512 rows each
tw_b = torch.full((M, K), 1.0 / K, dtype=torch.float32, device=dev)
out = torch.empty((M, H), dtype=torch.bfloat16, device=dev)
eng.engine.set_profiling(True)
for name, tw, ids in (("router", tw_r, ids_r), ("balanced", tw_b, ids_b)):
    if mode not in ("both", name):
        continue
    cnt = torch.bincount(ids.flatten().long(), minlength=E)
    acc = {"sort": [], "gemm1": [], "gemm2": [], "combine": []}
    for i in range(13):
        eng.forward_rows(x, tw, ids, out=out)
        p = eng.engine.get_profile()
        if i >= 3:
            for kk in acc:
                acc[kk].append(p[kk] * 1e3)
    med = {kk: sorted(v)[len(v) // 2] for kk, v in acc.items()}
    flops1 = 4.0 * M * K * H * I
    tf = flops1 / med["gemm1"] / 1e6
    peak = 5000 if (wl["fmt"] == "fp8" and wl.get("fp8_mode")) else 2500
    print(f"{wl_name} {name:9s} rows/expert min {int(cnt.min())} max {int(cnt.max())} | gemm1 {med['gemm1']:.1f} us ({tf:.0f} TFLOP/s = "
          f"{tf / peak:.3f} of {peak / 1000} PF) gemm2 {med['gemm2']:.1f} sort {med['sort']:.1f} combine {med['combine']:.1f} | "
          f"{eng.engine.last_kernels()['gemm1'][0]}", flush=True)
