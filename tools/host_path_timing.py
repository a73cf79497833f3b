#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point (lk_moe cpu_prefill / lkm_prefill_host) on the
headline workload: hidden states, ids and weights start in host memory, the fp32 output is copied
back.  Reported in DESIGN.md next to the HBM-resident number; it is never the bench `value`."""
import sys, time
sys.path.insert(0, '.')
import torch
from bench import WORKLOADS, make_weights
from lvllm_amd import ops
wl = WORKLOADS["mixtral8x7b_bf16_decode_m32"]
E, K, H, I, M = wl["E"], wl["K"], wl["H"], wl["I"], wl["M"]
dev = torch.device("cuda", 0)
w13, w2 = make_weights(E, 0, H, I, dev, "bf16")
eng = ops.RoutedExpertsEngine(w13, w2, top_k=K, act_dtype=torch.bfloat16)
del w13, w2
g = torch.Generator().manual_seed(7)
x = (torch.randn((M, H), generator=g) / 10).to(torch.bfloat16)
logits = torch.randn((M, E), generator=g)
tw, ids = ops.topk_softmax(logits.to(dev), K, True)
tw, ids = tw.cpu(), ids.cpu()
for _ in range(5):
    eng.prefill_host(x, tw, ids)
n = 100
t0 = time.perf_counter()
for _ in range(n):
    eng.prefill_host(x, tw, ids)
dt = (time.perf_counter() - t0) / n
print(f"host-pointer path (PCIe both ways, blocking): {dt*1e6:.1f} us/step -> {M/dt:.0f} tokens/s")
