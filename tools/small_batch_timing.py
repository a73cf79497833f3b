#!/usr/bin/env python3
"""Small-batch decode (1..16 tokens) at a WORKLOADS layer shape (default: Qwen3-30B-A3B), hipGraph replay of the whole step
(lkm_forward_routed): the no-scatter path (one to four tokens: router inside GEMM1, GEMM2 + weighted sum in one
workgroup per token: two launches) against the general path (router, sort, GEMM1, GEMM2, combine).  Development tool."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from bench import WORKLOADS, build_engine  # noqa: E402
from lvllm_amd import ops  # noqa: E402


def timed(fn, steps=400):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(5):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        fn()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps * 1e3


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "qwen3_30b_a3b_bf16_decode_m1"
    wl = dict(WORKLOADS[name])
    print(name, flush=True)
    E, K, H = wl["E"], wl["K"], wl["H"]
    dev = torch.device("cuda", 0)
    eng = build_engine(ops, wl, E, 0, dev)[0]
    gen = torch.Generator(device=dev).manual_seed(7)
    for M in (1, 2, 3, 4, 6, 8, 12, 16):
        x = (torch.randn((M, H), generator=gen, device=dev) / 10).to(torch.bfloat16)
        logits = torch.randn((M, E), generator=gen, device=dev)
        out = torch.empty((M, H), dtype=torch.float32, device=dev)
        res = []
        for tune in (dict(fuse=0, direct=16), dict(fuse=-1, direct=1 if M > 1 else 0), dict(fuse=-1, direct=-1)):
            eng.engine.set_tuning(**tune)
            res.append(timed(lambda: eng.forward_logits(x, logits, K, True, out=out)))
        print(f"M={M}: no-scatter path {res[0]:6.1f} us | router launch + (M=1: direct, M>1: sort / GEMM1 / GEMM2 / combine) "
              f"{res[1]:6.1f} us | five launches {res[2]:6.1f} us | {eng.engine.describe()[:0]}", flush=True)


if __name__ == "__main__":
    main()
