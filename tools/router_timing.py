#!/usr/bin/env python3
"""Router GEMM + top-k timing (development / profiles): fused lkm_router_gemm_topk vs torch F.linear +
the stand-alone routing operator, hipGraph replay, for the model shapes of SURVEY 8."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from lvllm_amd import ops  # noqa: E402

SHAPES = [("Mixtral-8x7B", 32, 4096, 8, 2, {}),
          ("Qwen3-30B-A3B", 1, 2048, 128, 8, {}),
          ("Qwen3-30B-A3B", 128, 2048, 128, 8, {}),
          ("DeepSeek-V3", 16, 7168, 256, 8, dict(scoring_func="sigmoid", num_expert_group=8, topk_group=4)),
          ("DeepSeek-V3", 256, 7168, 256, 8, dict(scoring_func="sigmoid", num_expert_group=8, topk_group=4)),
          ("GLM-4.5-Air prefill", 8192, 4096, 128, 8, {})]


def timed(fn, reps=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(10):
                fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps // 10):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (reps // 10 * 10)


def main():
    dev = torch.device("cuda", 0)
    for name, M, H, E, K, kw in SHAPES:
        gen = torch.Generator(device=dev).manual_seed(1)
        x = (torch.randn((M, H), generator=gen, device=dev) / 4).to(torch.bfloat16)
        w = (torch.randn((E, H), generator=gen, device=dev) / 8).to(torch.bfloat16)
        fused = timed(lambda: ops.router_topk(x, w, K, True, **kw))

        def unfused():
            lg = torch.nn.functional.linear(x, w).float()
            if kw:
                return ops.grouped_topk(x, lg, K, True, kw["num_expert_group"], kw["topk_group"], kw["scoring_func"])
            return ops.topk_softmax(lg, K, True)
        base = timed(unfused)
        wbytes = E * H * 2
        print(f"{name:22s} M={M:5d} H={H} E={E:3d}: fused router {fused:7.1f} us ({wbytes / fused / 1e3:6.1f} GB/s of gate weights) | "
              f"F.linear(bf16)+.float()+topk {base:7.1f} us | x{base / fused:.2f}")


if __name__ == "__main__":
    main()
