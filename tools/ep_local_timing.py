#!/usr/bin/env python3
"""What one EP rank's engine sees in bench.py's a2a-fixed mode: ep*M*K slots with K=1, 1/ep of them routed to
the E/ep local experts, the rest -1.  Times the local expert computation per planner choice (development)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from bench import make_weights  # noqa: E402
from lvllm_amd import ops  # noqa: E402


def timed(fn, reps=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    dev = torch.device("cuda", 0)
    E, K, H, I, M = 8, 2, 4096, 14336, 32
    for ep in (2, 4, 8):
        El = E // ep
        w13, w2 = make_weights(El, 0, H, I, dev, "bf16")
        eng = ops.RoutedExpertsEngine(w13, w2, top_k=1, act_dtype=torch.bfloat16, max_num_seqs=ep * M * K)
        gen = torch.Generator(device=dev).manual_seed(3)
        n = ep * M * K
        x = (torch.randn((n, H), generator=gen, device=dev) / 10).to(torch.bfloat16)
        gid = torch.randint(0, E, (n, 1), generator=gen, device=dev, dtype=torch.int32)
        ids = torch.where(gid < El, gid, torch.full_like(gid, -1))          # 1/ep of the slots are local
        tw = torch.rand((n, 1), generator=gen, device=dev)
        out = torch.empty((n, H), dtype=torch.float32, device=dev)
        floor = El * 3 * H * I * 2 / 6.8e12 * 1e6
        for cfg in ({}, {"valid_den": ep}, {"valid_den": ep, "tiled": 64}, {"valid_den": ep, "tiled": 128, "waves": 4, "pd1": 4, "pd2": 4}):
            eng.engine.set_tuning(tiled=0, tbmax=0, valid_den=0, waves=0, pd1=0, pd2=0)
            eng.engine.set_tuning(**cfg)
            t = timed(lambda: eng.decode(x, tw, ids, out=out))
            eng.engine.set_profiling(True)
            for _ in range(3):
                eng.decode(x, tw, ids, out=out)
            pr = eng.engine.get_profile()
            eng.engine.set_profiling(False)
            print("   kernels us:", {k: round(v * 1e3, 1) for k, v in pr.items()})
            print(f"ep={ep} local experts={El} slots={n} valid={int((ids >= 0).sum())} cfg={cfg or 'auto'}: {t:7.1f} us "
                  f"(HBM floor {floor:.0f} us) | {eng.engine.describe().split('|', 2)[2][:90]}")
        del eng, w13, w2


if __name__ == "__main__":
    main()
