#!/usr/bin/env python3
"""BASELINE.json configs[0] -- "Qwen3-30B-A3B bf16, 1 MoE layer, batch=1 seq=1 on the reference CPU path (plumbing,
no GPU)".  Runs here, on the CPU: the reference's own in-tree CPU fused-MoE kernel (oracle/_ref) and the oracle on
one Qwen3-30B-A3B-shaped layer (E 128, top-8, H 2048, I 768), routing through the oracle's softmax top-k + renorm;
checks they agree (the reference test's bf16 tolerance) and prints both times.  lk_moe itself is a closed wheel that
is not in the reference tree.  Test infrastructure: this script is not part of the product path."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import oracle as orc  # noqa: E402
from oracle import ref  # noqa: E402
from tests.helpers import torch_to_bits  # noqa: E402


def main():
    E, K, H, I, M = 128, 8, 2048, 768, 1
    g = torch.Generator().manual_seed(7)
    x = (torch.randn((M, H), generator=g) / 10).to(torch.bfloat16)
    w13 = (torch.randn((E, 2 * I, H), generator=g) / 10).to(torch.bfloat16)
    w2 = (torch.randn((E, H, I), generator=g) / 10).to(torch.bfloat16)
    logits = torch.randn((M, E), generator=g)
    tw, ids = orc.topk_softmax(logits.numpy(), K, renormalize=True)
    d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)
    bw13, bw2, bx = torch_to_bits(w13), torch_to_bits(w2), torch_to_bits(x)

    def timed(fn, n=50):
        fn()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        return (time.perf_counter() - t0) / n * 1e3, out

    t_or, o = timed(lambda: orc.moe(d, bw13, bw2, bx, ids, tw))
    print(f"oracle (port, {orc.num_threads()} threads): {t_or:.3f} ms / layer-step, routed experts {sorted(ids[0].tolist())}")
    if not ref.available():
        print("oracle/_ref not built here")
        return
    p13, p2 = ref.prepack(w13), ref.prepack(w2)
    ttw, tids = torch.from_numpy(tw), torch.from_numpy(ids)
    t_ref, r = timed(lambda: ref.fused_moe(x, p13, p2, ttw, tids))
    print(f"reference csrc/cpu/cpu_fused_moe.cpp ({ref.num_threads()} threads): {t_ref:.3f} ms / layer-step "
          f"-> {M / t_ref * 1e3:.0f} tokens/s; weights touched {K * 3 * H * I * 2 / 1e6:.1f} MB -> {K * 3 * H * I * 2 / t_ref / 1e6:.1f} GB/s")
    err = np.abs(o - r.float().numpy()).max()
    scale = max(1.0, float(np.abs(o).max()))
    assert err <= 1e-3 * scale + 1.6e-2 * float(np.abs(o).max()), err
    print(f"max |oracle - reference kernel| = {err:.3e} (max |out| {np.abs(o).max():.3f}); MI355X, same shape: 31.7 us / step "
          f"(profiles/r01_report_all_configs.md, row 1)")


if __name__ == "__main__":
    main()
