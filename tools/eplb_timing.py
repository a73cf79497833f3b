#!/usr/bin/env python3
"""EPLB device pieces on one MI355X (development / profiles): the id-map + load-recording kernel behind the
router (hipGraph replay, per launch) and the expert-image copies (GB/s of HBM->HBM), at the SURVEY 8 shapes."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from lvllm_amd import eplb, ops  # noqa: E402
from tools.router_timing import timed  # noqa: E402

# (name, tokens, top_k, logical experts, redundant slots)
SHAPES = [("Mixtral-8x7B decode", 32, 2, 8, 8), ("Qwen3-30B-A3B decode", 128, 8, 128, 16),
          ("DeepSeek-V3 decode", 256, 8, 256, 32), ("GLM-4.5-Air prefill", 8192, 8, 128, 16)]


def main():
    dev = torch.device("cuda", 0)
    for name, M, K, E, red in SHAPES:
        P = E + red
        rng = np.random.default_rng(1)
        w = (1.0 / np.arange(1, E + 1))[None, :].astype(np.float32)                 # Zipf loads -> replicas of hot experts
        p2l = eplb.rebalance_experts(w, P, 1, 1, 8)
        l2p, cnt = eplb.compute_logical_maps(p2l, E, max_slots=red + 1)
        l2p, cnt = l2p[0].to(torch.int32).to(dev), cnt[0].to(torch.int32).to(dev)
        prob = w[0].astype(np.float64)
        prob /= prob.sum()
        ids = torch.from_numpy(rng.choice(E, size=(M, K), p=prob).astype(np.int32)).to(dev)
        load = torch.zeros(P, dtype=torch.int32, device=dev)
        sw = torch.ones((), dtype=torch.int32, device=dev)
        t_rec = timed(lambda: ops.eplb_map_to_physical_and_record(ids, load, l2p, cnt, sw))
        t_map = timed(lambda: ops.eplb_map_to_physical_and_record(ids, None, l2p, cnt))
        print(f"{name:22s} M={M:5d} K={K} E={E:3d} P={P:3d}: map+record {t_rec:6.1f} us | map only {t_map:6.1f} us "
              f"(incl. one torch.empty per call)")
    # expert images: Mixtral bf16 expert (352 MB) and a DeepSeek-V3 fp8 expert (44 MB)
    for name, H, I, fmt in [("Mixtral bf16 expert", 4096, 14336, "bf16"), ("DSv3 fp8 expert", 7168, 2048, "fp8")]:
        E = 2
        if fmt == "bf16":
            eng = ops.RoutedExpertsEngine(torch.zeros((E, 2 * I, H), dtype=torch.bfloat16, device=dev),
                                          torch.zeros((E, H, I), dtype=torch.bfloat16, device=dev), top_k=1,
                                          act_dtype=torch.bfloat16)
        else:
            eng = ops.RoutedExpertsEngine(torch.zeros((E, 2 * I, H), dtype=torch.uint8, device=dev),
                                          torch.zeros((E, H, I), dtype=torch.uint8, device=dev), top_k=1,
                                          act_dtype=torch.bfloat16, fmt="fp8", group_n=128, group_k=128,
                                          w13_scale=torch.ones((E, 2 * I // 128, H // 128), device=dev),
                                          w2_scale=torch.ones((E, H // 128, I // 128), device=dev))
        st = eplb.EngineExpertStore(eng)
        img = torch.empty(st.expert_nbytes, dtype=torch.uint8, device=dev)
        for what, fn in (("export", lambda: st.export_expert(0, img)), ("import", lambda: st.import_expert(1, img))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                fn()
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) * 1e3 / 20
            print(f"{name:22s} image {st.expert_nbytes / 1e6:7.1f} MB: {what} {us:8.1f} us = "
                  f"{2 * st.expert_nbytes / us / 1e3:6.0f} GB/s read+write")


if __name__ == "__main__":
    main()
