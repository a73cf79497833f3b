#!/usr/bin/env python3
"""One-rank RCCL check of the reference's modular call sequence on the GPU: LkmPrepareAndFinalize.prepare ->
LkmExperts.apply -> finalize (lvllm_amd/modular.py) against the engine called directly.  The multi-rank logic is
covered on gloo (tests/test_ep_gloo.py); this exercises the HIP pack kernel + RCCL all-to-all + engine together."""
import os
import sys
from pathlib import Path

import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from lvllm_amd import ops  # noqa: E402
from lvllm_amd.modular import LkmExperts, LkmPrepareAndFinalize  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    E, K, H, I, M = 8, 2, 512, 256, 37
    g = torch.Generator(device=dev).manual_seed(3)
    x = (torch.randn((M, H), generator=g, device=dev) / 10).to(torch.bfloat16)
    w13 = (torch.randn((E, 2 * I, H), generator=g, device=dev) / 10).to(torch.bfloat16)
    w2 = (torch.randn((E, H, I), generator=g, device=dev) / 10).to(torch.bfloat16)
    tw, ids = ops.topk_softmax(torch.randn((M, E), generator=g, device=dev), K, True)
    direct = ops.RoutedExpertsEngine(w13, w2, top_k=K, act_dtype=torch.bfloat16).decode(x, tw, ids)
    pf, ex = LkmPrepareAndFinalize(E, H), LkmExperts()
    a1q, a1q_scale, meta, ids_d, w_d = pf.prepare(x, tw, ids, E, None, False, None, True)
    fused = torch.empty((a1q.size(0), H), dtype=torch.float32, device=dev)
    ex.apply(fused, a1q, w13, w2, w_d, ids_d, "silu", E, None, None, None, None, None, None, False)
    out = torch.empty((M, H), dtype=torch.float32, device=dev)
    pf.finalize(out, fused, tw, ids, False, ex.finalize_weight_and_reduce_impl())
    torch.cuda.synchronize()
    err = float((out - direct).abs().max())
    print(f"prepare -> apply -> finalize vs direct engine: max abs diff {err:.3e} (max |out| {float(direct.abs().max()):.3f})")
    assert err <= 1e-5 * max(1.0, float(direct.abs().max())), err
    dist.barrier()
    dist.destroy_process_group()
    print("OK")


if __name__ == "__main__":
    main()
