"""Shared helpers for the parity tests (test infrastructure; may use the oracle)."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from oracle import oracle as orc

GOLDEN = Path(__file__).resolve().parent / "golden"
TORCH_DT = {orc.F32: torch.float32, orc.BF16: torch.bfloat16, orc.F16: torch.float16}


def load_golden(name: str):
    z = np.load(GOLDEN / name)
    n = int(z["n"])
    for i in range(n):
        key = f"c{i}_meta"
        if key not in z:
            continue
        yield i, {k[len(f"c{i}_"):]: z[k] for k in z.files if k.startswith(f"c{i}_")}


def bits_to_torch(a: np.ndarray, dt: int) -> torch.Tensor:
    """uint16 bit patterns -> torch tensor of dtype dt (bf16/f16); f32 arrays pass through."""
    if dt == orc.F32:
        return torch.from_numpy(np.ascontiguousarray(a))
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).view(TORCH_DT[dt])


def torch_to_bits(t: torch.Tensor) -> np.ndarray:
    t = t.detach().cpu().contiguous()
    if t.dtype in (torch.bfloat16, torch.float16):
        return t.view(torch.int16).numpy().view(np.uint16)
    if t.dtype == torch.float8_e4m3fn:
        return t.view(torch.uint8).numpy()
    return t.numpy()


def dt_of(t: torch.Tensor) -> int:
    return {torch.float32: orc.F32, torch.bfloat16: orc.BF16, torch.float16: orc.F16}[t.dtype]


def make_routing(M: int, E: int, K: int, seed: int = 0, skew: float = 0.0, drop: float = 0.0):
    """Seeded synthetic routing through the ORACLE router (softmax + renorm).  skew>0 biases
    the logits Zipf-like; drop>0 replaces that fraction of ids by -1 (non-local experts)."""
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn((M, E), generator=g, dtype=torch.float32)
    if skew > 0:
        logits = logits + skew * torch.log(1.0 / torch.arange(1, E + 1, dtype=torch.float32))[None, :]
    w, ids = orc.topk_softmax(logits.numpy(), K, renormalize=True)
    if drop > 0:
        mask = torch.rand((M, K), generator=g).numpy() < drop
        ids = np.where(mask, -1, ids).astype(np.int32)
    return w, ids


def rel_err(a: np.ndarray, b: np.ndarray) -> float:
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


class TorchEpKernels:
    """torch restatement of the two expert-parallel exchange kernels (lvllm_amd/csrc/ep.hip, include/lkm.h:
    lkm_ep_row_bytes / lkm_ep_pack_tokens / lkm_ep_combine) with identical semantics and record layout; the
    gloo tests inject it as `kernels=` where the HIP kernels cannot run, tests/test_gpu_ep.py requires the HIP
    kernels to equal it bit for bit."""

    @staticmethod
    def ep_row_bytes(H: int, K: int) -> int:
        return (H * 2 + K * 8 + 15) // 16 * 16

    @staticmethod
    def ep_pack_tokens(hidden, tw, ids, num_experts, ep, capacity, send, slot_of, overflow, global_ids=False):
        from lvllm_amd.ep import owner_of
        M, K = ids.shape
        H = hidden.size(1)
        rowb = TorchEpKernels.ep_row_bytes(H, K)
        base, rem = divmod(num_experts, ep)
        first = [r * base + min(r, rem) for r in range(ep)]
        valid = (ids >= 0) & (ids < num_experts)
        owner = torch.where(valid, owner_of(ids.clamp(min=0), num_experts, ep), torch.full_like(ids, -1, dtype=torch.int64))
        send3 = send.view(ep, capacity, rowb)
        so = slot_of.view(ep, M)
        hb = hidden.contiguous().view(torch.uint8).view(M, H * 2)
        for p in range(ep):
            mine = owner == p                                  # [M, K]
            flag = mine.any(dim=1)
            slot = torch.cumsum(flag.to(torch.int64), 0) - 1
            n = int(flag.sum())
            if n > capacity:
                overflow += n - capacity
            keep = flag & (slot < capacity)
            so[p] = torch.where(keep, slot, torch.full_like(slot, -1)).to(torch.int32)
            rec_ids = torch.where(mine, ids if global_ids else ids - first[p], torch.full_like(ids, -1)).to(torch.int32)
            idx = slot[keep]
            send3[p, idx, :H * 2] = hb[keep]
            send3[p, idx, H * 2:H * 2 + 4 * K] = rec_ids[keep].contiguous().view(torch.uint8).view(-1, 4 * K)
            send3[p, idx, H * 2 + 4 * K:H * 2 + 8 * K] = tw[keep].contiguous().view(torch.uint8).view(-1, 4 * K)
            nrec = min(n, capacity)
            if nrec < capacity:
                minus1 = torch.full((capacity - nrec, K), -1, dtype=torch.int32, device=ids.device)
                send3[p, nrec:, H * 2:H * 2 + 4 * K] = minus1.view(torch.uint8).view(-1, 4 * K)

    @staticmethod
    def ep_combine(back, slot_of, out):
        ep, cap, H = back.shape
        M = out.size(0)
        acc = torch.zeros((M, H), dtype=torch.float32, device=back.device)
        so = slot_of.view(ep, M).to(torch.int64)
        for p in range(ep):                                    # p ascending: the kernel's order
            sel = so[p] >= 0
            acc[sel] = acc[sel] + back[p, so[p][sel]].to(torch.float32)
        out.copy_(acc.to(out.dtype))
        return out
