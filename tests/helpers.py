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
