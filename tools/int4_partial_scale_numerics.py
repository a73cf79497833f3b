#!/usr/bin/env python3
"""CPU only (torch): numerics of the two ways to apply an int4 group scale in a W4A16 GEMM, at Mixtral's K.
A = the reference's: weight = act_dtype((q - 8) * s), then the 16-bit MFMA (fused_moe.py:237-276; what the engine
    does today, bit for bit: 16-19 VALU per 8 weights, the kernel is issue-bound at ~41 % of the HBM roof).
B = candidate for the next round: (q - 8) is exact in bf16, accumulate it per 128-k group in fp32 on the MFMA and
    multiply the fp32 partial sum by s (7 VALU per 8 weights).
Prints both against the fp64 product of the dequantised weights, and |A - B| against the reference test's atol."""
import torch

torch.manual_seed(0)
M, N, K, g = 128, 1024, 4096, 128
x = (torch.randn(M, K) / 10).to(torch.bfloat16)
w = torch.randn(N, K) / 10
wg = w.view(N, K // g, g)
s = torch.maximum((wg.max(-1).values / 7).abs(), (wg.min(-1).values / -8).abs()).to(torch.bfloat16)
q = torch.round(wg / s.float()[..., None]).clamp(-8, 7)
true = x.double() @ (q.double() * s.double()[..., None]).view(N, K).T
yA = x.float() @ (q * s.float()[..., None]).to(torch.bfloat16).view(N, K).float().T
yB = (torch.einsum("mgk,ngk->mng", x.float().view(M, K // g, g), q) * s.float()[None]).sum(-1)
for name, y in (("A  act_dtype((q-8)*s), reference", yA), ("B  s * fp32 partial sums", yB)):
    err = (y.double() - true).abs()
    print(f"{name:34s} max abs err {err.max():.3e}  rms {err.pow(2).mean().sqrt():.3e}  (max|y| {true.abs().max():.3f})")
print(f"max |A - B| = {float((yA - yB).abs().max()):.3e}   (reference int4 test: atol 2e-2, test_moe.py:565-693)")
