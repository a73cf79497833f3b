"""The CPU oracle against the REFERENCE'S OWN in-tree CPU fused-MoE kernel run here (oracle/_ref, built by
`make -C oracle ref` from /root/reference/csrc/cpu/cpu_fused_moe.cpp where it lies).

Shapes, input recipe, seed and tolerances are those of the reference's test of that kernel
(tests/kernels/moe/test_cpu_fused_moe.py:200-262; tolerances tests/kernels/allclose_default.py:8-9:
bf16 atol 1e-3 rtol 1.6e-2, fp16 atol 1e-3 rtol 1e-3).  The GPU-side counterpart is
tests/test_gpu_moe.py::test_gpu_vs_reference_cpu_kernel."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from oracle import ref
from tests.helpers import torch_to_bits

pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference at build time)")

TOL = {torch.bfloat16: (1e-3, 1.6e-2), torch.float16: (1e-3, 1e-3)}


def reference_case(batch, E, H, I, dtype, seed=0):
    """test_cpu_fused_moe.py:219-244: inputs / weights ~ randn / (0.5 sqrt(fan_in)), softmax + torch.topk routing."""
    torch.manual_seed(seed)
    K = max(E // 2, 1)
    x = torch.randn((batch, H), dtype=dtype) / (0.5 * H ** 0.5)
    w13 = torch.randn((E, 2 * I, H), dtype=dtype) / (0.5 * H ** 0.5)
    w2 = torch.randn((E, H, I), dtype=dtype) / (0.5 * I ** 0.5)
    logits = torch.randn((batch, E), dtype=dtype)
    score = torch.softmax(logits, dim=-1, dtype=torch.float32)
    tw, ids = torch.topk(score, K)
    return x, w13, w2, tw.float(), ids.to(torch.int32)


def oracle_out(x, w13, w2, tw, ids, act, dtype):
    E, twoI, H = w13.shape
    odt = orc.BF16 if dtype == torch.bfloat16 else orc.F16
    d = orc.MoeDesc(E=E, H=H, I=twoI // 2, act_dtype=odt, wfmt=orc.W_BF16 if dtype == torch.bfloat16 else orc.W_F16,
                    activation=orc.ACT_SILU if act == "silu" else orc.ACT_SWIGLUOAI)
    out = orc.moe(d, torch_to_bits(w13), torch_to_bits(w2), torch_to_bits(x), ids.numpy(), tw.numpy())
    return torch.from_numpy(out).to(dtype)          # the in-tree kernel returns the activation dtype


@pytest.mark.parametrize("act", ["silu", "swigluoai"])
@pytest.mark.parametrize("batch", [1, 64, 256])
@pytest.mark.parametrize("H,I", [(128, 128), (2880, 128), (128, 2880), (2880, 2880)])
def test_oracle_matches_reference_cpu_kernel_bf16(batch, H, I, act):
    dtype = torch.bfloat16
    x, w13, w2, tw, ids = reference_case(batch, 8, H, I, dtype)
    got = ref.fused_moe(x, ref.prepack(w13), ref.prepack(w2), tw, ids, act=act)
    want = oracle_out(x, w13, w2, tw, ids, act, dtype)
    atol, rtol = TOL[dtype]
    torch.testing.assert_close(want.float(), got.float(), atol=atol, rtol=rtol)


def test_oracle_matches_reference_cpu_kernel_fp16_and_decode_shapes():
    """fp16 activations (the MOE_*_FP16 classes) and a Qwen3-30B-A3B-shaped layer at decode batch 1 (BASELINE configs[0])."""
    x, w13, w2, tw, ids = reference_case(64, 8, 256, 384, torch.float16)
    got = ref.fused_moe(x, ref.prepack(w13), ref.prepack(w2), tw, ids)
    want = oracle_out(x, w13, w2, tw, ids, "silu", torch.float16)
    torch.testing.assert_close(want.float(), got.float(), atol=1e-3, rtol=1e-3)
    # Qwen3-30B-A3B: H 2048, I 768, E 128, K 8, M 1; routing through the ORACLE router (softmax + renorm)
    g = torch.Generator().manual_seed(7)
    E, H, I, K = 128, 2048, 768, 8
    x = (torch.randn((1, H), generator=g) / 10).to(torch.bfloat16)
    w13 = (torch.randn((E, 2 * I, H), generator=g) / 10).to(torch.bfloat16)
    w2 = (torch.randn((E, H, I), generator=g) / 10).to(torch.bfloat16)
    w, ids = orc.topk_softmax(torch.randn((1, E), generator=g).numpy(), K, renormalize=True)
    tw, ids = torch.from_numpy(w), torch.from_numpy(ids)
    got = ref.fused_moe(x, ref.prepack(w13), ref.prepack(w2), tw, ids)
    want = oracle_out(x, w13, w2, tw, ids, "silu", torch.bfloat16)
    scale = float(want.float().abs().max())
    torch.testing.assert_close(want.float(), got.float(), atol=1e-3 * max(1.0, scale), rtol=1.6e-2)


def test_reference_kernel_is_deterministic_and_thread_count_independent():
    x, w13, w2, tw, ids = reference_case(64, 8, 128, 128, torch.bfloat16)
    p13, p2 = ref.prepack(w13), ref.prepack(w2)
    n0 = ref.num_threads()
    a = ref.fused_moe(x, p13, p2, tw, ids)
    ref.set_threads(1)
    b = ref.fused_moe(x, p13, p2, tw, ids)
    ref.set_threads(n0)
    assert torch.equal(a, b)
