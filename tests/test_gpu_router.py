"""Router GEMM + routing on the GPU vs the CPU oracle.

Tolerances: logits vs the exact (double-accumulated) value: 2e-4 * max(1, |logits|max), the reference's
ATOL_FP32 for its fp32 router GEMM (tests/kernels/test_fp32_router_gemm.py:24-25).  Routing on those
logits: bit-identical to the stand-alone routing operators given the same logits (which are themselves
bit-exact vs the oracle, tests/test_gpu_routing.py).
"""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.helpers import torch_to_bits

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(M, H, E, xdt, wdt, seed, bias=False):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn((M, H), generator=g) / 4).to(xdt)
    w = (torch.randn((E, H), generator=g) / 8).to(wdt)
    gb = torch.randn((E,), generator=g) / 4 if bias else None
    return x, w, gb


def _oracle_logits(x, w, gb, round_dt=orc.F32):
    xd = orc.BF16 if x.dtype == torch.bfloat16 else orc.F16
    if w.dtype == torch.float32:
        wb, wd = w.numpy(), orc.F32
    else:
        wb, wd = torch_to_bits(w), xd
    return orc.router_logits(torch_to_bits(x), xd, wb, wd, None if gb is None else gb.numpy(), round_dt)


@pytest.mark.parametrize("M,H,E,K", [(1, 4096, 8, 2), (32, 4096, 8, 2), (7, 2048, 128, 8), (33, 6144, 128, 4),
                                     (300, 1024, 60, 6), (16, 7168, 256, 8)])
@pytest.mark.parametrize("xdt,wdt", [(torch.bfloat16, torch.bfloat16), (torch.float16, torch.float16),
                                     (torch.bfloat16, torch.float32)])
def test_router_logits_and_topk(M, H, E, K, xdt, wdt):
    from lvllm_amd import ops
    x, w, gb = _case(M, H, E, xdt, wdt, seed=M * 7 + E, bias=(E == 128))
    tw, ids, logits = ops.router_topk(x.to(DEV), w.to(DEV), K, True, gate_bias=None if gb is None else gb.to(DEV),
                                      return_logits=True)
    ref = _oracle_logits(x, w, gb)
    np.testing.assert_allclose(logits.cpu().numpy(), ref, atol=2e-4 * max(1.0, float(np.abs(ref).max())), rtol=0)
    tw2, ids2 = ops.topk_softmax(logits, K, True)
    assert torch.equal(ids, ids2) and torch.equal(tw, tw2)
    # and the oracle's routing of OUR logits (bit-exact chain)
    ow, oi = orc.topk_softmax(logits.cpu().numpy(), K, renormalize=True)
    np.testing.assert_array_equal(ids.cpu().numpy(), oi)
    np.testing.assert_array_equal(tw.cpu().numpy(), ow)


@pytest.mark.parametrize("M,H,E,K", [(1500, 4096, 128, 8), (1024, 2048, 8, 2), (2100, 7168, 256, 8), (1300, 1024, 60, 6)])
@pytest.mark.parametrize("xdt", [torch.bfloat16, torch.float16])
def test_router_prefill_sizes_run_on_the_tiled_grouped_gemm(M, H, E, K, xdt):
    """M >= 1024 with 16-bit gate weights: the gate projection is one more grouped GEMM (a single expert of E
    rows over all M tokens) on gemm_tiled_kernel, fp32 split-K slabs summed by the routing kernel.  Ragged last
    token tile, E not a multiple of 16, one expert tile only (E = 8)."""
    from lvllm_amd import ops
    x, w, gb = _case(M, H, E, xdt, xdt, seed=M + E, bias=(E == 128))
    tw, ids, logits = ops.router_topk(x.to(DEV), w.to(DEV), K, True, gate_bias=None if gb is None else gb.to(DEV),
                                      return_logits=True)
    ref = _oracle_logits(x, w, gb)
    np.testing.assert_allclose(logits.cpu().numpy(), ref, atol=2e-4 * max(1.0, float(np.abs(ref).max())), rtol=0)
    ow, oi = orc.topk_softmax(logits.cpu().numpy(), K, renormalize=True)
    np.testing.assert_array_equal(ids.cpu().numpy(), oi)
    np.testing.assert_array_equal(tw.cpu().numpy(), ow)
    # a second call on the same workspace (other inputs in between) gives the same bits
    ops.router_topk(x[:64].to(DEV), w.to(DEV), K, True)
    tw2, ids2 = ops.router_topk(x.to(DEV), w.to(DEV), K, True, gate_bias=None if gb is None else gb.to(DEV))
    assert torch.equal(ids, ids2) and torch.equal(tw, tw2)


def test_router_grouped_sigmoid_bias_and_rounded_logits():
    from lvllm_amd import ops
    M, H, E, K = 24, 7168, 256, 8
    x, w, _ = _case(M, H, E, torch.bfloat16, torch.bfloat16, seed=11)
    sb = torch.randn((E,), generator=torch.Generator().manual_seed(5)) / 10
    tw, ids, logits = ops.router_topk(x.to(DEV), w.to(DEV), K, True, scoring_func="sigmoid", num_expert_group=8,
                                      topk_group=4, routed_scaling_factor=2.5, e_score_correction_bias=sb.to(DEV),
                                      logits_dtype=torch.bfloat16, return_logits=True)
    ref32 = _oracle_logits(x, w, None)
    got = logits.cpu().numpy()
    # the logits are bf16-valued and within one bf16 ulp of the rounded exact value
    assert np.array_equal(got, orc.bits_to_f32(orc.f32_to_bits(got, orc.BF16), orc.BF16))
    np.testing.assert_allclose(got, ref32, atol=0, rtol=2.0 ** -7)
    tw2, ids2 = ops.grouped_topk(x, logits, K, True, 8, 4, "sigmoid", 2.5, sb.to(DEV))
    assert torch.equal(ids, ids2) and torch.equal(tw, tw2)


def test_router_is_graph_capturable_and_loud():
    from lvllm_amd import ops
    from lvllm_amd._clib import LkmError
    x, w, _ = _case(32, 4096, 8, torch.bfloat16, torch.bfloat16, seed=3)
    xd, wd = x.to(DEV), w.to(DEV)
    ref = ops.router_topk(xd, wd, 2, True)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            out = ops.router_topk(xd, wd, 2, True)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out[1], ref[1]) and torch.equal(out[0], ref[0])
    with pytest.raises(LkmError):
        ops.router_topk(xd[:, :100].contiguous(), wd[:, :100].contiguous(), 2, True)      # H % 32 != 0
    with pytest.raises(LkmError):
        ops.router_topk(xd, wd.to(torch.float16), 2, True)                                # dtype mix
