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


# ------------------------------------------------------------------ quantised experts (sgl-kernels CPU MoE)
def fp8_case(M, N, K, E, topk, seed=0):
    """tests/kernels/moe/test_cpu_quant_fused_moe.py:117-181: fp8 weights ~ randn * 448, block scales ~ randn * 1e-3
    (either sign), activations ~ randn / sqrt(K), softmax + torch.topk routing."""
    torch.manual_seed(seed)
    a = torch.randn(M, K, dtype=torch.bfloat16) / K ** 0.5
    w1 = (torch.randn(E, 2 * N, K) * 448.0).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    w2 = (torch.randn(E, K, N) * 448.0).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    w1_s = torch.randn(E, -(-2 * N // 128), -(-K // 128)) * 1e-3
    w2_s = torch.randn(E, -(-K // 128), -(-N // 128)) * 1e-3
    score = torch.softmax(torch.randn(M, E, dtype=torch.bfloat16), dim=-1, dtype=torch.float32)
    tw, ids = torch.topk(score, topk)
    return a, w1, w2, w1_s, w2_s, tw, ids.to(torch.int32)


FP8_CONFIGS = [(256, 512, 8, 2), (512, 256, 8, 4), (512, 512, 8, 4), (768, 2048, 8, 2), (768, 2048, 128, 8)]


@pytest.mark.parametrize("M", [1, 2, 64, 121])
@pytest.mark.parametrize("N,K,E,topk", FP8_CONFIGS)
def test_oracle_matches_reference_fp8_w8a16_cpu_kernel(M, N, K, E, topk):
    """fp8 W8A16 with 128x128 block scales (the lk_moe MOE_FP8 semantics, routed_experts.py:1627-1668) against the
    reference's CPU kernel for exactly that scheme; shapes and tolerance (atol = rtol = 1e-2) of
    test_cpu_quant_fused_moe.py:150-201."""
    a, w1, w2, w1_s, w2_s, tw, ids = fp8_case(M, N, K, E, topk)
    got = ref.fused_experts_fp8_w8a16(a, w1, w2, w1_s, w2_s, tw, ids)
    d = orc.MoeDesc(E=E, H=K, I=N, act_dtype=orc.BF16, wfmt=orc.W_FP8, groupN=128, groupK=128)
    want = orc.moe(d, torch_to_bits(w1), torch_to_bits(w2), torch_to_bits(a), ids.numpy(), tw.numpy(),
                   s13=w1_s.numpy(), s2=w2_s.numpy())
    torch.testing.assert_close(torch.from_numpy(want).bfloat16(), got, atol=1e-2, rtol=1e-2)


def mxfp4_quantize(w: torch.Tensor):
    """OCP MXFP4 of a [.., K] tensor in blocks of 32 along K: shared E8M0 exponent ceil(log2(amax / 6)), E2M1 codes
    by round-to-nearest on the magnitude grid {0, .5, 1, 1.5, 2, 3, 4, 6}, two codes per byte (even element in the
    low nibble) -- the recipe of test_cpu_quant_fused_moe.py:226-262 (MXFP4QuantizeUtil.quantize)."""
    shp = w.shape
    blk = w.float().reshape(-1, 32)
    amax = blk.abs().amax(dim=1, keepdim=True)
    e = torch.ceil(torch.maximum(torch.log2(amax / 6.0), torch.tensor(-127.0)))
    y = blk / torch.exp2(e)
    bounds = torch.tensor([0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5.0])
    code = (y.abs().unsqueeze(-1) > bounds).sum(dim=-1).to(torch.uint8)
    code = code | ((y < 0).to(torch.uint8) << 3)
    code = code.reshape(shp)
    packed = code[..., 0::2] | (code[..., 1::2] << 4)
    scale = (e + 127).to(torch.uint8).reshape(*shp[:-1], shp[-1] // 32)
    return packed.contiguous(), scale.contiguous()


def mxfp4_case(M, N, K, E, topk, seed=0):
    """test_cpu_quant_fused_moe.py:398-420: activations and master weights ~ randn / 10."""
    torch.manual_seed(seed)
    a = torch.randn(M, K, dtype=torch.bfloat16) / 10
    q1, s1 = mxfp4_quantize(torch.randn(E, 2 * N, K, dtype=torch.bfloat16) / 10)
    q2, s2 = mxfp4_quantize(torch.randn(E, K, N, dtype=torch.bfloat16) / 10)
    score = torch.softmax(torch.randn(M, E, dtype=torch.bfloat16), dim=-1, dtype=torch.float32)
    tw, ids = torch.topk(score, topk)
    return a, q1, q2, s1, s2, tw, ids.to(torch.int32)


MXFP4_CONFIGS = [(128, 128, 4, 2), (256, 256, 8, 4), (352, 256, 8, 4), (512, 320, 8, 4)]


@pytest.mark.parametrize("M", [1, 2, 32, 121])
@pytest.mark.parametrize("N,K,E,topk", MXFP4_CONFIGS)
def test_oracle_matches_reference_mxfp4_cpu_kernel(M, N, K, E, topk):
    """MXFP4 experts (lk_moe MOE_MXFP4, routed_experts.py:1770-1813) against the reference's CPU kernel; shapes and
    tolerance (atol = rtol = 1e-2) of test_cpu_quant_fused_moe.py:386-441."""
    a, q1, q2, s1, s2, tw, ids = mxfp4_case(M, N, K, E, topk)
    got = ref.fused_experts_mxfp4(a, q1, q2, s1, s2, tw, ids)
    d = orc.MoeDesc(E=E, H=K, I=N, act_dtype=orc.BF16, wfmt=orc.W_MXFP4, groupN=1, groupK=32)
    want = orc.moe(d, q1.numpy(), q2.numpy(), torch_to_bits(a), ids.numpy(), tw.numpy(), s13=s1.numpy(), s2=s2.numpy())
    torch.testing.assert_close(torch.from_numpy(want).bfloat16(), got, atol=1e-2, rtol=1e-2)
    # our own tighter statement: both decode E2M1 x E8M0 exactly, so only summation order and the bf16 output differ
    torch.testing.assert_close(torch.from_numpy(want), got.float(), atol=2e-3 * max(1.0, float(np.abs(want).max())), rtol=1e-2)


# ------------------------------------------------------------------ int4 (GPTQ / compressed-tensors packing)
def int4_case(M, N, K, E, topk, g=128, seed=0):
    """tests/kernels/moe/test_cpu_quant_fused_moe.py:591-614, 686-690: random nibbles, scales |randn * 0.01| + 0.001,
    activations randn / (0.5 sqrt(K)); packed along the input dim into int32 words (nibble j = channel 8*row + j),
    i.e. the checkpoint layout w13_packed [E, H/8, 2I] that LvLLM's _process_wna16 receives."""
    torch.manual_seed(seed)
    a = torch.randn(M, K, dtype=torch.bfloat16) / (0.5 * K ** 0.5)
    w1 = torch.randint(0, 16, (E, K, 2 * N), dtype=torch.int32)
    w2 = torch.randint(0, 16, (E, N, K), dtype=torch.int32)
    s1 = (torch.randn(E, K // g, 2 * N, dtype=torch.bfloat16) * 0.01).abs() + 0.001
    s2 = (torch.randn(E, N // g, K, dtype=torch.bfloat16) * 0.01).abs() + 0.001

    def pack(w):
        e, kin, nout = w.shape
        v = w.view(e, kin // 8, 8, nout)
        out = torch.zeros(e, kin // 8, nout, dtype=torch.int32)
        for j in range(8):
            out |= (v[:, :, j, :] & 0xF) << (4 * j)
        return out

    score = torch.softmax(torch.randn(M, E, dtype=torch.bfloat16), dim=-1, dtype=torch.float32)
    tw, ids = torch.topk(score, topk)
    return a, pack(w1), pack(w2), s1, s2, tw, ids.to(torch.int32)


def lk_moe_int4_layout(packed: torch.Tensor, scale: torch.Tensor):
    """what RoutedExperts._process_wna16 passes to lk_moe (routed_experts.py:1461-1479): packed.transpose(1, 2)
    .contiguous().view(uint8) = [E, out, in/2] (low nibble = even input channel), scales [E, out, groups]."""
    return packed.transpose(1, 2).contiguous().view(torch.uint8), scale.transpose(1, 2).contiguous()


def assert_close_to_w4a8_kernel(got: torch.Tensor, kernel_out: torch.Tensor):
    """The reference kernel quantises activations to int8 (W4A8); a W4A16 result differs from it by that noise.
    Measured on these cases: max <= 3 %, mean <= 0.7 % of max|out| (worst at a single token).  A wrong nibble order,
    zero point or scale layout gives O(100 %): test_int4_format_criterion_rejects_a_wrong_nibble_order."""
    ref_max = float(kernel_out.float().abs().max())
    diff = (got.float() - kernel_out.float()).abs()
    assert float(diff.max()) <= 0.06 * ref_max, (float(diff.max()), ref_max)
    assert float(diff.mean()) <= 0.015 * ref_max, (float(diff.mean()), ref_max)


@pytest.mark.parametrize("M", [1, 64, 121])
@pytest.mark.parametrize("N,K,E,topk", [(256, 512, 8, 2), (512, 256, 8, 2), (512, 512, 8, 4), (768, 2048, 8, 2)])
def test_oracle_int4_format_matches_reference_cpu_kernel(M, N, K, E, topk):
    a, p1, p2, s1, s2, tw, ids = int4_case(M, N, K, E, topk)
    kern = ref.fused_experts_int4_gptq(a, p1, p2, s1, s2, tw, ids)
    q13, sc13 = lk_moe_int4_layout(p1, s1)
    q2, sc2 = lk_moe_int4_layout(p2, s2)
    d = orc.MoeDesc(E=E, H=K, I=N, act_dtype=orc.BF16, wfmt=orc.W_INT4, groupN=1, groupK=128)
    want = orc.moe(d, q13.numpy(), q2.numpy(), torch_to_bits(a), ids.numpy(), tw.numpy(),
                   s13=torch_to_bits(sc13), s2=torch_to_bits(sc2))
    assert_close_to_w4a8_kernel(torch.from_numpy(want), kern)


def test_int4_format_criterion_rejects_a_wrong_nibble_order():
    """negative control: the same weights with the two nibbles of every byte swapped (or the zero point off by one)
    must be far outside the acceptance band used above."""
    a, p1, p2, s1, s2, tw, ids = int4_case(64, 256, 512, 8, 2)
    kern = ref.fused_experts_int4_gptq(a, p1, p2, s1, s2, tw, ids)
    q13, sc13 = lk_moe_int4_layout(p1, s1)
    q2, sc2 = lk_moe_int4_layout(p2, s2)
    d = orc.MoeDesc(E=8, H=512, I=256, act_dtype=orc.BF16, wfmt=orc.W_INT4, groupN=1, groupK=128)
    swap = lambda q: ((q >> 4) | (q << 4)).to(torch.uint8)
    bad = orc.moe(d, swap(q13).numpy(), q2.numpy(), torch_to_bits(a), ids.numpy(), tw.numpy(),
                  s13=torch_to_bits(sc13), s2=torch_to_bits(sc2))
    with pytest.raises(AssertionError):
        assert_close_to_w4a8_kernel(torch.from_numpy(bad), kern)
    ref_max = float(kern.float().abs().max())
    assert float((torch.from_numpy(bad) - kern.float()).abs().max()) > 0.3 * ref_max
