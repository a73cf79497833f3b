"""GPU: shared experts folded into the routed grouped GEMM (lvllm_amd/shared_experts.py): the engine with E + n
experts and the extended slots equals routed experts + the dense shared MLP (CPU oracle), decode and prefill."""
import numpy as np
import pytest
import torch

from lvllm_amd import shared_experts as se
from oracle import oracle as orc
from tests.helpers import make_routing, torch_to_bits

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("M", [1, 33, 300])
@pytest.mark.parametrize("fmt", ["bf16", "fp8"])
def test_fused_shared_expert_equals_routed_plus_dense_mlp(M, fmt):
    from lvllm_amd.ops import RoutedExpertsEngine
    E, K, H, I, N = 8, 2, 512, 256, 2
    g = torch.Generator().manual_seed(11)
    w13 = (torch.randn((E, 2 * I, H), generator=g) / 4).to(torch.bfloat16)
    w2 = (torch.randn((E, H, I), generator=g) / 4).to(torch.bfloat16)
    s13 = (torch.randn((2 * N * I, H), generator=g) / 4).to(torch.bfloat16)
    s2 = (torch.randn((H, N * I), generator=g) / 4).to(torch.bfloat16)
    x = (torch.randn((M, H), generator=g) / 2).to(torch.bfloat16)
    tw, ids = make_routing(M, E, K, seed=M)
    slots = se.SharedExpertSlots(E, N, K, max_num_tokens=512, device=DEV)
    etw, eids = slots.inject(torch.from_numpy(tw).to(DEV), torch.from_numpy(ids).to(DEV))
    if fmt == "bf16":
        c13, c2, _, _ = se.split_shared_expert(s13, s2, N)
        f13, f2 = se.append_shared_experts(w13, c13), se.append_shared_experts(w2, c2)
        eng = RoutedExpertsEngine(f13.to(DEV), f2.to(DEV), top_k=K + N, act_dtype=torch.bfloat16)
        d = orc.MoeDesc(E=E + N, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)
        ref = orc.moe(d, torch_to_bits(f13), torch_to_bits(f2), torch_to_bits(x), eids.cpu().numpy(), etw.cpu().numpy())
        # ... which is routed + the shared expert as ONE dense expert of size N*I (exact algebra, CPU-tested)
        routed = orc.moe(orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16), torch_to_bits(w13),
                         torch_to_bits(w2), torch_to_bits(x), ids, tw)
        one = orc.moe(orc.MoeDesc(E=1, H=H, I=N * I, act_dtype=orc.BF16, wfmt=orc.W_BF16), torch_to_bits(s13[None]),
                      torch_to_bits(s2[None]), torch_to_bits(x), np.zeros((M, 1), np.int32), np.ones((M, 1), np.float32))
        np.testing.assert_allclose(ref, routed + one, atol=2e-5 * np.abs(one).max(), rtol=1e-5)
    else:
        q13, sc13 = orc.quant_fp8_block(w13.float().numpy(), 128, 128)
        q2, sc2 = orc.quant_fp8_block(w2.float().numpy(), 128, 128)
        sq13, ssc13 = orc.quant_fp8_block(s13[None].float().numpy(), 128, 128)
        sq2, ssc2 = orc.quant_fp8_block(s2[None].float().numpy(), 128, 128)
        c13, c2, cs13, cs2 = se.split_shared_expert(torch.from_numpy(sq13[0]), torch.from_numpy(sq2[0]), N,
                                                    w13_scale=torch.from_numpy(ssc13[0]), w2_scale=torch.from_numpy(ssc2[0]))
        f13, f2 = se.append_shared_experts(torch.from_numpy(q13), c13), se.append_shared_experts(torch.from_numpy(q2), c2)
        fs13, fs2 = se.append_shared_experts(torch.from_numpy(sc13), cs13), se.append_shared_experts(torch.from_numpy(sc2), cs2)
        eng = RoutedExpertsEngine(f13.to(DEV), f2.to(DEV), top_k=K + N, act_dtype=torch.bfloat16, fmt="fp8",
                                  w13_scale=fs13.to(DEV), w2_scale=fs2.to(DEV), group_n=128, group_k=128)
        d = orc.MoeDesc(E=E + N, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_FP8, groupN=128, groupK=128)
        ref = orc.moe(d, f13.numpy(), f2.numpy(), torch_to_bits(x), eids.cpu().numpy(), etw.cpu().numpy(),
                      s13=fs13.numpy(), s2=fs2.numpy())
    scale = max(1.0, float(np.abs(ref).max()))
    out = eng.decode(x.to(DEV), etw, eids).cpu().numpy()
    np.testing.assert_allclose(out, ref, atol=2e-3 * scale, rtol=1e-2, err_msg=eng.engine.describe())
    pre = eng.prefill(x.to(DEV), etw, eids).float().cpu().numpy()
    np.testing.assert_allclose(pre, ref, atol=2e-2 * scale, rtol=2e-2)
